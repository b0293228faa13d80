"""Diagnostic (GPU box): per-parameter gradient error of a drop-in model vs the fp64 oracle, frozen BN."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-segmentation_amd")); sys.path.insert(0, ROOT)
import torch
import models
from oracle import deeplab_ref, losses_ref, pspnet_ref
from oracle.weights import synth_batch, synth_state_dict
from utils.losses import CrossEntropyLoss2d
backbone, os_, shape, classes = sys.argv[1], int(sys.argv[2]), (2, 3, int(sys.argv[3]), int(sys.argv[3])), int(sys.argv[4])
cuda = torch.device("cuda:0")
m = models.DeepLab(classes, backbone=backbone, pretrained=False, output_stride=os_, freeze_bn=True)
man = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
sd = synth_state_dict(man, seed=6)
m.load_state_dict(sd); m.to(cuda).train(); m.freeze_bn()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.eval()
x, t = synth_batch(shape[0], 3, shape[2], shape[3], classes, seed=31)
out = m(x.to(cuda)); loss = CrossEntropyLoss2d(ignore_index=255)(out, t.to(cuda)); loss.backward()
ref = pspnet_ref.clone_state({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()})
ro = deeplab_ref.deeplab_forward(ref, x.double(), backbone, os_, training=True, bn_training=False)
rl = losses_ref.cross_entropy(ro, t); rl.backward()
print("fwd max|d| %.3e  max|logit| %.3f  loss %.7f vs %.7f" % ((out.detach().cpu().double() - ro.detach()).abs().max().item(), ro.abs().max().item(), loss.item(), rl.item()))
errs = []
for k, p in m.named_parameters():
    g, r = p.grad.detach().cpu().double(), ref[k].grad
    errs.append(((g - r).norm().item() / (r.norm().item() + 1e-30), k, r.norm().item()))
for i, (e, k, n) in enumerate(errs):
    if i < 12 or e > 3e-4: print("%-50s relL2 %.2e  |g| %.2e" % (k, e, n))
