"""Diagnostic (GPU box): are gradient deviations ReLU-mask flips?  Saves per-BN post-ReLU sign masks and grads."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-segmentation_amd")); sys.path.insert(0, ROOT)
import torch
import models
from segmi import ops
from oracle.weights import synth_batch, synth_state_dict
from utils.losses import CrossEntropyLoss2d
tag = sys.argv[1]
cuda = torch.device("cuda:0")
m = models.DeepLab(19, backbone="resnet101", pretrained=False, output_stride=16, freeze_bn=True)
man = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
sd = synth_state_dict(man, seed=6)
m.load_state_dict(sd); m.to(cuda).train(); m.freeze_bn()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.eval()
masks = []
orig = ops.batch_norm_act
def hooked(*a, **k):
    y = orig(*a, **k)
    if k.get("relu"): masks.append((y.detach() > 0).cpu())
    return y
ops.batch_norm_act = hooked
import segmi.nn as snn
x, t = synth_batch(2, 3, 129, 129, 19, seed=31)
out = m(x.to(cuda)); loss = CrossEntropyLoss2d(ignore_index=255)(out, t.to(cuda)); loss.backward()
torch.save({"masks": masks, "grads": {k: p.grad.detach().cpu() for k, p in m.named_parameters()}, "out": out.detach().cpu()}, "/tmp/diag_%s.pt" % tag)
print(tag, "saved", len(masks), "masks")
