"""Diagnostic (GPU box): ReLU-mask agreement HIP vs fp64 oracle and torch-CPU fp32 vs fp64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-segmentation_amd")); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from oracle import deeplab_ref, pspnet_ref
from oracle.weights import synth_batch, synth_state_dict
import models
man = [(k, tuple(v.shape)) for k, v in models.DeepLab(19, backbone="resnet101", pretrained=False, output_stride=16).state_dict().items()]
sd = synth_state_dict(man, seed=6)
x, t = synth_batch(2, 3, 129, 129, 19, seed=31)
hip = torch.load("/tmp/diag_dma.pt")
res = {}
for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
    pre = []
    orig = F.relu
    def rec(v, *a, **k):
        pre.append(v.detach().clone())
        return orig(v, *a, **k)
    deeplab_ref.F.relu = rec
    ref = pspnet_ref.clone_state({k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}, requires_grad=False)
    with torch.no_grad():
        deeplab_ref.deeplab_forward(ref, x.to(dt), "resnet101", 16, training=True, bn_training=False)
    deeplab_ref.F.relu = orig
    res[name] = pre
print("relu sites:", len(res["f64"]), len(hip["masks"]))
tot_h = tot_c = 0
for i, (p64, p32, mh) in enumerate(zip(res["f64"], res["f32"], hip["masks"])):
    assert p64.shape == mh.shape, (i, p64.shape, mh.shape)
    fh = (mh != (p64 > 0)); fc = ((p32 > 0) != (p64 > 0))
    if fh.any() or fc.any():
        print("site %3d %-22s flips hip %d (|pre64| there: %s)  cpu32 %d" % (i, tuple(p64.shape), int(fh.sum()), ["%.1e" % v for v in p64[fh].abs().tolist()[:4]], int(fc.sum())))
    tot_h += int(fh.sum()); tot_c += int(fc.sum())
print("total flips vs fp64: hip %d, cpu32 %d" % (tot_h, tot_c))
