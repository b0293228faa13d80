"""Probe (GPU box): UNet cfg1-shape frozen-BN parameter gradients under both conv arithmetics against the torch-CPU fp32 oracle
AND its fp64 run: is a per-tensor relative-L2 gap of ~2e-3 between two fp32 evaluations ReLU-flip noise (then the CPU fp32 oracle
is as far from fp64) or an accuracy loss of the arithmetic?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-segmentation_amd"))
import torch
import models
from oracle import losses_ref, pspnet_ref, unet_ref
from oracle.weights import synth_batch, synth_state_dict
from segmi import ops
from utils.losses import CrossEntropyLoss2d
rec = torch.load(os.path.join(ROOT, "tests", "golden", "unet.pt"), weights_only=False)["s64"]
sd = synth_state_dict(rec["manifest"], seed=2)
x, t = synth_batch(2, 3, 256, 256, 2, seed=99)
refs = {}
for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
    ref = pspnet_ref.clone_state({k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()})
    losses_ref.cross_entropy(unet_ref.unet_forward(ref, x.to(dt), training=True, bn_training=False), t).backward()
    refs[name] = {k: v.grad.double() for k, v in ref.items() if v.grad is not None}
dev = torch.device("cuda:0")
got = {}
for math in ("f32", "bf16x3"):
    ops.set_conv_math(math)
    m = models.UNet(2, freeze_bn=True); m.load_state_dict(sd); m.to(dev).train(); m.freeze_bn()
    CrossEntropyLoss2d(ignore_index=255)(m(x.to(dev)), t.to(dev)).backward()
    got[math] = {k: p.grad.detach().cpu().double() for k, p in m.named_parameters()}
def l2(a, b): return ((a - b).norm() / (b.norm() + 1e-30)).item()
rows = []
for k in refs["f64"]:
    rows.append((k, l2(got["f32"][k], refs["f32"][k]), l2(got["bf16x3"][k], refs["f32"][k]), l2(got["f32"][k], refs["f64"][k]),
                 l2(got["bf16x3"][k], refs["f64"][k]), l2(refs["f32"][k], refs["f64"][k])))
rows.sort(key=lambda r: -max(r[1:]))
print("%-34s %10s %10s | %10s %10s %10s" % ("tensor", "f32~cpu32", "x3~cpu32", "f32~f64", "x3~f64", "cpu32~f64"))
for r in rows[:12]:
    print("%-34s %10.2e %10.2e | %10.2e %10.2e %10.2e" % r)
import statistics
for i, n in enumerate(("HIP f32 vs cpu32", "HIP bf16x3 vs cpu32", "HIP f32 vs f64", "HIP bf16x3 vs f64", "cpu32 vs f64")):
    v = [r[i + 1] for r in rows]
    print("%-22s median %.2e max %.2e" % (n, statistics.median(v), max(v)))
