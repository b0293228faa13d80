"""Probe (GPU box): rounding error of the HIP convolution kernels against an fp64 convolution, next to torch-CPU fp32's error on
the same operands — forward, data gradient and filter gradient, direct and Winograd, at reduction lengths of the hot path.
    python tools/probes/conv_error_vs_fp64.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-segmentation_amd"))
import torch
import torch.nn.functional as F

from segmi import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def err(a, ref):
    a, ref = a.double().cpu(), ref.double()
    return ((a - ref).abs().max() / ref.abs().max()).item(), ((a - ref).norm() / ref.norm()).item()


print("%-34s %-8s | %-23s | %-23s | %s" % ("layer", "pass", "HIP max / rms", "torch-CPU fp32 max / rms", "rms ratio"))
for name, N, C, K, H, R, dil, wino in (("1x1 C2048->K512 32x32", 2, 2048, 512, 32, 1, 1, False), ("1x1 C256->K1024 32x32", 2, 256, 1024, 32, 1, 1, False),
                                       ("3x3 C512->K512 32x32 d2 direct", 2, 512, 512, 32, 3, 2, False), ("3x3 C512->K512 32x32 d2 winograd", 2, 512, 512, 32, 3, 2, True),
                                       ("3x3 C64->K64 64x64 direct", 2, 64, 64, 64, 3, 1, False), ("3x3 C2048->K512 32x32 winograd", 1, 2048, 512, 32, 3, 1, True)):
    ops.set_conv_winograd(wino, min_channels=0, min_subgrid=1, wgrad=wino)
    pad = dil * (R // 2)
    x = torch.relu(torch.randn(N, C, H, H, generator=g)) + 0.1
    w = torch.randn(K, C, R, R, generator=g) * (2.0 / (C * R * R)) ** 0.5
    gy = torch.randn(N, K, H, H, generator=g)
    res = {}
    for tag, dt, d in (("f64", torch.float64, "cpu"), ("cpu", torch.float32, "cpu"), ("hip", torch.float32, dev)):
        xx = x.detach().clone().to(dt).to(d).requires_grad_(True)
        ww = w.detach().clone().to(dt).to(d)
        if d != "cpu" and R > 1:
            ww = ww.contiguous(memory_format=torch.channels_last)
        ww.requires_grad_(True)
        y = ops.conv2d(xx, ww, None, 1, pad, dil) if d != "cpu" else F.conv2d(xx, ww, None, 1, pad, dil)
        y.backward(gy.to(dt).to(d))
        res[tag] = (y.detach().cpu(), xx.grad.cpu(), ww.grad.cpu())
    for i, p in enumerate(("fwd", "dgrad", "wgrad")):
        eh, ec = err(res["hip"][i], res["f64"][i]), err(res["cpu"][i], res["f64"][i])
        print("%-34s %-8s | %.2e / %.2e | %.2e / %.2e | %.2f" % (name, p, eh[0], eh[1], ec[0], ec[1], eh[1] / ec[1]))
