#!/usr/bin/env python
"""Micro-benchmark of single conv problems through the C ABI (GPU box).

    python tools/conv_bench.py [name ...] [--op fwd|dgrad|wgrad|all] [--iters 20]

Prints ms and TFLOP/s per (problem, op).  Run under `rocprofv3 --pmc ...` to read MFMA-busy and
stall counters for one kernel in isolation (profiles/*_pmc_*.txt).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-segmentation_amd"))
import torch  # noqa: E402

from segmi import ops  # noqa: E402

# name: (N, C, H, W, K, R, stride, pad, dil)   — PSPNet-R50 cfg2 shapes
PROBLEMS = {
    "psp_bottleneck": (8, 4096, 64, 64, 512, 3, 1, 1, 1),
    "l4_3x3_d4": (8, 512, 64, 64, 512, 3, 1, 4, 4),
    "l3_3x3_d2": (8, 256, 64, 64, 256, 3, 1, 2, 2),
    "l4_1x1_up": (8, 512, 64, 64, 2048, 1, 1, 0, 1),
    "l4_1x1_down": (8, 2048, 64, 64, 512, 1, 1, 0, 1),
    "l3_1x1_up": (8, 256, 64, 64, 1024, 1, 1, 0, 1),
    "l1_1x1": (8, 64, 128, 128, 256, 1, 1, 0, 1),
    "l3_1x1_down": (8, 1024, 64, 64, 256, 1, 1, 0, 1),
    "l2_1x1_up": (8, 128, 64, 64, 512, 1, 1, 0, 1),
    "l1_1x1_down": (8, 256, 128, 128, 64, 1, 1, 0, 1),
    "stem2": (8, 64, 256, 256, 64, 3, 1, 1, 1),
    "stem3": (8, 64, 256, 256, 128, 3, 1, 1, 1),
    "aux_3x3": (8, 1024, 64, 64, 512, 3, 1, 1, 1),
    # DeepLab-R101 cfg3 (16 x 513^2, output stride 16): layer3's bottleneck 1x1 layers on 33x33 maps
    "r101_l3_down": (16, 1024, 33, 33, 256, 1, 1, 0, 1),
    "r101_l3_up": (16, 256, 33, 33, 1024, 1, 1, 0, 1),
    "r101_l4_down": (16, 2048, 33, 33, 512, 1, 1, 0, 1),
    # DeepLab's ASPP: 3x3 at dilations 6 / 12 / 18 on the 33x33 map (too wide for the Winograd sub-grids: direct kernels)
    "aspp_d6": (16, 2048, 33, 33, 256, 3, 1, 6, 6),
    "aspp_d12": (16, 2048, 33, 33, 256, 3, 1, 12, 12),
    "aspp_d18": (16, 2048, 33, 33, 256, 3, 1, 18, 18),
}

ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="*", default=list(PROBLEMS))
ap.add_argument("--op", default="all")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--ldpad", type=int, default=0, help="x is a channel slice of a tensor with this many extra channels (pixel stride C + ldpad)")
args = ap.parse_args()
dev = torch.device("cuda:0")
for name in args.names:
    N, C, H, W, K, R, stride, pad, dil = PROBLEMS[name]
    x = ops.to_nhwc(torch.randn(N, C + args.ldpad, H, W, device=dev))[:, :C].requires_grad_(True)
    assert ops.ld_of(x) == C + args.ldpad
    w = torch.randn(K, C, R, R, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y_dx = ops.conv2d(x, w.detach(), None, stride, pad, dil)      # backward = dgrad only
    y_dw = ops.conv2d(x.detach(), w, None, stride, pad, dil)      # backward = wgrad only
    y = y_dx
    gy = ops.to_nhwc(torch.randn_like(y))
    flops = 2.0 * y.numel() * C * R * R
    which = ("fwd", "dgrad", "wgrad") if args.op == "all" else (args.op,)
    for op in which:
        def run():
            if op == "fwd":
                with torch.no_grad():
                    ops.conv2d(x, w, None, stride, pad, dil)
            elif op == "dgrad":
                torch.autograd.grad(y_dx, x, gy, retain_graph=True)
            else:
                torch.autograd.grad(y_dw, w, gy, retain_graph=True)
        for _ in range(3):
            run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            run()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / args.iters
        print("%-16s %-6s %8.3f ms %7.1f TF/s" % (name, op, ms, flops / ms / 1e9), flush=True)
