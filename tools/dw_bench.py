#!/usr/bin/env python
"""Depthwise 3x3 kernels alone at DeepLab-Xception's layer shapes (cfg5: 8 x 3x512x512): forward (+BN-statistics epilogue), data gradient,
filter gradient, and the fused BatchNorm+ReLU load (segmi_dwconv2d_fwd_pre) — microseconds per launch and algorithmic TB/s (input + output
once), with three rotating buffer sets so that no launch finds its operands in the Infinity Cache.

    python tools/dw_bench.py [--iters 40]           (tuning hook: SEGMI_DW_SPT = strips per thread)
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-segmentation_amd"))

SHAPES = [(8, 728, 32, 32, 1, 50), (8, 1024, 32, 32, 2, 1), (8, 1536, 32, 32, 2, 2), (8, 728, 64, 64, 1, 2), (8, 256, 128, 128, 1, 2), (8, 128, 256, 256, 1, 2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    a = ap.parse_args()
    from segmi import lib, ops
    from segmi._lib import ConvDesc, check
    dev = torch.device("cuda:0")
    st = lambda: torch.cuda.current_stream().cuda_stream
    print("SEGMI_DW_SPT=%s" % os.environ.get("SEGMI_DW_SPT", "1"))
    print("%-26s %5s %9s %9s %9s %9s   (us per launch / TB/s algorithmic)" % ("shape N,C,H,W,dil", "uses", "fwd+stats", "fwd_pre", "dgrad", "wgrad"))
    tot = [0.0] * 4
    for N, C, H, W, D, uses in SHAPES:
        sets = []
        for i in range(3):
            x = ops.empty_nhwc(N, C, H, W, dev).normal_()
            y = ops.empty_nhwc(N, C, H, W, dev)
            dy = ops.empty_nhwc(N, C, H, W, dev).normal_()
            dx = ops.empty_nhwc(N, C, H, W, dev)
            sets.append((x, y, dy, dx))
        w = torch.randn(9 * C, device=dev)
        dw = torch.empty(9 * C, device=dev)
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        d = ConvDesc(N, H, W, C, C, 3, 3, H, W, 1, D, D, ops.ld_of(sets[0][0]), ops.ld_of(sets[0][1]))
        parts = lib.segmi_dwconv2d_fwd_stats_parts(d)
        part = torch.empty(max(parts, 1) * 3 * C, device=dev)
        nws = lib.segmi_dwconv2d_wgrad_workspace(d)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)

        def fwd(x, y, dy, dx):
            check(lib.segmi_dwconv2d_fwd_stats(d, x.data_ptr(), w.data_ptr(), y.data_ptr(), part.data_ptr(), st()), "fwd")

        def pre(x, y, dy, dx):
            check(lib.segmi_dwconv2d_fwd_pre(d, x.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, w.data_ptr(), y.data_ptr(), part.data_ptr(), st()), "pre")

        def dgrad(x, y, dy, dx):
            check(lib.segmi_dwconv2d_dgrad(d, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), st()), "dgrad")

        def wgrad(x, y, dy, dx):
            check(lib.segmi_dwconv2d_wgrad(d, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nws, st()), "wgrad")

        row = []
        for k, fn in enumerate((fwd, pre, dgrad, wgrad)):
            for i in range(6):
                fn(*sets[i % 3])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.iters):
                fn(*sets[i % 3])
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            row.append((us, 2 * N * C * H * W * 4 / us / 1e6))
            tot[k] += us * uses
        print("%-26s %5d " % ("%d,%d,%d,%d,d%d" % (N, C, H, W, D), uses) + " ".join("%5.1f/%4.2f" % r for r in row))
    print("per cfg5 step (uses x us): fwd+stats %.2f ms, fwd_pre %.2f ms, dgrad %.2f ms, wgrad %.2f ms" % tuple(t / 1e3 for t in tot))


if __name__ == "__main__":
    main()
