"""Lovasz-Softmax alone at the cfg5 shard shape (8 x 150 x 512 x 512, ignore = -1): forward + backward time, survivors of the tail
pruning, for three kinds of logits and with the pruning switched off (the round-4 full sort).

    python tools/lovasz_bench.py [--iters 10] [--modes random trained saturated] [--prune 1 0]
    python tools/lovasz_bench.py --up 4 --prune 1      # the model's tail: logits at 1/4 resolution -> bilinear x4 (align_corners) -> loss,
                                                       # forward + backward down to the low-resolution gradient, fused and unfused

random    = random-init-like logits (what bench.py's synthetic step feeds the loss)
trained   = the target logit boosted by 6 on 80 % of the pixels (confident and mostly right)
saturated = boosted by 40: foreground probabilities round to 1, error 0 -> those classes keep every element (worst case)
Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-segmentation_amd"))


def make(mode, N, C, H, W, dev):
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(N, C, H, W, generator=g)
    t = torch.randint(0, C, (N, H, W), generator=g)
    t[:, : H // 20, :] = -1
    if mode != "random":
        hit = torch.rand(N, H, W, generator=g) < 0.8
        x.scatter_add_(1, t.clamp(0, C - 1).unsqueeze(1), (hit & (t >= 0)).float().unsqueeze(1) * (40.0 if mode == "saturated" else 6.0))
    from segmi import ops
    return ops.to_nhwc(x.to(dev)), t.to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--modes", nargs="+", default=["random", "trained", "saturated"])
    ap.add_argument("--prune", nargs="+", type=int, default=[1, 0])
    ap.add_argument("--shape", nargs=4, type=int, default=[8, 150, 512, 512])
    ap.add_argument("--up", type=int, default=0, help="logits live at 1/UP of the target resolution (models/deeplabv3_plus.py:361)")
    a = ap.parse_args()
    from segmi import lib, ops
    import utils.losses as L
    dev = torch.device("cuda:0")
    N, C, H, W = a.shape
    crit = L.LovaszSoftmax(ignore_index=-1)
    if a.up:
        for mode in a.modes:
            x, t = make(mode, N, C, H, W, dev)
            lo = ops.to_nhwc(torch.nn.functional.avg_pool2d(x, a.up) * (2.0 if mode == "random" else 1.0)).detach()
            del x
            for fuse in (1, 0):
                c2 = L.LovaszSoftmax(ignore_index=-1, fuse_upsample=bool(fuse))
                ld = lo.clone().requires_grad_(True)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                tt = 0.0
                for it in range(a.iters + 2):
                    ld.grad = None
                    ev[0].record()
                    loss = c2(ops.interpolate_bilinear(ld, (H, W), True), t)
                    loss.backward()
                    ev[1].record()
                    torch.cuda.synchronize()
                    if it >= 2:
                        tt += ev[0].elapsed_time(ev[1])
                kept, full = ops.lovasz_last_stats()
                print("%-9s up x%d fused=%d  upsample + fwd + bwd %7.3f ms  loss %.6f  survivors %.3f %% of C*P  |grad| %.6e"
                      % (mode, a.up, fuse, tt / a.iters, loss.item(), 100.0 * kept / (C * N * H * W), ld.grad.abs().sum().item()), flush=True)
        return
    for mode in a.modes:
        x, t = make(mode, N, C, H, W, dev)
        for prune in a.prune:
            assert lib.segmi_lovasz_set_prune(prune) == 0
            xd = x.detach().requires_grad_(True)
            for _ in range(2):
                xd.grad = None
                crit(xd, t).backward()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            tf = tb = 0.0
            for _ in range(a.iters):
                xd.grad = None
                ev[0].record()
                loss = crit(xd, t)
                ev[1].record()
                loss.backward()
                ev[2].record()
                torch.cuda.synchronize()
                tf += ev[0].elapsed_time(ev[1])
                tb += ev[1].elapsed_time(ev[2])
            kept, full = ops.lovasz_last_stats()
            print("%-9s prune=%d  fwd %7.3f ms  bwd %7.3f ms  loss %.6f  survivors %d / %d = %.3f %%  (of C*P: %.3f %%)"
                  % (mode, prune, tf / a.iters, tb / a.iters, loss.item(), kept, full, 100.0 * kept / max(full, 1),
                     100.0 * kept / (C * N * H * W)), flush=True)
    lib.segmi_lovasz_set_prune(1)


if __name__ == "__main__":
    main()
