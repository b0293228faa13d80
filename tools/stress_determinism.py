"""Bitwise-repeatability stress of the training step with SEVERAL PROCESSES SHARING ONE GPU.

Every segmi kernel is deterministic (fixed-order split-K, no float atomics), so the same weights and batch must give the same
bits on every repetition.  A race inside a kernel (a missing wait in a hand-scheduled loop, an LDS slice reused too early)
shows up as an occasional difference — most readily when another process perturbs the schedule on the same compute units,
which is exactly how the two-rank tests run on the one-GPU box.  Each worker builds the model of `tests/test_distributed_gpu.py`
(PSPNet-R50, 72x88 input: 9x11 maps, M = 198 rows per shard), runs forward + backward `--iters` times and compares logits and
all gradients with the first repetition bit for bit.

    python tools/stress_determinism.py [--procs 2] [--iters 40] [--hw 72 88] [--batch 2]
"""
import argparse
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, args, ret):
    for p in (ROOT, os.path.join(ROOT, "pytorch-segmentation_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import models
    from segmi import ops
    from utils.losses import CrossEntropyLoss2d
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    m = models.PSPNet(5, backbone="resnet50", pretrained=False).to(dev).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.eval()
    g = torch.Generator().manual_seed(11 + rank)
    h, w = args.hw
    x = torch.randn(args.batch, 3, h, w, generator=g).to(dev)
    t = torch.randint(0, 5, (args.batch, h, w), generator=g).to(dev)
    crit = CrossEntropyLoss2d(ignore_index=255, process_group=None)
    first, bad = None, []
    for it in range(args.iters):
        m.zero_grad(set_to_none=True)
        out, aux = m(x)
        loss = crit(out, t) + 0.4 * crit(aux, t)
        loss.backward()
        cur = {"out": out.detach().clone(), "aux": aux.detach().clone()}
        # the low-resolution logits the heads were interpolated from, and a second interpolation of them (diagnostics)
        for name, hi in (("out", out), ("aux", aux)):
            src = ops.upsample_source(hi)
            if src is not None:
                cur[name + ".lowres"] = src[0].detach().clone()
                cur[name + ".again"] = ops.interpolate_bilinear(src[0].detach(), hi.shape[2:], src[1]).clone()
        cur.update({k: p.grad.detach().clone() for k, p in m.named_parameters()})
        if first is None:
            first = cur
            continue
        for k, v in cur.items():
            if not torch.equal(v, first[k]):
                d = (v - first[k]).abs()
                idx = (d > 0).nonzero()
                where = "%d elements, index min %s max %s" % (idx.shape[0], idx.min(0).values.tolist(), idx.max(0).values.tolist())
                bad.append((it, k + "  [" + where + "]", d.max().item(), v.abs().max().item()))
    torch.cuda.synchronize()
    ret[rank] = {"math": "f32", "bad": bad[:40], "nbad": len(bad), "finite": bool(torch.isfinite(first["out"]).all())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=2)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--hw", type=int, nargs=2, default=[72, 88])
    args = ap.parse_args()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(args, ret), nprocs=args.procs, join=True)
    total = 0
    for r in range(args.procs):
        res = ret[r]
        total += res["nbad"]
        print("proc %d (conv math %s): %d repetitions, %d tensors differed from the first repetition%s"
              % (r, res["math"], args.iters, res["nbad"], "" if res["finite"] else "  [non-finite logits]"))
        for it, k, d, a in res["bad"]:
            print("    iteration %d  %s  max|diff| %.3e (max|value| %.3e)" % (it, k, d, a))
    print("REPEATABLE" if total == 0 else "NOT REPEATABLE: %d differences" % total)
    sys.exit(0 if total == 0 else 1)


if __name__ == "__main__":
    main()
