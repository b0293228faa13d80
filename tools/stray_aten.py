"""Which ATen device kernels still run inside one cfg2 training step, and which Python line launches them.

The product path is libsegmi.so; what torch itself launches in a step (autograd's gradient accumulation `add`, `zeros`
materialised for unused outputs, `clone`s made by AccumulateGrad, stray `copy_`s) is overhead the round-1 review listed
(812 launches per step, 28 fills, 44 copies, 6 adds).  This prints, for one steady-state step, every ATen operator that
launched a device kernel with its count and the innermost repository frames of its call sites.

    python tools/stray_aten.py [--config cfg2]
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-segmentation_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    args = ap.parse_args()
    import bench
    import utils.losses as losses_mod
    from segmi.optim import SGD
    arch, kw, classes, n, h, w, _, loss_name, ign = bench.CONFIGS[args.config]
    dev = torch.device("cuda:0")
    model = bench.build_model(args.config, dev)
    opt = SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    crit = getattr(losses_mod, loss_name)(ignore_index=ign)
    x, t = bench.synth_batch(args.config, dev, 0)
    psp = arch[:3] == "PSP"

    def step():
        opt.zero_grad(set_to_none=True)
        if psp:
            out, aux = model(x)
            loss = crit(out, t) + 0.4 * crit(aux, t)
        else:
            loss = crit(model(x), t)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    sites = collections.defaultdict(collections.Counter)
    kernels = collections.Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            kernels[ev.name[:100]] += 1
            continue
        if not ev.name.startswith("aten::") or not ev.kernels:
            continue
        if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::") and ev.cpu_parent.kernels:
            continue                                  # count the outermost ATen op that owns the kernel
        frames = [f for f in (ev.stack or []) if ROOT in f and "stray_aten" not in f][:2] or ["(autograd engine / no repository frame)"]
        sites[ev.name][" <- ".join(f.replace(ROOT + "/", "") for f in frames)] += 1
    print("device kernels in one step: %d" % sum(kernels.values()))
    print("ATen operators that launched device kernels:")
    for name, c in sorted(sites.items(), key=lambda kv: -sum(kv[1].values())):
        print("  %-28s %4d" % (name, sum(c.values())))
        for site, k in c.most_common(6):
            print("      %4d  %s" % (k, site))


if __name__ == "__main__":
    main()
