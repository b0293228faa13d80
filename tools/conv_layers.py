#!/usr/bin/env python
"""Per-layer table of the conv implicit-GEMM launches of one training step (GPU box):
variant, geometry, launches, total ms, TFLOP/s — the worklist for kernel tuning.

    python tools/conv_layers.py [cfg2|cfg4] > gpurun_out/conv_layers.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-segmentation_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from segmi.profile import KernelTimer  # noqa: E402
import utils.losses as losses_mod  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dev = torch.device("cuda:0")
arch, _, _, _, _, _, _, loss_name, ign = bench.CONFIGS[cfg]
model = bench.build_model(cfg, dev)
opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
crit = getattr(losses_mod, loss_name)(ignore_index=ign)
x, t = bench.synth_batch(cfg, dev, 0)


def step():
    opt.zero_grad(set_to_none=True)
    if arch[:3] == "PSP":
        out, aux = model(x)
        loss = crit(out, t) + 0.4 * crit(aux, t)
    else:
        loss = crit(model(x), t)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
with KernelTimer() as kt:
    step()
rows = kt.by_detail()
tot = sum(r[3] for r in rows)
print("%-52s %-34s %3s %9s %7s %6s" % ("variant", "geometry", "n", "ms", "TF/s", "%"))
for name, det, n, ms, fl in rows:
    print("%-52s %-34s %3d %9.3f %7.1f %6.2f" % (name, det, n, ms, fl / ms / 1e9, 100 * ms / tot))
print("total conv ms %.2f, %.1f TF/s" % (tot, sum(r[4] for r in rows) / tot / 1e9))
