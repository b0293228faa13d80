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
from segmi import ops  # noqa: E402
from segmi.profile import KernelTimer  # noqa: E402
ops.set_wgrad_stream(False)       # per-launch durations: every launch in order on ONE stream
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
PEAK_TF = 157.3      # fp32 MFMA ceiling, TFLOP/s
PEAK_TB = 8.0                                                      # HBM3E spec (MI355X_MICROARCH.md; ~6.3 TB/s achievable)
# roofline time of a launch = max(FLOPs / matrix peak, algorithmic bytes / HBM peak); "bound" names the larger term
print("%-52s %-34s %3s %9s %7s %6s %6s %5s %6s" % ("variant", "geometry", "n", "ms", "TF/s", "TB/s", "%roof", "bound", "%step"))
for name, det, n, ms, fl, nb in rows:
    t_m, t_h = fl / (PEAK_TF * 1e12), nb / (PEAK_TB * 1e12)
    roof = max(t_m, t_h)
    print("%-52s %-34s %3d %9.3f %7.1f %6.2f %6.1f %5s %6.2f" % (name, det, n, ms, fl / ms / 1e9, nb / ms / 1e9, 100 * roof / (ms * 1e-3),
                                                              "mfma" if t_m >= t_h else "hbm", 100 * ms / tot))
print("total conv ms %.2f, %.1f TF/s (fp32 MFMA: matrix ceiling %.1f TF/s, HBM %.1f TB/s)" % (tot, sum(r[4] for r in rows) / tot / 1e9, PEAK_TF, PEAK_TB))
