#!/usr/bin/env python
"""HBM roofline of the memory-bound kernels of one training step (GPU box): every HBM-bound C-ABI call is timed with HIP
events (segmi.profile.KernelTimer(membound=True)) and priced with its ALGORITHMIC bytes — each tensor it must read or write
crosses HBM once — against the 8 TB/s HBM3E peak of MI355X_MICROARCH.md (~6.3 TB/s is what streaming kernels reach; the
Infinity Cache can push a consumer that runs right after its producer above that).

    python tools/membound_ops.py [cfg2|cfg4|...] > gpurun_out/membound_ops.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-segmentation_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from segmi import ops as _ops  # noqa: E402
from segmi.profile import KernelTimer  # noqa: E402
_ops.set_wgrad_stream(False)      # per-launch durations: every launch in order on ONE stream
import utils.losses as losses_mod  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dev = torch.device("cuda:0")
arch, _, _, _, _, _, _, loss_name, ign = bench.CONFIGS[cfg]
model = bench.build_model(cfg, dev)
from segmi.optim import SGD  # noqa: E402
opt = SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
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
with KernelTimer(membound=True) as kt:
    step()
summ = kt.summary()
PEAK = 8.0e12
mem = {k: v for k, v in summ.items() if k.startswith("segmi_")}
conv_ms = sum(v["total_ms"] for k, v in summ.items() if not k.startswith("segmi_"))
tot_ms, tot_b = sum(v["total_ms"] for v in mem.values()), sum(v["bytes"] for v in mem.values())
# calls whose operands are small and L2-resident (gathers over the low-resolution logits / the pyramid GEMM results, exp-heavy): their
# algorithmic HBM bytes are a few MB, so the HBM column says nothing about them — they are latency / ALU bound
L2_BOUND = ("segmi_upsample_ce_fwd", "segmi_upsample_ce_bwd", "segmi_pyramid_up_fwd", "segmi_pyramid_up_bwd", "segmi_copy_rows", "segmi_nchw_to_nhwc")
print("%-28s %5s %9s %10s %8s %7s  %s" % ("C-ABI call", "n", "ms/step", "GB/step", "TB/s", "% peak", "bound"))
for k, v in sorted(mem.items(), key=lambda kv: -kv[1]["total_ms"]):
    print("%-28s %5d %9.3f %10.3f %8.2f %7.1f  %s" % (k, v["launches"], v["total_ms"], v["bytes"] / 1e9, v["bytes"] / v["total_ms"] / 1e9,
                                                   100 * v["bytes"] / (v["total_ms"] * 1e-3) / PEAK, "L2 / ALU" if k in L2_BOUND else "HBM"))
print("%-28s %5d %9.3f %10.3f %8.2f %7.1f" % ("all of the above", sum(v["launches"] for v in mem.values()), tot_ms, tot_b / 1e9,
                                           tot_b / tot_ms / 1e9, 100 * tot_b / (tot_ms * 1e-3) / PEAK))
print("(%s, one training step; convolution launches of the same step: %.2f ms; peak = 8 TB/s HBM3E; bytes = algorithmic: every operand "
      "tensor once; event pairs around each call add ~2 us of idle time to calls shorter than ~10 us)" % (cfg, conv_ms))
