#!/usr/bin/env python
"""GPU idle time inside the training steps of a rocprofv3 kernel trace (GPU box).

    rocprofv3 --kernel-trace --output-format csv -d DIR -o r -- python bench.py --steps 7 --warmup 2 --no-cpu --no-roofline --no-alt
    python tools/gpu_gaps.py DIR/**/r_kernel_trace.csv [steps=9] > profiles/rNN_cfg2_gpu_gaps.txt

Takes the union of the kernels' [start, end) intervals (all queues), drops everything before the first / after the last `sgd_multi_kernel`
bracketing the steady-state steps, and prints busy / idle time per step, the idle time by gap size, and the largest gaps with
the kernels on either side.
"""
import csv
import sys

path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]))
rows.sort()
sgd = [i for i, r in enumerate(rows) if "sgd_multi_kernel" in r[2]]
assert len(sgd) >= 3, "no optimizer launches in the trace"
lo, hi = sgd[1], sgd[-1]                 # from the end of the 2nd step's optimizer to the last step's: whole steps, warm
steps = len(sgd) - 2
seg = rows[lo:hi + 1]
t_begin, t_end = seg[0][1], seg[-1][1]
gaps, cur_end, last = [], seg[0][1], seg[0][2]      # cur_end: end of the union of the intervals so far; last: the kernel that set it
for s, e, n in seg[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, last, n))
    if e > cur_end:
        cur_end, last = e, n
wall = t_end - t_begin
idle = sum(g[0] for g in gaps)
print("%d steps, %.2f ms per step wall; GPU busy %.2f ms (%.1f %%), idle %.2f ms per step in %d gaps per step"
      % (steps, wall / steps / 1e6, (wall - idle) / steps / 1e6, 100.0 * (wall - idle) / wall, idle / steps / 1e6, len(gaps) // steps))
for lo_us, hi_us in ((0, 5), (5, 20), (20, 100), (100, 1000), (1000, 1e9)):
    sel = [g[0] for g in gaps if lo_us * 1e3 <= g[0] < hi_us * 1e3]
    print("   gaps of %4g - %4g us: %6d per step, %.3f ms per step" % (lo_us, hi_us, len(sel) // steps, sum(sel) / steps / 1e6))
print("largest gaps (us, kernel before -> kernel after):")
for g in sorted(gaps, reverse=True)[:25]:
    print("   %8.1f  %s -> %s" % (g[0] / 1e3, g[1], g[2]))
by = {}
for g in gaps:
    k = (g[1], g[2])
    by.setdefault(k, [0, 0])
    by[k][0] += g[0]
    by[k][1] += 1
print("idle time by (kernel before -> kernel after), ms per step:")
for k, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:25]:
    print("   %7.3f  %5.1f x  %s -> %s" % (v[0] / steps / 1e6, v[1] / steps, k[0], k[1]))
