#!/usr/bin/env python
""""Top kernels, achieved vs bound" tables from bench.py JSON lines (one per config): the convolution variants (MFMA-bound, executed
TFLOP/s against the 157.3 TFLOP/s fp32 matrix peak) and the HBM-bound C-ABI calls (algorithmic TB/s against 8 TB/s) of the
instrumented step, ranked together by their share of the step.

    python tools/top_kernels.py gpurun_out/r4f_bench_cfg5.log [more.json ...] > profiles/r04_top_kernels.txt
"""
import json
import sys

for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    r = d["roofline"]
    rows = []
    for name, v in r["variants"].items():
        ms = v["launches"] * v["avg_us"] / 1e3
        bound = "HBM (transforms)" if v["tflops"] == 0 else "MFMA fp32 157.3 TF/s"
        ach = "-" if v["tflops"] == 0 else "%.1f TF/s (%.0f %%)" % (v["tflops"], 100 * v["tflops"] / 157.3)
        rows.append((ms, name, v["launches"], bound, ach))
    for h in r["hbm_bound_calls"]["top"]:
        rows.append((h["ms_per_step"], h["call"], h["launches"], "HBM 8 TB/s", "%.2f TB/s (%.0f %%)" % (h["tbs"], 100 * h["frac"])))
    rows.sort(reverse=True)
    step = d["ms_per_step"]
    print("== %s   %.2f img/s, %.2f ms/step; conv launches %.2f ms (executed %.1f TF/s), HBM-bound calls %.2f ms (%.2f TB/s algorithmic)"
          % (d["config"]["workload"].split(":")[0], d["value"], step, r["all_conv"]["ms_per_step"], r["all_conv"]["achieved"],
             r["hbm_bound_calls"]["ms_per_step"], r["hbm_bound_calls"]["achieved"] or 0))
    print("   %-66s %4s %8s %6s  %-22s %s" % ("kernel / C-ABI call (in-order instrumented step)", "n", "ms", "%step", "bound", "achieved"))
    for ms, name, n, bound, ach in rows[:12]:
        print("   %-66s %4d %8.3f %6.1f  %-22s %s" % (name[:66], n, ms, 100 * ms / step, bound, ach))
    print()
