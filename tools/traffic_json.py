#!/usr/bin/env python
"""Build profiles/rNN_cfg2_conv_traffic_<math>.json (bench.py looks the dominant kernel up in it BY NAME) from the two rocprofv3 PMC passes of `tools/gpu_round.sh pmc`
(gpurun_out/pmc/{fetch,write}/r_counter_collection.csv): HBM/MALL bytes of the conv implicit-GEMM launches of ONE cfg2 step.

FETCH_SIZE / WRITE_SIZE are reported in KB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-counts
16 B/lane streaming reads by 2x; WRITE_SIZE is used as reported.  bench.py was run with --steps 1 --warmup 1, so every
kernel's counters are summed over two steps and halved.
"""
import collections
import csv
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/r02_cfg2_conv_traffic_f32.json"
STEPS = 2.0
CONV = ("conv_dma_kernel", "conv_wgrad_dma_kernel", "splitk_reduce_kernel", "conv_gather_kernel", "conv_wgrad_kernel",
        "wino_input_kernel", "wino_output_kernel", "wino_filter_kernel", "wino_dy_kernel", "wino_filter_grad_kernel")


def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


launches = collections.Counter()


def load(sub, counter):
    per, total = collections.Counter(), 0.0
    for r in csv.DictReader(open(os.path.join(root, sub, "r_counter_collection.csv"))):
        if r["Counter_Name"] != counter:
            continue
        v = float(r["Counter_Value"]) / STEPS
        total += v
        per[short(r["Kernel_Name"])] += v
        if counter == "FETCH_SIZE":
            launches[short(r["Kernel_Name"])] += 1.0 / STEPS
    return per, total


fetch, fetch_all = load("fetch", "FETCH_SIZE")
write, write_all = load("write", "WRITE_SIZE")
cf = sum(v for k, v in fetch.items() if k.startswith(CONV)) * 1024.0
cw = sum(v for k, v in write.items() if k.startswith(CONV)) * 1024.0
doc = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python bench.py --steps 1 "
              "--warmup 1 --no-cpu --no-roofline; MI355X (tools/gpu_round.sh pmc + tools/traffic_json.py)",
    "scope": "all conv launches (conv_dma_kernel*, conv_wgrad_dma_kernel*, splitk_reduce_kernel, the Winograd transforms wino_*) of ONE cfg2 training step",
    "conv_winograd": any(k.startswith("wino_") for k in fetch),
    "fetch_bytes_raw": cf,
    "fetch_bytes_corrected": 2 * cf,
    "write_bytes": cw,
    "correction": "FETCH_SIZE x2 on gfx950 for 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
    "traffic_bytes_per_step": 2 * cf + cw,
    "all_kernels_fetch_bytes_raw": fetch_all * 1024.0,
    "all_kernels_write_bytes": write_all * 1024.0,
    "per_kernel": {k: {"launches_per_step": launches[k], "fetch_bytes_corrected": 2 * fetch[k] * 1024.0, "write_bytes": write[k] * 1024.0,
                       "traffic_bytes_per_launch": (2 * fetch[k] + write[k]) * 1024.0 / launches[k]}
                   for k in fetch if k.startswith(CONV) and launches[k] > 0},
    "by_kernel_fetch_kb_raw": {k: int(v) for k, v in fetch.most_common(12) if k.startswith(CONV)},
    "by_kernel_write_kb": {k: int(v) for k, v in write.most_common(12) if k.startswith(CONV)},
    "notes": ["fetch counts Infinity-Cache (MALL) hits as well as HBM reads"],
}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("fetch_bytes_corrected", "write_bytes", "traffic_bytes_per_step")}))
