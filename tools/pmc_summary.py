#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: average counter values per launch, MfmaUtil,
effective clock (needs the matching kernel_trace.csv in the same directory for durations)."""
import collections
import csv
import os
import sys

d = sys.argv[1]
pre = sys.argv[2] if len(sys.argv) > 2 else "p"
rows = list(csv.DictReader(open(os.path.join(d, pre + "_counter_collection.csv"))))
dur = collections.defaultdict(list)
tr = os.path.join(d, pre + "_kernel_trace.csv")
if os.path.exists(tr):
    for r in csv.DictReader(open(tr)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in acc.items():
    if not any(k in n for k in sys.argv[3:] or ["conv_"]):
        continue
    short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    us = sum(dur[n]) / len(dur[n]) if dur[n] else float("nan")
    print("%s   launches=%d   avg duration %.1f us" % (short, len(next(iter(c.values()))), us))
    avg = {k: sum(v) / len(v) for k, v in c.items()}
    for k, v in sorted(avg.items()):
        print("    %-28s %.4g" % (k, v))
    if "GRBM_GUI_ACTIVE" in avg:
        g = avg["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
        print("    => effective clock %.2f GHz" % (g / us / 1e3))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
            print("    => MfmaUtil %.1f %%  (MFMA busy cycles / (1024 SIMDs x active cycles))" % (100 * avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 1024)))
        if "SQ_WAVE_CYCLES" in avg:
            wc = avg["SQ_WAVE_CYCLES"]
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if k in avg:
                    print("    => %s / SQ_WAVE_CYCLES = %.1f %%" % (k, 100 * avg[k] / wc))
