#!/bin/bash
# round 6, call f: Lovasz keep test in the logit domain — tests, per-kernel split of the fused / unfused tail, cfg5 A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${TAG:-r06f}
( timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_edge_cases_gpu.py tests/test_fullsize_properties_gpu.py -m gpu -q -x -p no:cacheprovider -k "lovasz or Lovasz or loss or edge" 2>&1 | tail -12 ) > gpurun_out/${T}_lovasz_tests.txt; tail -3 gpurun_out/${T}_lovasz_tests.txt | cut -c1-400
rm -rf gpurun_out/prof
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python tools/lovasz_bench.py --up 4 --modes random --iters 10 2>&1 | tail -3 ) > gpurun_out/${T}_prof.log
find gpurun_out/prof -name "*kernel_trace*" -delete
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${T}_lovasz_up_kernel_stats.csv
rm -rf gpurun_out/prof
cat gpurun_out/${T}_prof.log
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/${T}_lovasz_up_kernel_stats.csv')))
for r in rows[:22]:
    n=r['Name'].replace('(anonymous namespace)::','')
    print(n[:64].ljust(64), r['Calls'].rjust(4), "avg %.1f us"%(float(r['AverageNs'])/1e3))
PY
( timeout 600 python tools/lovasz_bench.py --up 4 --modes random trained saturated --iters 10 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${T}_lovasz_up_bench.txt; cat gpurun_out/${T}_lovasz_up_bench.txt
TAG=$T bash tools/ab.sh cfg5 SEGMI_LOVASZ_FUSE_UP 0 1 2
