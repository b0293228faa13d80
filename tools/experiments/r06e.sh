#!/bin/bash
# round 6, call e: per-kernel split of the fused / unfused upsample + Lovasz tail (rocprofv3), BN backward block count
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "lovasz" 2>&1 | tail -5 ) > gpurun_out/r06e_lovasz_tests.txt; tail -2 gpurun_out/r06e_lovasz_tests.txt
rm -rf gpurun_out/prof
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python tools/lovasz_bench.py --up 4 --modes random --iters 10 2>&1 | tail -3 ) > gpurun_out/r06e_prof.log
find gpurun_out/prof -name "*kernel_trace*" -delete
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06e_lovasz_up_kernel_stats.csv
rm -rf gpurun_out/prof
head -30 gpurun_out/r06e_lovasz_up_kernel_stats.csv | cut -c1-160
for v in 1024 1536 3072; do TAG=r06e_b$v bash tools/ab.sh cfg2 SEGMI_BN_BWD_BLOCKS 2048 $v 1; done
