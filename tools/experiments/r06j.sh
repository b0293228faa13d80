#!/bin/bash
# round 6, call j: Lovasz count / emit / backward with the pixel's row in registers (loads up front) now that the keep test needs no exp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for m in 0 1 2 4 5 7; do
  echo "SEGMI_LOVASZ_ROWREGS=$m" | tee -a gpurun_out/r06j_lovasz_rowregs.txt
  ( SEGMI_LOVASZ_ROWREGS=$m timeout 300 python tools/lovasz_bench.py --modes random trained --prune 1 --iters 10 2>&1 | grep -v amdgpu.ids ) | tee -a gpurun_out/r06j_lovasz_rowregs.txt
done
( SEGMI_LOVASZ_ROWREGS=7 timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "lovasz" 2>&1 | tail -3 ) | tee -a gpurun_out/r06j_lovasz_rowregs.txt
