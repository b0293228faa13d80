#!/bin/bash
# round 6, call h: dense thread geometry (cx = C/4 when that is not a power of two) for the bilinear kernels and the fused CE backward
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r06i
( timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_edge_cases_gpu.py tests/test_deeplab_gpu.py tests/test_pspnet_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6 ) > gpurun_out/${T}_tests.txt; tail -2 gpurun_out/${T}_tests.txt | cut -c1-300
CFGS="cfg5 cfg2 cfg3" TAG=$T bash tools/gpu_round.sh membound
for c in cfg5 cfg3 cfg2; do
  for i in 1 2; do
    r=$(timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$c run $i: $r" | tee -a gpurun_out/${T}_bench.txt
  done
done
