#!/bin/bash
# round 6, call d: fused upsample + Lovasz (tests, loss alone, cfg5 A/B), BN row-order experiment (cfg2), cfg5 full-size audit
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "lovasz or batch_norm" 2>&1 | tail -15 ) > gpurun_out/r06d_lovasz_tests.txt; tail -3 gpurun_out/r06d_lovasz_tests.txt
( timeout 600 python tools/lovasz_bench.py --up 4 --modes random trained --iters 10 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06d_lovasz_up_bench.txt; cat gpurun_out/r06d_lovasz_up_bench.txt
TAG=r06d bash tools/ab.sh cfg5 SEGMI_LOVASZ_FUSE_UP 0 1 2
for v in 1 2 3 4 7; do TAG=r06d_rev$v bash tools/ab.sh cfg2 SEGMI_BN_REV 0 $v 1; done
( timeout 900 python -m pytest tests/test_fullsize_golden_gpu.py -m gpu -q -x -p no:cacheprovider -k "cfg5" 2>&1 | tail -8 | cut -c1-1500 ) > gpurun_out/r06d_fullsize_cfg5.txt; tail -3 gpurun_out/r06d_fullsize_cfg5.txt | cut -c1-300
