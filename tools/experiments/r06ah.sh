# 64 x 64 tiles for launches that under-fill the chip with 64 x 128 tiles: SEGMI_CONV_QUARTER = fill threshold in % of 768 slots (0 off)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv2d_fwd_dgrad or stats_epilogue" 2>&1 | tail -2 ) > gpurun_out/r06ah_tests.txt; cat gpurun_out/r06ah_tests.txt
( SEGMI_CONV_QUARTER=100 timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_deeplab_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv2d_fwd_dgrad or stats_epilogue or deeplab" 2>&1 | tail -2 ) >> gpurun_out/r06ah_tests.txt; tail -2 gpurun_out/r06ah_tests.txt
: > gpurun_out/r06ah_quarter.txt
for i in 1 2; do for c in cfg3 cfg5 cfg4; do for v in 0 75 100; do
 r=$(SEGMI_CONV_QUARTER=$v timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "$c QUARTER=$v run $i: $r" | tee -a gpurun_out/r06ah_quarter.txt
done; done; done
