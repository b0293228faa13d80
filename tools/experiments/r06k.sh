cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_deeplab_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 ) > gpurun_out/r06k_tests.txt; cat gpurun_out/r06k_tests.txt
: > gpurun_out/r06k_bench.txt
for i in 1 2; do for c in cfg2 cfg5 cfg3; do
 r=$(timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "$c run $i: $r" | tee -a gpurun_out/r06k_bench.txt
done; done
