# default 'auto' upload mode: tests, auto vs blocking on every config (alternating), then the bench lines + rocprofv3 stats of the final tree on this box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_graph_gpu.py tests/test_trainer_gpu.py tests/test_distributed_gpu.py tests/test_determinism_gpu.py tests/test_pspnet_gpu.py tests/test_unet_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 ) > gpurun_out/r06av_tests.txt; cat gpurun_out/r06av_tests.txt
: > gpurun_out/r06av_bench.txt
for i in 1 2; do for c in cfg1 cfg2 cfg3 cfg4 cfg5; do for up in blocking auto; do
 r=$(SEGMI_SGD_TABLE_UPLOAD=$up SEGMI_BENCH_MEMSTATS=1 timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/tmp/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "$c $up run $i: $r  $(grep memstats /tmp/err.txt | tail -1)" | tee -a gpurun_out/r06av_bench.txt
done; done; done
TAG=r06 bash tools/gpu_round.sh bench prof benchall
