# optimizer table: blocking / throttled (pinned staging + copy) / mapped (kernel reads the pinned slot, no copy command), alternating in one call
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r06au_bench.txt
for i in 1 2; do for c in cfg1 cfg2 cfg5; do for up in blocking throttled mapped; do
 r=$(SEGMI_SGD_TABLE_UPLOAD=$up SEGMI_BENCH_MEMSTATS=1 timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/tmp/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "$c $up run $i: $r  $(grep memstats /tmp/err.txt | tail -1)" | tee -a gpurun_out/r06au_bench.txt
done; done; done
( SEGMI_SGD_TABLE_UPLOAD=mapped timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_trainer_gpu.py -m gpu -q -x -p no:cacheprovider -k "sgd or SGD or trainer" 2>&1 | tail -3 ) > gpurun_out/r06au_tests_mapped.txt; cat gpurun_out/r06au_tests_mapped.txt
