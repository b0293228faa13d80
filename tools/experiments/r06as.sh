# optimizer chunk table through pinned staging (no host block at the end of backward): tests, GPU idle time, whole steps A/B (SEGMI_SGD_TABLE_UPLOAD=blocking = before)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_graph_gpu.py tests/test_trainer_gpu.py tests/test_distributed_gpu.py tests/test_determinism_gpu.py -m gpu -q -x -p no:cacheprovider -k "sgd or SGD or graph or trainer or bucket or rank or determin or repeatable" 2>&1 | tail -4 ) > gpurun_out/r06as_tests.txt; cat gpurun_out/r06as_tests.txt
: > gpurun_out/r06_gpu_gaps.txt
for c in cfg2 cfg5 cfg1; do for up in blocking pinned; do
  rm -rf /tmp/tr; SEGMI_SGD_TABLE_UPLOAD=$up timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o r -- python bench.py --config $c --steps 7 --warmup 2 --no-cpu --no-roofline --no-alt > /tmp/tr.full 2>&1
  f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
  { echo "=== $c, SEGMI_SGD_TABLE_UPLOAD=$up (under rocprofv3 --kernel-trace): $(grep '^{' /tmp/tr.full | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms')")"; python tools/gpu_gaps.py $f | head -24; } >> gpurun_out/r06_gpu_gaps.txt 2>&1
done; done
grep "===\|steps," gpurun_out/r06_gpu_gaps.txt | cut -c1-200
: > gpurun_out/r06as_bench.txt
for i in 1 2; do for c in cfg2 cfg3 cfg5 cfg1 cfg4; do for up in blocking pinned; do
 r=$(SEGMI_SGD_TABLE_UPLOAD=$up timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "$c $up run $i: $r" | tee -a gpurun_out/r06as_bench.txt
done; done; done
