# 'auto' after moving the one-time costs (pinned slots, decision) to the first three steps: repeated whole-step runs, auto vs blocking; the full default bench line twice
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r06aw_bench.txt
( timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_trainer_gpu.py tests/test_graph_gpu.py -m gpu -q -x -p no:cacheprovider -k "sgd or SGD or trainer or graph" 2>&1 | tail -2 ) > gpurun_out/r06aw_tests.txt; cat gpurun_out/r06aw_tests.txt
for i in 1 2 3; do for c in cfg2 cfg4 cfg1; do for up in blocking auto; do
 r=$(SEGMI_SGD_TABLE_UPLOAD=$up timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "$c $up run $i: $r" | tee -a gpurun_out/r06aw_bench.txt
done; done; done
for i in 1 2; do
 r=$(timeout 600 python bench.py --cpu-cap 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_us'])")
 echo "default bench line (cpu-cap 10) run $i: $r" | tee -a gpurun_out/r06aw_bench.txt
done
