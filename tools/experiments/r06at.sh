# optimizer table upload: blocking (rounds 1-5) / pinned (host unbounded) / throttled (pinned, host <= 1 step ahead), alternating in one call
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r06at_bench.txt
for i in 1 2; do for c in cfg1 cfg2 cfg5 cfg3; do for up in blocking pinned throttled; do
 r=$(SEGMI_SGD_TABLE_UPLOAD=$up SEGMI_BENCH_MEMSTATS=1 timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/tmp/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "$c $up run $i: $r  $(grep memstats /tmp/err.txt | tail -1)" | tee -a gpurun_out/r06at_bench.txt
done; done; done
