cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r06aj_quarter.txt
for i in 1 2; do for c in cfg2 cfg1; do for v in 0 75; do
 r=$(SEGMI_CONV_QUARTER=$v timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "$c QUARTER=$v run $i: $r" | tee -a gpurun_out/r06aj_quarter.txt
done; done; done
