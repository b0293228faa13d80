# launch-paced configs (cfg5: 840 launches per step, cfg1: 7 ms steps) eager vs replayed from a hipGraph (bench.py --graph), alternating in one call; cfg2 for reference
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r06aq_graph.txt
for i in 1 2; do for c in cfg5 cfg1 cfg2; do for g in "" "--graph"; do
 r=$(timeout 500 python bench.py --config $c --no-cpu --no-alt --no-roofline $g 2>/tmp/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['hip_graph'])" 2>&1 | tail -1)
 echo "$c ${g:-eager} run $i: $r" | tee -a gpurun_out/r06aq_graph.txt
 grep -i "error\|Traceback" /tmp/err.txt | head -3 >> gpurun_out/r06aq_graph.txt
done; done; done
