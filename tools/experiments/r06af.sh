# wave priority by PHASE: s_setprio 1 for the K loop, 0 for prologue / epilogue (SEGMI_CONV_DBG=64) against the default
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r06af_phase_prio.txt
for v in 0 64 0 64; do echo "== SEGMI_CONV_DBG=$v" >> gpurun_out/r06af_phase_prio.txt
  SEGMI_CONV_DBG=$v timeout 300 python tools/conv_bench.py l4_1x1_up l4_1x1_down l3_1x1_up l3_1x1_down l4_3x3_d4 stem3 --op fwd 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06af_phase_prio.txt
  SEGMI_CONV_DBG=$v timeout 300 python tools/conv_bench.py l4_1x1_up l3_1x1_up --op dgrad 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06af_phase_prio.txt
done
for i in 1 2; do for v in 0 64; do for c in cfg2 cfg3; do
 r=$(SEGMI_CONV_DBG=$v timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "DBG=$v $c run $i: $r" | tee -a gpurun_out/r06af_phase_prio.txt
done; done; done
