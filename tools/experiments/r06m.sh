# workgroup roles per CU (SEGMI_CONV_PRIO): 0 off, 1 / 3 = s_setprio level of the leading workgroup
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv or winograd or bn_stats" 2>&1 | tail -3 ) > gpurun_out/r06m_tests.txt; cat gpurun_out/r06m_tests.txt
: > gpurun_out/r06m_conv_bench.txt
for v in 0 1 3 0 1 3; do echo "== SEGMI_CONV_PRIO=$v" >> gpurun_out/r06m_conv_bench.txt
  SEGMI_CONV_PRIO=$v timeout 300 python tools/conv_bench.py l4_1x1_up l4_1x1_down l3_1x1_up l3_1x1_down l3_3x3_d2 l4_3x3_d4 stem3 l1_1x1 --op fwd 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06m_conv_bench.txt
  SEGMI_CONV_PRIO=$v timeout 300 python tools/conv_bench.py l4_1x1_up l3_1x1_up l3_3x3_d2 --op dgrad 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06m_conv_bench.txt
done
: > gpurun_out/r06m_bench.txt
for i in 1 2; do for v in 0 1 3; do for c in cfg2 cfg3; do
 r=$(SEGMI_CONV_PRIO=$v timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "prio=$v $c run $i: $r" | tee -a gpurun_out/r06m_bench.txt
done; done; done
