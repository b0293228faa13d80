# cost of the two-level summation (flush every 8 chunks): SEGMI_CONV_DBG=32 never flushes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r06v_flush.txt
for v in 0 32 0 32; do echo "== SEGMI_CONV_DBG=$v" >> gpurun_out/r06v_flush.txt
  SEGMI_CONV_DBG=$v timeout 300 python tools/conv_bench.py l4_1x1_down l3_1x1_down l4_3x3_d4 psp_bottleneck aux_3x3 --op fwd 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06v_flush.txt
done
cat gpurun_out/r06v_flush.txt
