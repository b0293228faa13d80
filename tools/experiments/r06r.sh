# one workgroup per CU (dbg 16) with / without operand traffic (2) and epilogue (4): how fast is a workgroup ALONE on its CU?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r06r_solo.txt
for v in 0 16 18 22 24 0; do echo "== SEGMI_CONV_DBG=$v" >> gpurun_out/r06r_solo.txt
  SEGMI_CONV_DBG=$v timeout 300 python tools/conv_bench.py l4_1x1_up l4_1x1_down l3_1x1_up stem3 --op fwd 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06r_solo.txt
done
cat gpurun_out/r06r_solo.txt
