# does the pixel stride of the activation operand matter?  (rows 2 KB / 8 KB apart: same L2 channel for a whole chunk column?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r06p_ldpad.txt
for v in 0 32 16 64 0 32; do echo "== ldpad=$v" >> gpurun_out/r06p_ldpad.txt
  timeout 300 python tools/conv_bench.py l4_1x1_up l4_1x1_down l3_1x1_up l3_1x1_down stem3 --op fwd --ldpad $v 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06p_ldpad.txt
done
cat gpurun_out/r06p_ldpad.txt
