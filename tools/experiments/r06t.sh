# pieces: address arithmetic only (128) vs nothing (32) vs everything, solo (16) and two workgroups per CU; no traffic (2), no epilogue (4)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r06t_pieces.txt
for v in 22 150 54 6 134 38 22 150; do echo "== SEGMI_CONV_DBG=$v" >> gpurun_out/r06t_pieces.txt
  SEGMI_CONV_DBG=$v timeout 300 python tools/conv_bench.py l4_1x1_down l4_1x1_up --op fwd 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06t_pieces.txt
done
cat gpurun_out/r06t_pieces.txt
