# n-tile group width of the tile order (dbg bits 8.. = GN): 8 (default) / 4 / 2 / 16 / 1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r06q_gn.txt
for v in 0 1024 512 4096 256 0; do echo "== SEGMI_CONV_DBG=$v (GN=$((v/256)))" >> gpurun_out/r06q_gn.txt
  SEGMI_CONV_DBG=$v timeout 300 python tools/conv_bench.py l4_1x1_up l4_1x1_down l3_1x1_up stem3 --op fwd 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06q_gn.txt
done
cat gpurun_out/r06q_gn.txt
