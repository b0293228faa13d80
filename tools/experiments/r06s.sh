# solo workgroup (16), no traffic (2), no epilogue (4) = 22; then without the DMA pieces (32) / without the ds_reads (64) / both
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r06s_loop.txt
for v in 22 54 86 118 6 38 70 102; do echo "== SEGMI_CONV_DBG=$v" >> gpurun_out/r06s_loop.txt
  SEGMI_CONV_DBG=$v timeout 300 python tools/conv_bench.py l4_1x1_down l4_1x1_up --op fwd 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06s_loop.txt
done
cat gpurun_out/r06s_loop.txt
