# ablations of conv_dma_kernel (SEGMI_CONV_DBG bits: 1 no epilogue stores, 2 no operand traffic after the first chunk, 4 no epilogue at all)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r06o2_ablation.txt
for v in 0 8 12 2 0 8; do echo "== SEGMI_CONV_DBG=$v" >> gpurun_out/r06o2_ablation.txt
  SEGMI_CONV_DBG=$v timeout 300 python tools/conv_bench.py l4_1x1_up l4_1x1_down l3_1x1_up l3_1x1_down stem3 --op fwd 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06o2_ablation.txt
done
cat gpurun_out/r06o2_ablation.txt
