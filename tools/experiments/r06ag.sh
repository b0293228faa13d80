# transposed-accumulator epilogue (SEGMI_CONV_TR=1, default) against the LDS epilogue (=0): tests, per layer, in-step table, whole step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_edge_cases_gpu.py tests/test_pspnet_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv or winograd or edge or pspnet" 2>&1 | tail -3 ) > gpurun_out/r06ag_tests.txt; cat gpurun_out/r06ag_tests.txt
: > gpurun_out/r06ag_conv_bench.txt
for v in 0 1 0 1; do echo "== SEGMI_CONV_TR=$v" >> gpurun_out/r06ag_conv_bench.txt
  SEGMI_CONV_TR=$v timeout 300 python tools/conv_bench.py l4_1x1_up l4_1x1_down l3_1x1_up l3_1x1_down l3_3x3_d2 l4_3x3_d4 psp_bottleneck --op fwd 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06ag_conv_bench.txt
  SEGMI_CONV_TR=$v timeout 300 python tools/conv_bench.py l4_1x1_up l3_1x1_up l4_3x3_d4 l3_3x3_d2 --op dgrad 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06ag_conv_bench.txt
done
for i in 1 2; do for v in 0 1; do
  ( SEGMI_CONV_TR=$v timeout 300 python tools/conv_layers.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06ag_layers_tr${v}_$i.txt; echo "TR=$v $i: $(tail -1 gpurun_out/r06ag_layers_tr${v}_$i.txt)"
done; done
: > gpurun_out/r06ag_bench.txt
for i in 1 2; do for v in 0 1; do for c in cfg2 cfg3; do
 r=$(SEGMI_CONV_TR=$v timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "TR=$v $c run $i: $r" | tee -a gpurun_out/r06ag_bench.txt
done; done; done
