# epilogue: buffer stores (no 64-bit address arithmetic, no exec-masked store blocks), bias / accumulate / final add as real branches: prev = HEAD library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=pytorch-segmentation_amd/segmi/libsegmi.so
cp $L /tmp/new.so; cp tools/experiments/libsegmi_prev.so /tmp/prev.so
( timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv or winograd or edge" 2>&1 | tail -3 ) > gpurun_out/r06ab_tests.txt; cat gpurun_out/r06ab_tests.txt
: > gpurun_out/r06ab_conv_bench.txt
for v in prev new prev new; do cp /tmp/$v.so $L; echo "== $v" >> gpurun_out/r06ab_conv_bench.txt
  timeout 300 python tools/conv_bench.py l4_1x1_up l4_1x1_down l3_1x1_up l3_1x1_down l3_3x3_d2 l4_3x3_d4 stem3 --op fwd 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06ab_conv_bench.txt
  timeout 300 python tools/conv_bench.py l4_1x1_up l3_1x1_up l4_3x3_d4 stem3 --op dgrad 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06ab_conv_bench.txt
done
: > gpurun_out/r06ab_bench.txt
for i in 1 2; do for v in prev new; do cp /tmp/$v.so $L; for c in cfg2 cfg3; do
 r=$(timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "$v $c run $i: $r" | tee -a gpurun_out/r06ab_bench.txt
done; done; done
cp /tmp/new.so $L
