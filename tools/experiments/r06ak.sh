# whole-tile tap skipping in the tap-mask kernels: tests, ASPP layers, cfg3 / cfg2 / cfg1 whole step; prev = HEAD library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=pytorch-segmentation_amd/segmi/libsegmi.so
cp $L /tmp/new.so; cp tools/experiments/libsegmi_prev.so /tmp/prev.so
( timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_edge_cases_gpu.py tests/test_deeplab_gpu.py tests/test_unet_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 ) > gpurun_out/r06ak_tests.txt; cat gpurun_out/r06ak_tests.txt
: > gpurun_out/r06ak_conv_bench.txt
for v in prev new prev new; do cp /tmp/$v.so $L; echo "== $v" >> gpurun_out/r06ak_conv_bench.txt
  timeout 300 python tools/conv_bench.py aspp_d6 aspp_d12 aspp_d18 stem3 --op fwd 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06ak_conv_bench.txt
  timeout 300 python tools/conv_bench.py aspp_d6 aspp_d12 aspp_d18 stem3 --op dgrad 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06ak_conv_bench.txt
done
: > gpurun_out/r06ak_bench.txt
for i in 1 2; do for v in prev new; do cp /tmp/$v.so $L; for c in cfg3 cfg2 cfg1; do
 r=$(timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "$v $c run $i: $r" | tee -a gpurun_out/r06ak_bench.txt
done; done; done
cp /tmp/new.so $L
