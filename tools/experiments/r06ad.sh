# statistics epilogue with constant counts on full tiles: tests, in-step per-layer table and whole step, prev = HEAD library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=pytorch-segmentation_amd/segmi/libsegmi.so
cp $L /tmp/new.so; cp tools/experiments/libsegmi_prev.so /tmp/prev.so
( timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_pspnet_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv or stats or pspnet or bn" 2>&1 | tail -3 ) > gpurun_out/r06ad_tests.txt; cat gpurun_out/r06ad_tests.txt
for i in 1 2; do for v in prev new; do cp /tmp/$v.so $L
  ( timeout 300 python tools/conv_layers.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06ad_layers_${v}_$i.txt; echo "$v $i: $(tail -1 gpurun_out/r06ad_layers_${v}_$i.txt)"
done; done
: > gpurun_out/r06ad_bench.txt
for i in 1 2 3; do for v in prev new; do cp /tmp/$v.so $L; for c in cfg2; do
 r=$(timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "$v $c run $i: $r" | tee -a gpurun_out/r06ad_bench.txt
done; done; done
cp /tmp/new.so $L
