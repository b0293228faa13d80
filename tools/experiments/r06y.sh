# pointwise forms with tails (fprop: Cs % 32 != 0; wgrad: M % 32 != 0): tests + cfg5 / cfg3 / cfg1 A/B against 442b3b4's library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=pytorch-segmentation_amd/segmi/libsegmi.so
cp $L /tmp/new.so; cp tools/experiments/libsegmi_base.so /tmp/base.so
( timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_deeplab_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 ) > gpurun_out/r06y_tests.txt; cat gpurun_out/r06y_tests.txt
: > gpurun_out/r06y_bench.txt
for i in 1 2; do for v in base new; do cp /tmp/$v.so $L; for c in cfg5 cfg3 cfg1; do
 r=$(timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "$v $c run $i: $r" | tee -a gpurun_out/r06y_bench.txt
done; done; done
cp /tmp/new.so $L
