cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=pytorch-segmentation_amd/segmi/libsegmi.so
cp $L /tmp/new.so; cp tools/experiments/libsegmi_base.so /tmp/base.so
: > gpurun_out/r06w_conv_bench.txt
for v in base new base new; do cp /tmp/$v.so $L; echo "== $v" >> gpurun_out/r06w_conv_bench.txt
  timeout 300 python tools/conv_bench.py l4_1x1_up l4_1x1_down l3_1x1_down l4_3x3_d4 --op fwd 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06w_conv_bench.txt
done
cp /tmp/new.so $L
cat gpurun_out/r06w_conv_bench.txt; rocm-smi --showclocks 2>/dev/null | head -20
