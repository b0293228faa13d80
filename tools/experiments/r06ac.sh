# in-step per-layer conv table (with BN-statistics epilogues and accumulating dgrads): prev = HEAD library vs new
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=pytorch-segmentation_amd/segmi/libsegmi.so
cp $L /tmp/new.so; cp tools/experiments/libsegmi_prev.so /tmp/prev.so
for i in 1 2; do for v in prev new; do cp /tmp/$v.so $L
  ( timeout 300 python tools/conv_layers.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06ac_layers_${v}_$i.txt; echo "$v $i: $(tail -1 gpurun_out/r06ac_layers_${v}_$i.txt)"
done; done
cp /tmp/new.so $L
