# UNet frozen-BN gradient test under the HEAD library (64x64 tiles) and the tap-skip library; then the GPU suite on the tap-skip library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=pytorch-segmentation_amd/segmi/libsegmi.so
cp $L /tmp/new.so; cp tools/experiments/libsegmi_prev.so /tmp/prev.so
: > gpurun_out/r06al_unet.txt
for v in prev new; do cp /tmp/$v.so $L; echo "== $v" >> gpurun_out/r06al_unet.txt
 ( timeout 600 python -m pytest tests/test_unet_gpu.py -m gpu -q -s -p no:cacheprovider -k frozen 2>&1 | grep -v amdgpu.ids | tail -12 ) >> gpurun_out/r06al_unet.txt
done
cat gpurun_out/r06al_unet.txt
cp /tmp/new.so $L
( timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06al_gpu_suite.txt; tail -5 gpurun_out/r06al_gpu_suite.txt
