# final tree (SGD table upload auto): bench lines + rocprofv3 stats + layer tables on ONE box
cd $GRAFT_REPO_ROOT
TAG=r06 bash tools/gpu_round.sh bench prof benchall layers
( timeout 300 python tools/conv_layers.py cfg3 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_cfg3_conv_layers_f32.txt; tail -1 gpurun_out/r06_cfg3_conv_layers_f32.txt
