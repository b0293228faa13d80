# final tree, ONE box: suite, smoke, every bench line, rocprofv3 stats, PMC traffic, layer tables, HBM-bound call tables
cd $GRAFT_REPO_ROOT
TAG=r06 CFGS="cfg2 cfg3 cfg5" bash tools/gpu_round.sh suite smoke bench benchall prof pmc layers profcfg membound
( timeout 300 python tools/conv_layers.py cfg3 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_cfg3_conv_layers_f32.txt; tail -1 gpurun_out/r06_cfg3_conv_layers_f32.txt
