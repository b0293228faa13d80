# final tree: new property / conv cases, then the bench lines again (executed-FLOP accounting of launches that skip taps), cfg2 / cfg3 layer tables
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_fullsize_properties_gpu.py tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv" 2>&1 | tail -3 ) > gpurun_out/r06an_tests.txt; cat gpurun_out/r06an_tests.txt
TAG=r06 bash tools/gpu_round.sh bench benchall layers
( timeout 300 python tools/conv_layers.py cfg3 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_cfg3_conv_layers_f32.txt; tail -1 gpurun_out/r06_cfg3_conv_layers_f32.txt
