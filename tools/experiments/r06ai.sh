cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r06ai_quarter_layers.txt
for v in 0 75 0 75; do echo "== SEGMI_CONV_QUARTER=$v" >> gpurun_out/r06ai_quarter_layers.txt
  SEGMI_CONV_QUARTER=$v timeout 300 python tools/conv_bench.py r101_l3_down r101_l3_up r101_l4_down --op fwd 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06ai_quarter_layers.txt
  SEGMI_CONV_QUARTER=$v timeout 300 python tools/conv_bench.py r101_l3_down r101_l3_up r101_l4_down --op dgrad 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06ai_quarter_layers.txt
done
cat gpurun_out/r06ai_quarter_layers.txt
