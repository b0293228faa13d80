# (1) the 8-rank one-GPU launcher test five times with full stderr; (2) new conv cases; (3) wgrad row-chunk skipping A/B on the ASPP layers and on cfg3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r06am_launcher.txt
for i in 1 2 3 4 5; do
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python bench.py --gpus 8 --config cfg1 --steps 3 --warmup 1 --no-cpu --no-roofline --no-alt --sync-bn > /tmp/l$i.out 2> /tmp/l$i.err; rc=$?
  echo "run $i rc $rc: $(grep '^{' /tmp/l$i.out | cut -c1-120)" >> gpurun_out/r06am_launcher.txt
  if [ $rc -ne 0 ]; then grep -v "amdgpu.ids" /tmp/l$i.err | grep -iv "warn" | head -150 >> gpurun_out/r06am_launcher.txt; fi
done
cat gpurun_out/r06am_launcher.txt | cut -c1-200 | head -60
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv2d" 2>&1 | tail -5 ) > gpurun_out/r06am_tests.txt; cat gpurun_out/r06am_tests.txt
: > gpurun_out/r06am_wgrad.txt
for i in 1 2; do for v in 0 1; do echo "== SEGMI_WGRAD_SKIPROWS=$v" >> gpurun_out/r06am_wgrad.txt
  SEGMI_WGRAD_SKIPROWS=$v timeout 300 python tools/conv_bench.py aspp_d6 aspp_d12 aspp_d18 --op wgrad 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06am_wgrad.txt
done; done
for i in 1 2; do for v in 0 1; do
 r=$(SEGMI_WGRAD_SKIPROWS=$v timeout 400 python bench.py --config cfg3 --no-cpu --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "skiprows=$v cfg3 run $i: $r" | tee -a gpurun_out/r06am_wgrad.txt
done; done
cat gpurun_out/r06am_wgrad.txt
