#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, per-layer conv table, rocprof kernel stats.
# Outputs under gpurun_out/.   usage: tools/gpu_round.sh [tests] [bench] [layers] [prof] [pmc]
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
what="${*:-tests bench layers prof}"
for w in $what; do
case $w in
tests)
  ( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/pytest_gpu.log
  ( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) > gpurun_out/smoke.log
  cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log ;;
bench)
  ( timeout 600 python bench.py 2>&1 | tail -1 ) > gpurun_out/bench.log; cat gpurun_out/bench.log ;;
benchfast)
  ( timeout 600 python bench.py --no-cpu 2>&1 | tail -1 ) > gpurun_out/bench.log; cat gpurun_out/bench.log ;;
layers)
  ( timeout 300 python tools/conv_layers.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/conv_layers.txt; tail -3 gpurun_out/conv_layers.txt ;;
prof)
  rm -rf gpurun_out/prof
  ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python bench.py --steps 5 --warmup 2 --no-cpu --no-roofline 2>&1 | tail -2 ) > gpurun_out/prof.log
  find gpurun_out/prof -name "*kernel_trace*" -size +30M -delete
  find gpurun_out/prof -type f | head ;;
pmc)
  rm -rf gpurun_out/pmc
  ( timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc/fetch -o r -- python bench.py --steps 1 --warmup 1 --no-cpu --no-roofline 2>&1 | tail -2 ) > gpurun_out/pmc.log
  ( timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc/write -o r -- python bench.py --steps 1 --warmup 1 --no-cpu --no-roofline 2>&1 | tail -2 ) >> gpurun_out/pmc.log
  find gpurun_out/pmc -type f | head ;;
esac
done
