# GPU idle time inside the steps: rocprofv3 kernel trace of the bench command, side stream on and off; cfg2 and cfg5
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for c in cfg2 cfg5; do for ws in 1 0; do
  rm -rf /tmp/tr; ( SEGMI_WGRAD_STREAM=$ws timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o r -- python bench.py --config $c --steps 7 --warmup 2 --no-cpu --no-roofline --no-alt > /tmp/tr.full 2>&1; grep '^{' /tmp/tr.full | cut -c1-160 > /tmp/tr.log; grep -v amdgpu.ids /tmp/tr.full | grep -v '^{' | tail -15 > gpurun_out/r06ar_${c}_$ws.log )
  f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
  { echo "# $c, SEGMI_WGRAD_STREAM=$ws: $(cat /tmp/tr.log)"; python tools/gpu_gaps.py $f; } > gpurun_out/r06_${c}_gpu_gaps_ws$ws.txt 2>&1
  head -8 gpurun_out/r06_${c}_gpu_gaps_ws$ws.txt | cut -c1-200
done; done
