# GPU idle time inside the steps after the step-boundary fix: kernel trace of cfg2 / cfg3 under SEGMI_SGD_TABLE_UPLOAD=blocking and auto
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; : > gpurun_out/r06_gpu_gaps_after.txt
for c in cfg2 cfg3; do for up in blocking auto; do
  rm -rf /tmp/tr; SEGMI_SGD_TABLE_UPLOAD=$up timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o r -- python bench.py --config $c --steps 9 --warmup 5 --no-cpu --no-roofline --no-alt > /tmp/tr.full 2>&1
  f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
  { echo "=== $c, SEGMI_SGD_TABLE_UPLOAD=$up (under rocprofv3 --kernel-trace): $(grep '^{' /tmp/tr.full | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms')")"; python tools/gpu_gaps.py $f | sed -n '1,12p;/idle time by/,+5p'; } >> gpurun_out/r06_gpu_gaps_after.txt 2>&1
done; done
grep "===\|steps," gpurun_out/r06_gpu_gaps_after.txt | cut -c1-200
