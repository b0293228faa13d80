#!/bin/bash
# One GPU-box session (gpurun -- 'bash tools/gpu_round.sh <steps...>'); outputs under gpurun_out/, prefixed with $TAG (default r06).
#   suite     whole `-m gpu` suite (+ audit.json of the full-size audits)          smoke     __graft_entry__.smoke()
#   bench     the driver's default bench line (cfg2, every leg)                    benchall  one line per BASELINE config (with cpu_baseline)
#   prof      rocprofv3 --kernel-trace --stats of cfg2 (side stream on / in order) pmc       FETCH_SIZE / WRITE_SIZE passes -> conv traffic json
#   layers    per-layer conv table (tools/conv_layers.py)                          membound  HBM-bound call table per config
#   profcfg   rocprofv3 stats, in order, for cfg1 cfg3 cfg5                           sqpmc     SQ counters (MfmaUtil ...) of 1x1 layers, base vs this tree
# (the per-round recipes of rounds 1-5, incl. the retired bf16x3 arithmetic's, are in this file's git history)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
T="${TAG:-r06}"
what="${*:-suite smoke bench}"
stats() {   # stats <out.csv> <env...> -- <bench args...>
  local out=$1; shift
  rm -rf gpurun_out/prof
  ( env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python bench.py --steps 7 --warmup 2 --no-cpu --no-roofline --no-alt $BARGS 2>&1 | tail -2 ) > gpurun_out/prof.log
  find gpurun_out/prof -name "*kernel_trace*" -delete
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out"
  rm -rf gpurun_out/prof
}
for w in $what; do
case $w in
suite)
  rm -f gpurun_out/audit.json
  ( timeout 2400 python -m pytest tests -m gpu -q -rf -s -p no:cacheprovider --durations=15 2>&1 | grep -E "fullsize|2-rank|passed|failed|FAILED|ERROR|rel-L2|L2 error|Error|assert|UNet grad|distance from the fp64|s call|s setup" | tail -220 ) > gpurun_out/${T}_gpu_suite.txt
  tail -4 gpurun_out/${T}_gpu_suite.txt | cut -c1-300
  cp gpurun_out/audit.json gpurun_out/${T}_fullsize_audit.json 2>/dev/null ;;
smoke)
  ( timeout 900 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -8 ) > gpurun_out/${T}_smoke.txt; tail -2 gpurun_out/${T}_smoke.txt | cut -c1-300 ;;
bench)
  ( timeout 900 python bench.py 2>&1 | tail -1 ) > gpurun_out/${T}_cfg2_bench.json
  python -c "import json; d=json.loads(open('gpurun_out/${T}_cfg2_bench.json').read()); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['executed_step_frac'], d['cpu_baseline']['value'], [d[k]['value'] for k in ('alt_direct',) if d.get(k)])" 2>&1 | tail -1 ;;
benchall)
  for c in cfg1 cfg3 cfg4 cfg5; do
    ( timeout 700 python bench.py --config $c --no-alt --cpu-cap 120 2>&1 | tail -1 ) > gpurun_out/${T}_${c}_bench.json
    python -c "import json; d=json.loads(open('gpurun_out/${T}_${c}_bench.json').read()); r=d['roofline']; print('$c', d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['all_conv']['ms_per_step'], r['hbm_bound_calls']['ms_per_step'], (d.get('cpu_baseline') or {}).get('value'))" 2>&1 | tail -1
  done ;;
prof)
  BARGS="" stats gpurun_out/${T}_cfg2_kernel_stats_f32.csv SEGMI_WGRAD_STREAM=1
  BARGS="" stats gpurun_out/${T}_cfg2_kernel_stats_f32_inorder.csv SEGMI_WGRAD_STREAM=0
  head -4 gpurun_out/${T}_cfg2_kernel_stats_f32_inorder.csv | cut -c1-200 ;;
profcfg)
  for c in cfg1 cfg3 cfg5; do BARGS="--config $c" stats gpurun_out/${T}_${c}_kernel_stats.csv SEGMI_WGRAD_STREAM=0; done ;;
pmc)
  # HBM traffic of the conv launches of one cfg2 step: FETCH_SIZE and WRITE_SIZE in SEPARATE passes, kernel-trace only
  rm -rf gpurun_out/pmc_f32
  ( timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f32/fetch -o r -- python bench.py --steps 1 --warmup 1 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) > gpurun_out/pmc_f32.log
  ( timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_f32/write -o r -- python bench.py --steps 1 --warmup 1 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) >> gpurun_out/pmc_f32.log
  for d in fetch write; do f=$(find gpurun_out/pmc_f32/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/pmc_f32/$d/r_counter_collection.csv 2>/dev/null; done
  python tools/traffic_json.py gpurun_out/pmc_f32 gpurun_out/${T}_cfg2_conv_traffic_f32.json
  find gpurun_out/pmc_f32 -name "*kernel_trace*" -delete; find gpurun_out/pmc_f32 -name "*.csv" -size +8M -delete ;;
layers)
  ( timeout 300 python tools/conv_layers.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${T}_cfg2_conv_layers_f32.txt; tail -1 gpurun_out/${T}_cfg2_conv_layers_f32.txt ;;
sqpmc)
  # matrix-pipe counters (MfmaUtil, wait shares, effective clock) of one 1x1 layer per reduction length, all three passes, under the
  # library of 442b3b4 (tools/experiments/libsegmi_base.so, if present) and under this tree's: --pmc with --kernel-trace only
  L=pytorch-segmentation_amd/segmi/libsegmi.so; cp $L /tmp/cur.so
  for v in base cur; do
    [ $v = base ] && { [ -f tools/experiments/libsegmi_base.so ] || continue; cp tools/experiments/libsegmi_base.so $L; }
    [ $v = cur ] && cp /tmp/cur.so $L
    rm -rf gpurun_out/sqpmc_$v
    ( timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
        --output-format csv -d gpurun_out/sqpmc_$v/sq -o r -- python tools/conv_bench.py l4_1x1_up l4_1x1_down l3_1x1_up --iters 5 2>&1 | grep -v amdgpu.ids | tail -12 ) > gpurun_out/sqpmc_$v.log
    for d in gpurun_out/sqpmc_$v/*/; do f=$(find $d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $d/r_counter_collection.csv 2>/dev/null; f=$(find $d -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" $d/r_kernel_trace.csv 2>/dev/null; done
    ( echo "=== library: $v"; cat gpurun_out/sqpmc_$v.log; python tools/pmc_summary.py gpurun_out/sqpmc_$v/sq r conv_dma conv_wgrad 2>&1 | head -120 ) > gpurun_out/${T}_conv_pmc_$v.txt
    rm -rf gpurun_out/sqpmc_$v
  done
  cp /tmp/cur.so $L; grep -h "MfmaUtil\|launches=" gpurun_out/${T}_conv_pmc_*.txt | head -40 ;;
membound)
  for c in ${CFGS:-cfg2 cfg5}; do ( timeout 300 python tools/membound_ops.py $c 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${T}_${c}_membound_ops.txt; tail -2 gpurun_out/${T}_${c}_membound_ops.txt; done ;;
esac
done
