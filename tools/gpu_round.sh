#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, per-layer conv table, rocprof kernel stats.
# Outputs under gpurun_out/.   usage: tools/gpu_round.sh [tests] [testswino] [bench] [layers] [prof] [pmc] [x3] [x3prof] [x3pmc] [graph]
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
  ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python bench.py --steps 5 --warmup 2 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) > gpurun_out/prof.log
  find gpurun_out/prof -name "*kernel_trace*" -size +30M -delete
  find gpurun_out/prof -type f | head ;;
pmc)
  # HBM traffic of the conv launches of one cfg2 step, per arithmetic: FETCH_SIZE and WRITE_SIZE in SEPARATE passes, kernel-trace only
  for m in f32 bf16x3; do
    rm -rf gpurun_out/pmc_$m
    ( timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_$m/fetch -o r -- python bench.py --steps 1 --warmup 1 --no-cpu --no-roofline --no-alt --conv-math $m 2>&1 | tail -2 ) > gpurun_out/pmc_$m.log
    ( timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_$m/write -o r -- python bench.py --steps 1 --warmup 1 --no-cpu --no-roofline --no-alt --conv-math $m 2>&1 | tail -2 ) >> gpurun_out/pmc_$m.log
    for d in fetch write; do f=$(find gpurun_out/pmc_$m/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/pmc_$m/$d/r_counter_collection.csv 2>/dev/null; done
    python tools/traffic_json.py gpurun_out/pmc_$m gpurun_out/cfg2_conv_traffic_$m.json
    find gpurun_out/pmc_$m -name "*kernel_trace*" -delete
  done ;;
x3)
  # first hardware run of the bf16x3 convolution arithmetic: opt-in accuracy tests, then A/B of the isolated layers and
  # of the whole training step (SEGMI_CONV_MATH is read at the first convolution of the process)
  ( SEGMI_TEST_BF16X3=1 timeout 900 python -m pytest tests/test_conv_bf16x3_gpu.py -m gpu -q -s 2>&1 | tail -60 ) > gpurun_out/x3_tests.log
  cat gpurun_out/x3_tests.log
  for m in f32 bf16x3; do
    ( SEGMI_CONV_MATH=$m timeout 300 python tools/conv_bench.py psp_bottleneck l4_3x3_d4 l4_1x1_up l3_1x1_down l1_1x1 stem3 2>&1 | grep -v amdgpu.ids ) > gpurun_out/x3_convbench_$m.txt
    ( timeout 600 python bench.py --no-cpu --conv-math $m 2>&1 | tail -1 ) > gpurun_out/x3_bench_$m.log
  done
  paste gpurun_out/x3_convbench_f32.txt gpurun_out/x3_convbench_bf16x3.txt
  cat gpurun_out/x3_bench_*.log ;;
x3prof)
  rm -rf gpurun_out/x3prof
  ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x3prof -o r -- python bench.py --steps 5 --warmup 2 --no-cpu --no-roofline --conv-math bf16x3 2>&1 | tail -2 ) > gpurun_out/x3prof.log
  find gpurun_out/x3prof -name "*kernel_trace*" -size +30M -delete
  find gpurun_out/x3prof -type f | head ;;
x3pmc)
  # matrix-pipe / issue counters of ONE layer in isolation under both arithmetics (same counter set as profiles/r01_psp_bottleneck_pmc.txt;
  # --pmc with --kernel-trace only), then an LDS/L2 pass: is the bf16x3 loop VALU-issue bound as modelled, or waiting on operand DMA?
  rm -rf gpurun_out/x3pmc
  for m in f32 bf16x3; do
    ( SEGMI_CONV_MATH=$m timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
        --output-format csv -d gpurun_out/x3pmc/sq_$m -o r -- python tools/conv_bench.py psp_bottleneck l4_1x1_up --iters 5 2>&1 | tail -8 ) > gpurun_out/x3pmc_sq_$m.log
    ( SEGMI_CONV_MATH=$m timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum \
        --output-format csv -d gpurun_out/x3pmc/mem_$m -o r -- python tools/conv_bench.py psp_bottleneck l4_1x1_up --iters 5 2>&1 | tail -8 ) > gpurun_out/x3pmc_mem_$m.log
  done
  find gpurun_out/x3pmc -name "*kernel_trace*" -size +30M -delete
  for d in gpurun_out/x3pmc/*/; do echo "== $d"; python tools/pmc_summary.py $d r 2>&1 | head -40; done ;;
graph)
  # first hardware run of hipGraph capture (segmi/graph.py): opt-in tests, then eager vs replayed step on the launch-paced
  # UNet config and on the bench line
  ( SEGMI_TEST_GRAPH=1 timeout 900 python -m pytest tests/test_graph_gpu.py -m gpu -q 2>&1 | tail -25 ) > gpurun_out/graph_tests.log
  cat gpurun_out/graph_tests.log
  for c in cfg1 cfg2; do
    ( timeout 600 python bench.py --config $c --no-cpu --no-roofline 2>&1 | tail -1 ) > gpurun_out/graph_${c}_eager.log
    ( timeout 600 python bench.py --config $c --no-cpu --no-roofline --graph 2>&1 | tail -3 ) > gpurun_out/graph_${c}_graph.log
    cat gpurun_out/graph_${c}_eager.log gpurun_out/graph_${c}_graph.log
  done ;;
testsx3)
  # acceptance run of the bf16x3 conv arithmetic: the ENTIRE gpu suite (incl. the BASELINE-shape audits) at unchanged tolerances
  ( SEGMI_CONV_MATH=bf16x3 timeout 1500 python -m pytest tests -m gpu -q -rf -s 2>&1 | grep -E "fullsize|passed|failed|FAILED|ERROR|rel-L2|L2 error|Error|assert" | tail -70 ) > gpurun_out/pytest_gpu_bf16x3.log
  ( SEGMI_CONV_MATH=bf16x3 timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) >> gpurun_out/pytest_gpu_bf16x3.log
  cat gpurun_out/pytest_gpu_bf16x3.log ;;
wino)
  # first hardware run of the branch: Winograd tests (fwd / dgrad / wgrad), then A/B of the per-layer table and the bench line
  ( timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "winograd or conv2d_fwd_dgrad_wgrad or filter_transposes" 2>&1 | tail -15 ) > gpurun_out/wino_tests.log
  cat gpurun_out/wino_tests.log
  for wg in 0 1; do
    ( SEGMI_CONV_WINOGRAD=1 SEGMI_CONV_WINOGRAD_WGRAD=$wg timeout 200 python tools/conv_layers.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/conv_layers_winograd_wgrad$wg.txt
    tail -1 gpurun_out/conv_layers_winograd_wgrad$wg.txt
    ( SEGMI_CONV_WINOGRAD=1 SEGMI_CONV_WINOGRAD_WGRAD=$wg timeout 300 python bench.py --no-cpu --no-alt 2>&1 | tail -1 | cut -c1-220 ) > gpurun_out/bench_winograd_wgrad$wg.log
    cat gpurun_out/bench_winograd_wgrad$wg.log
  done ;;
testswino)
  # acceptance run of Winograd F(2x2,3x3) as the algorithm of the eligible 3x3 layers: the ENTIRE gpu suite at unchanged tolerances
  ( SEGMI_CONV_WINOGRAD=1 SEGMI_CONV_WINOGRAD_WGRAD=1 timeout 1500 python -m pytest tests -m gpu -q -rf -s 2>&1 | grep -E "fullsize|passed|failed|FAILED|ERROR|rel-L2|L2 error|Error|assert" | tail -70 ) > gpurun_out/pytest_gpu_winograd.log
  ( SEGMI_CONV_WINOGRAD=1 timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) >> gpurun_out/pytest_gpu_winograd.log
  cat gpurun_out/pytest_gpu_winograd.log ;;
testsf32)
  ( SEGMI_CONV_MATH=f32 timeout 1500 python -m pytest tests -m gpu -q -rf -s 2>&1 | grep -E "fullsize|passed|failed|FAILED|ERROR|rel-L2|L2 error|Error|assert" | tail -70 ) > gpurun_out/pytest_gpu_f32.log
  ( SEGMI_CONV_MATH=f32 timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) >> gpurun_out/pytest_gpu_f32.log
  cat gpurun_out/pytest_gpu_f32.log ;;
graphdbg)
  ( SEGMI_TEST_GRAPH=1 timeout 600 python -X faulthandler -m pytest tests/test_graph_gpu.py -m gpu -x -q -s 2>&1 | grep -v "^  File" | head -80 ) > gpurun_out/graph_tests_dbg.log
  cat gpurun_out/graph_tests_dbg.log ;;
x3exp)
  # where does the bf16x3 loop's time go?  fprop only: full arithmetic vs no split (planes = raw bits) vs no matrix instructions
  for m in f32 bf16x3; do
    ( SEGMI_CONV_MATH=$m timeout 300 python tools/conv_bench.py psp_bottleneck l4_3x3_d4 l4_1x1_up --op fwd 2>&1 | grep -v amdgpu.ids ) > gpurun_out/x3exp_$m.txt
    echo "== $m"; cat gpurun_out/x3exp_$m.txt
  done ;;
quick)
  # the tests touched last + bench with the alt leg
  ( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py tests/test_distributed_gpu.py -m gpu -q -rf -s 2>&1 | grep -E "passed|failed|FAILED|ERROR|Error|assert|UNet grad" | tail -30 ) > gpurun_out/quick_f32.log
  ( SEGMI_CONV_MATH=bf16x3 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py -m gpu -q -rf -s 2>&1 | grep -E "passed|failed|FAILED|ERROR|Error|assert|UNet grad" | tail -30 ) > gpurun_out/quick_bf16x3.log
  cat gpurun_out/quick_f32.log gpurun_out/quick_bf16x3.log ;;
spawn)
  # the self-launching multi-GPU bench on a 1-GPU box: two ranks share cuda:0 (gloo on device tensors; RCCL needs one GPU per rank)
  ( timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu --no-roofline 2>&1 | tail -4 ) > gpurun_out/spawn.log; cat gpurun_out/spawn.log ;;
r3a)
  # round 3, call 1: first hardware run of the merged Winograd branch (filter gradient, batch-aware tiles, wgrad epilogue pinning)
  ( SEGMI_CONV_WINOGRAD=1 SEGMI_CONV_WINOGRAD_WGRAD=1 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_pspnet_gpu.py tests/test_unet_gpu.py -m gpu -q -rf 2>&1 | tail -40 ) > gpurun_out/r3a_wino_tests.log
  cat gpurun_out/r3a_wino_tests.log
  ( timeout 400 python bench.py --no-cpu --no-alt 2>&1 | tail -1 ) > gpurun_out/r3a_bench_direct.log
  ( SEGMI_CONV_WINOGRAD=1 timeout 400 python bench.py --no-cpu --no-alt 2>&1 | tail -1 ) > gpurun_out/r3a_bench_wino.log
  ( SEGMI_CONV_WINOGRAD=1 SEGMI_CONV_WINOGRAD_WGRAD=1 timeout 400 python bench.py --no-cpu --no-alt 2>&1 | tail -1 ) > gpurun_out/r3a_bench_winowg.log
  for f in direct wino winowg; do python -c "import json,sys; d=json.loads(open('gpurun_out/r3a_bench_$f.log').read()); print('$f', d['value'], d['ms_per_step'], d['roofline']['all_conv'])" 2>&1 | tail -1; done
  ( SEGMI_CONV_WINOGRAD=1 SEGMI_CONV_WINOGRAD_WGRAD=1 timeout 300 python tools/conv_layers.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r3a_conv_layers_winowg.txt; tail -2 gpurun_out/r3a_conv_layers_winowg.txt
  ( SEGMI_CONV_WINOGRAD=1 SEGMI_CONV_WINOGRAD_WGRAD=1 timeout 1200 python -m pytest tests -m gpu -q -rf -s 2>&1 | grep -E "fullsize|passed|failed|FAILED|ERROR|rel-L2|L2 error|Error|assert" | tail -70 ) > gpurun_out/r3a_pytest_gpu_winowg.log
  cat gpurun_out/r3a_pytest_gpu_winowg.log ;;
r3b)
  # round 3, call 2: Winograd default (batched filter gradient), both-algorithm suite, bench + experiments, profiles
  ( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_pspnet_gpu.py tests/test_unet_gpu.py -m gpu -q -rf -x 2>&1 | tail -30 ) > gpurun_out/r3b_quick_tests.log
  tail -5 gpurun_out/r3b_quick_tests.log
  ( timeout 400 python bench.py --no-cpu 2>&1 | tail -1 ) > gpurun_out/r3b_bench.log
  ( SEGMI_WGRAD_STREAM=1 timeout 400 python bench.py --no-cpu --no-alt 2>&1 | tail -1 ) > gpurun_out/r3b_bench_side.log
  ( SEGMI_CONV_WINOGRAD_MIN_CHANNELS=128 timeout 400 python bench.py --no-cpu --no-alt 2>&1 | tail -1 ) > gpurun_out/r3b_bench_min128.log
  ( SEGMI_CONV_WINOGRAD_MIN_CHANNELS=64 timeout 400 python bench.py --no-cpu --no-alt 2>&1 | tail -1 ) > gpurun_out/r3b_bench_min64.log
  for f in bench bench_side bench_min128 bench_min64; do python -c "import json,sys; d=json.loads(open('gpurun_out/r3b_$f.log').read()); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['executed_step_frac'], r['all_conv']['ms_per_step'], [d[k]['value'] for k in ('alt','alt_direct') if d.get(k)])" 2>&1 | tail -1; done
  ( timeout 300 python tools/conv_layers.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r3b_conv_layers.txt; tail -1 gpurun_out/r3b_conv_layers.txt
  ( timeout 1500 python -m pytest tests -m gpu -q -rf -s 2>&1 | grep -E "fullsize|passed|failed|FAILED|ERROR|rel-L2|L2 error|Error|assert" | tail -70 ) > gpurun_out/r3b_pytest_gpu.log
  cat gpurun_out/r3b_pytest_gpu.log
  ( timeout 600 python __graft_entry__.py smoke 2>&1 | tail -6 ) > gpurun_out/r3b_smoke.log; cat gpurun_out/r3b_smoke.log ;;
pkrepro)
  # two-kernel reproducer of the round-2 packed-fp32 observation: victim (packed / asm / scalar) alone, next to an fp32-MFMA process,
  # next to a bf16-MFMA process (separate PROCESSES sharing the GPU), then the product-level stress at HEAD under bf16x3
  hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_mfma_repro tools/probes/pk_mfma_repro.hip 2>/dev/null
  ( for v in pk asm scalar; do
      echo "== victim $v alone"; /tmp/pk_mfma_repro victim 6 $v
      echo "== victim $v next to an fp32-MFMA process"; /tmp/pk_mfma_repro aggressor 9 f32 & sleep 1; /tmp/pk_mfma_repro victim 6 $v; wait
      echo "== victim $v next to a bf16-MFMA process"; /tmp/pk_mfma_repro aggressor 9 bf16 & sleep 1; /tmp/pk_mfma_repro victim 6 $v; wait
    done ) > gpurun_out/pkrepro.txt 2>&1
  cat gpurun_out/pkrepro.txt
  ( SEGMI_CONV_MATH=bf16x3 timeout 300 python tools/stress_determinism.py --procs 2 --iters 150 2>&1 | grep -v amdgpu.ids | tail -12 ) > gpurun_out/stress_bf16x3.txt
  cat gpurun_out/stress_bf16x3.txt ;;
pk2)
  ( timeout 400 python tools/probes/pk_two_process.py --seconds 10 2>&1 | grep -v amdgpu.ids ) > gpurun_out/pk_two_process.txt; cat gpurun_out/pk_two_process.txt ;;
r3c)
  # round 3, call 3: kept Winograd V + side-stream filter gradients (default), grouped SyncBN, DDP bucket slots; bf16x3 status
  ( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_pspnet_gpu.py tests/test_distributed_gpu.py tests/test_graph_gpu.py -m gpu -q -rf -x 2>&1 | tail -30 ) > gpurun_out/r3c_quick_tests.log
  tail -6 gpurun_out/r3c_quick_tests.log
  ( timeout 400 python bench.py --no-cpu 2>&1 | tail -1 ) > gpurun_out/r3c_bench.log
  ( SEGMI_CONV_WINOGRAD_KEEP_V=0 timeout 400 python bench.py --no-cpu --no-alt 2>&1 | tail -1 ) > gpurun_out/r3c_bench_nokeep.log
  ( SEGMI_WGRAD_STREAM=0 timeout 400 python bench.py --no-cpu --no-alt 2>&1 | tail -1 ) > gpurun_out/r3c_bench_inorder.log
  for f in bench bench_nokeep bench_inorder; do python -c "import json,sys; d=json.loads(open('gpurun_out/r3c_$f.log').read()); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['executed_step_frac'], r['all_conv']['ms_per_step'], [d[k]['value'] for k in ('alt','alt_direct') if d.get(k)])" 2>&1 | tail -1; done
  ( timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu --no-roofline 2>&1 | tail -2 ) > gpurun_out/r3c_spawn2.log; cat gpurun_out/r3c_spawn2.log
  bash tools/gpu_round.sh pkrepro
  ( SEGMI_CONV_MATH=bf16x3 timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_conv_bf16x3_gpu.py tests/test_distributed_gpu.py -m gpu -q -rf -s -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|ERROR|rel-L2|L2 error|Error|assert|UNet grad" | tail -40 ) > gpurun_out/r3c_pytest_gpu_bf16x3_subset.log
  cat gpurun_out/r3c_pytest_gpu_bf16x3_subset.log ;;
r3final)
  # round 3, evidence run of the final tree: whole suite, smoke, full bench line, rocprof stats, PMC traffic, tables, other configs
  ( timeout 1500 python -m pytest tests -m gpu -q -rf -s -p no:cacheprovider 2>&1 | grep -E "fullsize|passed|failed|FAILED|ERROR|rel-L2|L2 error|Error|assert|UNet grad" | tail -70 ) > gpurun_out/r3f_pytest_gpu.log
  tail -4 gpurun_out/r3f_pytest_gpu.log
  ( timeout 600 python __graft_entry__.py smoke 2>&1 | tail -6 ) > gpurun_out/r3f_smoke.log; tail -2 gpurun_out/r3f_smoke.log | cut -c1-200
  ( timeout 900 python bench.py 2>&1 | tail -1 ) > gpurun_out/r3f_bench.log
  python -c "import json; d=json.loads(open('gpurun_out/r3f_bench.log').read()); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['executed_step_frac'], d['cpu_baseline'], [d[k]['value'] for k in ('alt','alt_direct') if d.get(k)])" 2>&1 | tail -1
  rm -rf gpurun_out/prof
  ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python bench.py --steps 5 --warmup 2 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) > gpurun_out/prof.log
  find gpurun_out/prof -name "*kernel_trace*" -delete
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r3f_kernel_stats_f32.csv
  rm -rf gpurun_out/pmc_f32
  ( timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f32/fetch -o r -- python bench.py --steps 1 --warmup 1 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) > gpurun_out/pmc_f32.log
  ( timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_f32/write -o r -- python bench.py --steps 1 --warmup 1 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) >> gpurun_out/pmc_f32.log
  for d in fetch write; do f=$(find gpurun_out/pmc_f32/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/pmc_f32/$d/r_counter_collection.csv 2>/dev/null; done
  python tools/traffic_json.py gpurun_out/pmc_f32 gpurun_out/r3f_cfg2_conv_traffic_f32.json
  find gpurun_out/pmc_f32 -name "*kernel_trace*" -delete; find gpurun_out/pmc_f32 -name "*.csv" -size +8M -delete
  ( timeout 300 python tools/conv_layers.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r3f_conv_layers.txt; tail -1 gpurun_out/r3f_conv_layers.txt
  ( timeout 300 python tools/membound_ops.py cfg2 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r3f_membound_ops.txt; tail -3 gpurun_out/r3f_membound_ops.txt
  ( timeout 300 python tools/stray_aten.py 2>&1 | grep -v amdgpu.ids | tail -20 ) > gpurun_out/r3f_stray_aten.txt
  for c in cfg1 cfg3 cfg4 cfg5; do ( timeout 400 python bench.py --config $c --no-cpu --no-roofline --no-alt 2>&1 | tail -1 ) > gpurun_out/r3f_bench_$c.log; python -c "import json; d=json.loads(open('gpurun_out/r3f_bench_$c.log').read()); print('$c', d['value'], d['ms_per_step'])" 2>&1 | tail -1; done
  bash tools/gpu_round.sh pk2 ;;
r5final)
  # round 5, evidence run of the final tree (the whole suite runs in its own call): full bench line, rocprof stats (default and
  # in order), PMC traffic, per-layer / per-call tables, every other BASELINE config WITH its cpu_baseline and roofline
  ( timeout 900 python bench.py 2>&1 | tail -1 ) > gpurun_out/r5f_bench_cfg2.json
  python -c "import json; d=json.loads(open('gpurun_out/r5f_bench_cfg2.json').read()); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['executed_step_frac'], d['cpu_baseline']['value'], [d[k]['value'] for k in ('alt','alt_direct') if d.get(k)])" 2>&1 | tail -1
  for mode in default inorder; do
    rm -rf gpurun_out/prof
    ( cd /tmp; SEGMI_WGRAD_STREAM=$([ $mode = inorder ] && echo 0 || echo 1) timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) > gpurun_out/prof.log
    find gpurun_out/prof -name "*kernel_trace*" -delete
    f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r5f_cfg2_kernel_stats_f32_$mode.csv
  done
  head -4 gpurun_out/r5f_cfg2_kernel_stats_f32_inorder.csv | cut -c1-200
  rm -rf gpurun_out/pmc_f32
  ( cd /tmp; timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_f32/fetch -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) > gpurun_out/pmc_f32.log
  ( cd /tmp; timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_f32/write -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) >> gpurun_out/pmc_f32.log
  for d in fetch write; do f=$(find gpurun_out/pmc_f32/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/pmc_f32/$d/r_counter_collection.csv 2>/dev/null; done
  python tools/traffic_json.py gpurun_out/pmc_f32 gpurun_out/r5f_cfg2_conv_traffic_f32.json
  find gpurun_out/pmc_f32 -name "*kernel_trace*" -delete; find gpurun_out/pmc_f32 -name "*.csv" -size +8M -delete
  ( timeout 300 python tools/conv_layers.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5f_cfg2_conv_layers.txt; tail -1 gpurun_out/r5f_cfg2_conv_layers.txt
  for c in cfg2 cfg5 cfg3 cfg1; do ( timeout 300 python tools/membound_ops.py $c 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5f_${c}_membound_ops.txt; tail -2 gpurun_out/r5f_${c}_membound_ops.txt; done
  ( timeout 300 python tools/stray_aten.py 2>&1 | grep -v amdgpu.ids | tail -20 ) > gpurun_out/r5f_cfg2_stray_aten.txt
  for c in cfg1 cfg3 cfg4 cfg5; do
    ( timeout 600 python bench.py --config $c --no-alt --cpu-cap 110 2>&1 | tail -1 ) > gpurun_out/r5f_bench_$c.json
    python -c "import json; d=json.loads(open('gpurun_out/r5f_bench_$c.json').read()); print('$c', d['value'], d['ms_per_step'], d['roofline']['executed_step_frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])" 2>&1 | tail -1
  done
  python tools/top_kernels.py gpurun_out/r5f_bench_cfg1.json gpurun_out/r5f_bench_cfg3.json gpurun_out/r5f_bench_cfg4.json gpurun_out/r5f_bench_cfg5.json gpurun_out/r5f_bench_cfg2.json > gpurun_out/r5f_top_kernels.txt
  for c in cfg5 cfg3; do
    rm -rf gpurun_out/prof
    ( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 5 --warmup 2 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) > gpurun_out/prof.log
    find gpurun_out/prof -name "*kernel_trace*" -delete
    f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r5f_${c}_kernel_stats.csv
  done
  ( timeout 300 python tools/lovasz_bench.py --iters 5 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5f_lovasz_alone_bench.txt; cat gpurun_out/r5f_lovasz_alone_bench.txt ;;
r5last)
  # round 5, last call on the final tree: whole suite + smoke, cfg5 evidence refreshed (depthwise filter gradients on the side stream,
  # Xception block tails), HBM traffic of the Lovasz kernels (PMC, separate passes)
  ( timeout 1150 python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider 2>&1 | tail -28 ) > gpurun_out/r5l_gpu_suite.txt; tail -4 gpurun_out/r5l_gpu_suite.txt
  cp gpurun_out/audit.json gpurun_out/r5l_fullsize_audit.json 2>/dev/null
  ( timeout 300 python __graft_entry__.py smoke 2>&1 | grep graft | cut -c1-400 ) > gpurun_out/r5l_smoke.txt; cut -c1-160 gpurun_out/r5l_smoke.txt
  ( timeout 600 python bench.py --config cfg5 --no-alt --cpu-cap 150 2>&1 | tail -1 ) > gpurun_out/r5l_bench_cfg5.json
  python -c "import json; d=json.loads(open('gpurun_out/r5l_bench_cfg5.json').read()); print('cfg5', d['value'], d['ms_per_step'], d['roofline']['executed_step_frac'], d['roofline']['hbm_bound_calls']['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])" 2>&1 | tail -1
  ( timeout 300 python tools/membound_ops.py cfg5 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5l_cfg5_membound_ops.txt; tail -2 gpurun_out/r5l_cfg5_membound_ops.txt
  rm -rf gpurun_out/prof
  ( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) > gpurun_out/prof.log
  find gpurun_out/prof -name "*kernel_trace*" -delete
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r5l_cfg5_kernel_stats.csv
  ( timeout 300 python bench.py --no-cpu --no-alt --no-roofline 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('cfg2', j['value'], j['ms_per_step'])" )
  rm -rf gpurun_out/lpmc
  for pc in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc $pc --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lpmc/$pc -o r -- python $GRAFT_REPO_ROOT/tools/lovasz_bench.py --iters 2 --modes random --prune 1 2>&1 | tail -1 ) > /dev/null
    f=$(find gpurun_out/lpmc/$pc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/lpmc/$pc/r_counter_collection.csv
    f=$(find gpurun_out/lpmc/$pc -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/lpmc/$pc/r_kernel_trace.csv
    echo "== $pc (KB per launch; FETCH_SIZE under-counts 16 B/lane streaming reads 2x on gfx950: MI355X_MICROARCH.md)" >> gpurun_out/r5l_lovasz_traffic.txt
    python tools/pmc_summary.py gpurun_out/lpmc/$pc r lovasz segsort >> gpurun_out/r5l_lovasz_traffic.txt 2>&1
  done
  find gpurun_out/lpmc -name "*.csv" -size +4M -delete
  tail -30 gpurun_out/r5l_lovasz_traffic.txt ;;
r3g)
  # round 3, call 5: in-order rocprof stats (agreement with bench.py's instrumented step), packed-fp32 cross experiment, new tests,
  # whole suite under bf16x3 at HEAD
  rm -rf gpurun_out/prof
  ( SEGMI_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python bench.py --steps 5 --warmup 2 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) > gpurun_out/prof.log
  find gpurun_out/prof -name "*kernel_trace*" -delete
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r3g_kernel_stats_f32_inorder.csv; head -4 gpurun_out/r3g_kernel_stats_f32_inorder.csv | cut -c1-200
  ( timeout 400 python tools/probes/pk_two_process.py --cross --seconds 8 2>&1 | grep -v amdgpu.ids ) > gpurun_out/pk_two_process_cross.txt; cat gpurun_out/pk_two_process_cross.txt
  ( timeout 900 python -m pytest tests/test_determinism_gpu.py tests/test_conv_bf16x3_gpu.py tests/test_ops_gpu.py tests/test_pspnet_gpu.py -m gpu -q -rf -s -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|ERROR|distance from the fp64|Error|assert" | tail -30 ) > gpurun_out/r3g_new_tests.log; cat gpurun_out/r3g_new_tests.log
  ( SEGMI_CONV_MATH=bf16x3 timeout 1500 python -m pytest tests -m gpu -q -rf -s -p no:cacheprovider 2>&1 | grep -E "fullsize|passed|failed|FAILED|ERROR|rel-L2|L2 error|Error|assert|UNet grad|distance from the fp64" | tail -70 ) > gpurun_out/r3g_pytest_gpu_bf16x3.log
  tail -12 gpurun_out/r3g_pytest_gpu_bf16x3.log | cut -c1-300 ;;
r3h)
  # round 3, last call: side-stream priority A/B, then the whole suite + smoke on the final tree
  python -c "import torch; print('stream priority range', torch.cuda.Stream.priority_range())" 2>&1 | tail -1
  for pr in 0 1 -1; do ( SEGMI_WGRAD_STREAM_PRIORITY=$pr timeout 400 python bench.py --no-cpu --no-alt --no-roofline 2>&1 | tail -1 ) > gpurun_out/r3h_bench_prio$pr.log; python -c "import json; d=json.loads(open('gpurun_out/r3h_bench_prio$pr.log').read()); print('priority $pr', d['value'], d['ms_per_step'])" 2>&1 | tail -1; done
  ( timeout 1500 python -m pytest tests -m gpu -q -rf -s -p no:cacheprovider 2>&1 | grep -E "fullsize|passed|failed|FAILED|ERROR|rel-L2|L2 error|Error|assert|UNet grad|distance from the fp64" | tail -70 ) > gpurun_out/r3h_pytest_gpu.log
  tail -4 gpurun_out/r3h_pytest_gpu.log | cut -c1-300
  ( timeout 600 python __graft_entry__.py smoke 2>&1 | tail -6 ) > gpurun_out/r3h_smoke.log; tail -1 gpurun_out/r3h_smoke.log | cut -c1-200 ;;
r3pmc)
  # matrix-pipe counters of the Winograd contraction (batched implicit-GEMM / filter-gradient launches) and of a direct 1x1 layer, one
  # layer each in isolation (--pmc with --kernel-trace only; same counter set as profiles/r01_conv_dma_pmc.txt)
  rm -rf gpurun_out/r3pmc
  ( timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
      --output-format csv -d gpurun_out/r3pmc/sq -o r -- python tools/conv_bench.py aux_3x3 l4_1x1_up --iters 5 2>&1 | tail -8 ) > gpurun_out/r3pmc_sq.log
  for d in gpurun_out/r3pmc/*/; do f=$(find $d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $d/r_counter_collection.csv 2>/dev/null; f=$(find $d -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" $d/r_kernel_trace.csv 2>/dev/null; done
  ( cat gpurun_out/r3pmc_sq.log; python tools/pmc_summary.py gpurun_out/r3pmc/sq r conv_ wino_ 2>&1 | head -80 ) > gpurun_out/r3pmc_summary.txt; cat gpurun_out/r3pmc_summary.txt
  find gpurun_out/r3pmc -name "*.csv" -size +4M -delete ;;
r3i)
  bash tools/gpu_round.sh r3pmc
  ( timeout 700 python -m pytest tests/test_conv_bf16x3_gpu.py tests/test_distributed_gpu.py -m gpu -q -rf --durations=12 -p no:cacheprovider 2>&1 | tail -30 ) > gpurun_out/r3i_tests.log; cat gpurun_out/r3i_tests.log
  ( timeout 300 python bench.py --no-cpu --no-alt --no-roofline 2>&1 | tail -1 ) > gpurun_out/r3i_bench.log; python -c "import json; d=json.loads(open('gpurun_out/r3i_bench.log').read()); print('bench', d['value'], d['ms_per_step'])" ;;
r3j)
  # hand-written segmented Lovasz sort: bit-identity vs the library sort, goldens, cfg5 audit, cfg5 A/B
  ( timeout 400 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_golden_gpu.py tests/test_fullsize_properties_gpu.py -k "lovasz or losses_match or cfg5" -m gpu -q -rf -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/r3j_tests.log; cat gpurun_out/r3j_tests.log
  ( timeout 200 python bench.py --config cfg5 --no-cpu --no-roofline --no-alt 2>&1 | tail -1 ) > gpurun_out/r3j_cfg5_seg.log
  ( SEGMI_LOVASZ_SORT=rocprim timeout 200 python bench.py --config cfg5 --no-cpu --no-roofline --no-alt 2>&1 | tail -1 ) > gpurun_out/r3j_cfg5_rocprim.log
  for f in seg rocprim; do python -c "import json; d=json.loads(open('gpurun_out/r3j_cfg5_$f.log').read()); print('cfg5 $f', d['value'], d['ms_per_step'])" 2>&1 | tail -1; done ;;
r3k)
  ( timeout 200 python -m pytest tests/test_ops_gpu.py -k "lovasz" -m gpu -q -p no:cacheprovider 2>&1 | tail -3 ) > gpurun_out/r3k_tests.log; cat gpurun_out/r3k_tests.log
  ( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -6 ) > gpurun_out/r3k_smoke.log; cut -c1-220 gpurun_out/r3k_smoke.log ;;
r4a)
  # round 4, call 1: kernel-level evidence for the four other BASELINE configs (rocprofv3 kernel stats, filter gradients in order so
  # durations are per-kernel), the HBM-bound call table of each, and the un-profiled bench line of each
  for c in cfg1 cfg3 cfg4 cfg5; do
    rm -rf gpurun_out/prof_$c
    ( SEGMI_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$c -o r -- python bench.py --config $c --steps 7 --warmup 2 --no-cpu --no-roofline --no-alt 2>&1 | tail -1 ) > gpurun_out/r4a_prof_$c.log
    find gpurun_out/prof_$c -name "*kernel_trace*" -delete
    f=$(find gpurun_out/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4a_${c}_kernel_stats.csv
    rm -rf gpurun_out/prof_$c
    ( timeout 300 python tools/membound_ops.py $c 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r4a_${c}_membound_ops.txt
    ( timeout 400 python bench.py --config $c --no-cpu --no-alt 2>&1 | tail -1 ) > gpurun_out/r4a_bench_$c.log
    python -c "import json; d=json.loads(open('gpurun_out/r4a_bench_$c.log').read()); r=d['roofline']; print('$c', d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['all_conv']['ms_per_step'])" 2>&1 | tail -1
    head -8 gpurun_out/r4a_${c}_kernel_stats.csv | cut -c1-160
  done ;;
r4b)
  # round 4, call 2: ADVICE fixes, BN statistics from the conv epilogue, split-major wgrad order: tests, then A/B bench lines
  ( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_distributed_gpu.py tests/test_pspnet_gpu.py tests/test_unet_gpu.py -m gpu -q -rf -x -p no:cacheprovider 2>&1 | tail -30 ) > gpurun_out/r4b_tests.log
  tail -8 gpurun_out/r4b_tests.log
  for v in default nostats noflat; do
    case $v in default) e="";; nostats) e="SEGMI_CONV_BN_STATS=0";; noflat) e="SEGMI_WGRAD_FLAT=0";; esac
    ( env $e timeout 400 python bench.py --no-cpu --no-alt 2>&1 | tail -1 ) > gpurun_out/r4b_bench_$v.log
    python -c "import json; d=json.loads(open('gpurun_out/r4b_bench_$v.log').read()); r=d['roofline']; print('cfg2 $v', d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['all_conv']['ms_per_step'])" 2>&1 | tail -1
  done
  for c in cfg1 cfg3 cfg5; do for v in default nostats; do
    case $v in default) e="";; nostats) e="SEGMI_CONV_BN_STATS=0";; esac
    ( env $e timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>&1 | tail -1 ) > gpurun_out/r4b_bench_${c}_$v.log
    python -c "import json; d=json.loads(open('gpurun_out/r4b_bench_${c}_$v.log').read()); print('$c $v', d['value'], d['ms_per_step'])" 2>&1 | tail -1
  done; done ;;
r4c)
  # round 4, call 3: single-read Lovasz scatter, strip-tiled depthwise kernels: tests, then cfg5 A/B
  ( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -rf -x -p no:cacheprovider -k "depthwise or lovasz or losses_match or winograd_keeps or colsum or conv_transpose" 2>&1 | tail -12 ) > gpurun_out/r4c_tests.log
  tail -5 gpurun_out/r4c_tests.log
  ( timeout 300 python -m pytest tests/test_fullsize_golden_gpu.py tests/test_deeplab_gpu.py -m gpu -q -rf -x -p no:cacheprovider -k "cfg5 or xception" 2>&1 | tail -12 ) > gpurun_out/r4c_tests_cfg5.log
  tail -4 gpurun_out/r4c_tests_cfg5.log
  for v in default nostrip rocprim; do
    case $v in default) e="";; nostrip) e="SEGMI_DW_STRIP=0";; rocprim) e="SEGMI_LOVASZ_SORT=rocprim";; esac
    ( env $e timeout 400 python bench.py --config cfg5 --no-cpu --no-alt --no-roofline 2>&1 | tail -1 ) > gpurun_out/r4c_bench_cfg5_$v.log
    python -c "import json; d=json.loads(open('gpurun_out/r4c_bench_cfg5_$v.log').read()); print('cfg5 $v', d['value'], d['ms_per_step'])" 2>&1 | tail -1
  done
  rm -rf gpurun_out/prof_cfg5
  ( SEGMI_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg5 -o r -- python bench.py --config cfg5 --steps 7 --warmup 2 --no-cpu --no-roofline --no-alt 2>&1 | tail -1 ) > gpurun_out/r4c_prof_cfg5.log
  find gpurun_out/prof_cfg5 -name "*kernel_trace*" -delete
  f=$(find gpurun_out/prof_cfg5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4c_cfg5_kernel_stats.csv
  rm -rf gpurun_out/prof_cfg5
  head -14 gpurun_out/r4c_cfg5_kernel_stats.csv | cut -c1-150
  ( timeout 300 python bench.py --config cfg1 --no-cpu --no-alt --no-roofline 2>&1 | tail -1 | cut -c1-200 ) ;;
r4d)
  # round 4, call 4: the input pipeline (f4) on hardware, then the whole suite on the current tree
  ( timeout 600 python -m pytest tests/test_augment.py -m gpu -q -rf -x -p no:cacheprovider 2>&1 | tail -25 ) > gpurun_out/r4d_augment.log
  tail -8 gpurun_out/r4d_augment.log
  ( timeout 1800 python -m pytest tests -m gpu -q -rf -s -p no:cacheprovider 2>&1 | grep -E "fullsize|passed|failed|FAILED|ERROR|rel-L2|L2 error|Error|assert|UNet grad|distance from the fp64" | tail -70 ) > gpurun_out/r4d_pytest_gpu.log
  tail -6 gpurun_out/r4d_pytest_gpu.log | cut -c1-400
  ( timeout 300 python bench.py --config cfg5 --no-cpu --no-alt --no-roofline 2>&1 | tail -1 | cut -c1-200 ) ;;
r4e)
  # round 4, call 5: re-run of the files that failed in r4d (WeakSet membership of tensors) + the new distributed-readiness tests
  ( timeout 1500 python -m pytest tests/test_augment.py tests/test_ops_gpu.py tests/test_trainer_gpu.py tests/test_distributed_gpu.py tests/test_determinism_gpu.py tests/test_graph_gpu.py -m gpu -q -rf -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|ERROR|Error|assert" | tail -40 ) > gpurun_out/r4e_tests.log
  tail -12 gpurun_out/r4e_tests.log | cut -c1-400 ;;
r4f)
  ( timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_distributed_gpu.py tests/test_determinism_gpu.py tests/test_fullsize_golden_gpu.py -m gpu -q -rf -s -p no:cacheprovider -k "pairing or two_rank or reducer_waits or determinism or repeatable or fullsize or rccl or bench_launcher or grad_slots or side_stream" 2>&1 | grep -E "fullsize|passed|failed|FAILED|ERROR|Error|assert" | tail -40 ) > gpurun_out/r4f_tests.log
  tail -14 gpurun_out/r4f_tests.log | cut -c1-700
  ( timeout 400 python bench.py --no-cpu --no-alt 2>&1 | tail -1 ) > gpurun_out/r4f_bench.log
  python -c "import json; d=json.loads(open('gpurun_out/r4f_bench.log').read()); r=d['roofline']; print('cfg2', d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['all_conv']['ms_per_step'])" 2>&1 | tail -1 ;;
r4g)
  # fp64 accumulation in the BN backward reduction: re-run of the two fixed tests, BN / model tests, the four audits (distance from the
  # fp64 gradient digests), cfg2 bench
  ( timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_distributed_gpu.py tests/test_fullsize_golden_gpu.py tests/test_pspnet_gpu.py tests/test_unet_gpu.py -m gpu -q -rf -s -p no:cacheprovider -k "pairing or two_rank_syncbn_step or fullsize or batch_norm or pspnet or unet" 2>&1 | grep -E "fullsize|passed|failed|FAILED|ERROR|Error|assert" | tail -40 ) > gpurun_out/r4g_tests.log
  tail -6 gpurun_out/r4g_tests.log | cut -c1-300
  python - <<'PY'
import json
d=json.load(open('gpurun_out/audit.json'))
for k,v in sorted(d.items()):
    if '/f32/' in k and 'grad_f64_rel_err_median' in v:
        print(k, "grad f64 HIP med %.3e max %.3e | ref med %.3e max %.3e | ratio %.2f %.2f"%(v['grad_f64_rel_err_median'], v['grad_f64_rel_err_max'], v['ref_grad_f64_rel_err_median'], v['ref_grad_f64_rel_err_max'], v['grad_f64_rel_err_median']/v['ref_grad_f64_rel_err_median'], v['grad_f64_rel_err_max']/v['ref_grad_f64_rel_err_max']))
PY
  ( timeout 400 python bench.py --no-cpu --no-alt 2>&1 | tail -1 ) > gpurun_out/r4g_bench.log
  python -c "import json; d=json.loads(open('gpurun_out/r4g_bench.log').read()); r=d['roofline']; print('cfg2', d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['all_conv']['ms_per_step'])" 2>&1 | tail -1
  ( timeout 300 python tools/membound_ops.py cfg2 2>&1 | grep -v amdgpu.ids | head -8 ) ;;
r4h)
  # two-level fp32 accumulation in the conv K loop: error probe, conv / model / audit tests, noise breakdown, bench
  ( timeout 300 python tools/probes/conv_error_vs_fp64.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -20 ) > gpurun_out/r4h_conv_error.txt; cat gpurun_out/r4h_conv_error.txt
  ( timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_golden_gpu.py tests/test_pspnet_gpu.py tests/test_unet_gpu.py tests/test_deeplab_gpu.py tests/test_fullsize_properties_gpu.py -m gpu -q -rf -s -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|ERROR|Error|assert" | tail -20 ) > gpurun_out/r4h_tests.log
  tail -6 gpurun_out/r4h_tests.log | cut -c1-300
  python - <<'PY'
import json
d=json.load(open('gpurun_out/audit.json'))
for k,v in sorted(d.items()):
    if '/f32/' in k and 'grad_f64_rel_err_median' in v:
        print(k, "logit dist fp64 HIP %.3e ref %.3e | grad f64 HIP med %.3e max %.3e | ratio %.2f %.2f"%(v['hip_err_f64'], v['ref_err_f64'], v['grad_f64_rel_err_median'], v['grad_f64_rel_err_max'], v['grad_f64_rel_err_median']/v['ref_grad_f64_rel_err_median'], v['grad_f64_rel_err_max']/v['ref_grad_f64_rel_err_max']))
PY
  ( timeout 400 python bench.py --no-cpu --no-alt 2>&1 | tail -1 ) > gpurun_out/r4h_bench.log
  python -c "import json; d=json.loads(open('gpurun_out/r4h_bench.log').read()); r=d['roofline']; print('cfg2', d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['all_conv']['ms_per_step'], r['hbm_bound_calls']['ms_per_step'])" 2>&1 | tail -1 ;;
r4j)
  ( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_pspnet_gpu.py tests/test_deeplab_gpu.py tests/test_distributed_gpu.py -m gpu -q -rf -p no:cacheprovider -k "fan or pspnet or deeplab or resnet or two_rank or filter_gradients or reducer" 2>&1 | grep -E "passed|failed|FAILED|ERROR|Error|assert" | tail -12 ) > gpurun_out/r4j_tests.log
  tail -5 gpurun_out/r4j_tests.log | cut -c1-300
  for v in default nofan; do
    case $v in default) e="";; nofan) e="SEGMI_CONV_FAN=0";; esac
    ( env $e timeout 400 python bench.py --no-cpu --no-alt 2>&1 | tail -1 ) > gpurun_out/r4j_bench_$v.log
    python -c "import json; d=json.loads(open('gpurun_out/r4j_bench_$v.log').read()); r=d['roofline']; print('cfg2 $v', d['value'], d['ms_per_step'], r['achieved'], r['all_conv']['ms_per_step'], r['hbm_bound_calls']['ms_per_step'])" 2>&1 | tail -1
  done
  ( timeout 300 python tools/stray_aten.py 2>&1 | grep -v amdgpu.ids | tail -14 ) ;;
r4final)
  # round 4, evidence run of the final tree
  ( timeout 1800 python -m pytest tests -m gpu -q -rf -s -p no:cacheprovider 2>&1 | grep -E "fullsize|passed|failed|FAILED|ERROR|rel-L2|L2 error|Error|assert|UNet grad|distance from the fp64" | tail -70 ) > gpurun_out/r4f_pytest_gpu.log
  tail -3 gpurun_out/r4f_pytest_gpu.log | cut -c1-300
  cp gpurun_out/audit.json gpurun_out/r4f_fullsize_audit.json 2>/dev/null
  ( timeout 600 python __graft_entry__.py smoke 2>&1 | tail -6 ) > gpurun_out/r4f_smoke.log; tail -1 gpurun_out/r4f_smoke.log | cut -c1-200
  ( timeout 900 python bench.py 2>&1 | tail -1 ) > gpurun_out/r4f_bench.log
  python -c "import json; d=json.loads(open('gpurun_out/r4f_bench.log').read()); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['executed_step_frac'], d['cpu_baseline']['value'], [d[k]['value'] for k in ('alt','alt_direct') if d.get(k)])" 2>&1 | tail -1
  rm -rf gpurun_out/prof
  ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python bench.py --steps 5 --warmup 2 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) > gpurun_out/prof.log
  find gpurun_out/prof -name "*kernel_trace*" -delete
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4f_kernel_stats_f32.csv
  rm -rf gpurun_out/prof
  ( SEGMI_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python bench.py --steps 5 --warmup 2 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) > gpurun_out/prof.log
  find gpurun_out/prof -name "*kernel_trace*" -delete
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4f_kernel_stats_f32_inorder.csv; head -3 gpurun_out/r4f_kernel_stats_f32_inorder.csv | cut -c1-200
  rm -rf gpurun_out/prof gpurun_out/pmc_f32
  ( timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f32/fetch -o r -- python bench.py --steps 1 --warmup 1 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) > gpurun_out/pmc_f32.log
  ( timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_f32/write -o r -- python bench.py --steps 1 --warmup 1 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) >> gpurun_out/pmc_f32.log
  for d in fetch write; do f=$(find gpurun_out/pmc_f32/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/pmc_f32/$d/r_counter_collection.csv 2>/dev/null; done
  python tools/traffic_json.py gpurun_out/pmc_f32 gpurun_out/r4f_cfg2_conv_traffic_f32.json
  find gpurun_out/pmc_f32 -name "*kernel_trace*" -delete; find gpurun_out/pmc_f32 -name "*.csv" -size +8M -delete
  ( timeout 300 python tools/conv_layers.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r4f_conv_layers.txt; tail -1 gpurun_out/r4f_conv_layers.txt
  ( timeout 300 python tools/membound_ops.py cfg2 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r4f_membound_ops.txt; tail -2 gpurun_out/r4f_membound_ops.txt
  ( timeout 300 python tools/stray_aten.py 2>&1 | grep -v amdgpu.ids | tail -20 ) > gpurun_out/r4f_stray_aten.txt
  ( timeout 300 python tools/grad_noise.py cfg2 cfg3 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r4f_grad_noise.txt
  for c in cfg1 cfg3 cfg4 cfg5; do
    ( timeout 400 python bench.py --config $c --no-cpu --no-alt 2>&1 | tail -1 ) > gpurun_out/r4f_bench_$c.log
    python -c "import json; d=json.loads(open('gpurun_out/r4f_bench_$c.log').read()); r=d['roofline']; print('$c', d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['all_conv']['ms_per_step'], r['hbm_bound_calls']['ms_per_step'])" 2>&1 | tail -1
    rm -rf gpurun_out/prof_$c
    ( SEGMI_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$c -o r -- python bench.py --config $c --steps 7 --warmup 2 --no-cpu --no-roofline --no-alt 2>&1 | tail -1 ) > gpurun_out/r4f_prof_$c.log
    find gpurun_out/prof_$c -name "*kernel_trace*" -delete
    f=$(find gpurun_out/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4f_${c}_kernel_stats.csv
    rm -rf gpurun_out/prof_$c
    ( timeout 300 python tools/membound_ops.py $c 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r4f_${c}_membound_ops.txt
  done
  ( SEGMI_LOVASZ_SORT=rocprim timeout 300 python bench.py --config cfg5 --no-cpu --no-alt --no-roofline 2>&1 | tail -1 ) > gpurun_out/r4f_bench_cfg5_rocprim.log
  python -c "import json; d=json.loads(open('gpurun_out/r4f_bench_cfg5_rocprim.log').read()); print('cfg5 rocprim', d['value'], d['ms_per_step'])" 2>&1 | tail -1
  ( SEGMI_CONV_MATH=bf16x3 timeout 900 python -m pytest tests/test_fullsize_golden_gpu.py tests/test_unet_gpu.py -m gpu -q -rf -s -p no:cacheprovider 2>&1 | grep -E "fullsize|passed|failed|FAILED|UNet grad|assert" | tail -30 ) > gpurun_out/r4f_bf16x3_audit.log
  tail -3 gpurun_out/r4f_bf16x3_audit.log | cut -c1-300 ;;
r4z)
  # round 4, last call: the whole suite + smoke + the bench lines on the final tree (Winograd from 128 channels, windowed Lovasz scatter)
  ( timeout 1800 python -m pytest tests -m gpu -q -rf -s -p no:cacheprovider 2>&1 | grep -E "fullsize|passed|failed|FAILED|ERROR|rel-L2|L2 error|Error|assert|UNet grad|distance from the fp64" | tail -70 ) > gpurun_out/r4z_pytest_gpu.log
  tail -3 gpurun_out/r4z_pytest_gpu.log | cut -c1-300
  cp gpurun_out/audit.json gpurun_out/r4z_fullsize_audit.json 2>/dev/null
  ( timeout 600 python __graft_entry__.py smoke 2>&1 | tail -6 ) > gpurun_out/r4z_smoke.log; tail -1 gpurun_out/r4z_smoke.log | cut -c1-200
  ( timeout 900 python bench.py 2>&1 | tail -1 ) > gpurun_out/r4z_bench.log
  python -c "import json; d=json.loads(open('gpurun_out/r4z_bench.log').read()); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['executed_step_frac'], d['cpu_baseline']['value'], [d[k]['value'] for k in ('alt','alt_direct') if d.get(k)])" 2>&1 | tail -1
  for c in cfg1 cfg3 cfg4 cfg5; do
    ( timeout 400 python bench.py --config $c --no-cpu --no-alt 2>&1 | tail -1 ) > gpurun_out/r4z_bench_$c.log
    python -c "import json; d=json.loads(open('gpurun_out/r4z_bench_$c.log').read()); r=d['roofline']; print('$c', d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['all_conv']['ms_per_step'], r['hbm_bound_calls']['ms_per_step'])" 2>&1 | tail -1
  done
  ( SEGMI_LOVASZ_SORT=rocprim timeout 300 python bench.py --config cfg5 --no-cpu --no-alt --no-roofline 2>&1 | tail -1 ) > gpurun_out/r4z_bench_cfg5_rocprim.log
  python -c "import json; d=json.loads(open('gpurun_out/r4z_bench_cfg5_rocprim.log').read()); print('cfg5 rocprim', d['value'], d['ms_per_step'])" 2>&1 | tail -1 ;;
r4y)
  # profiles of the final tree (after the Winograd threshold change): rocprof stats (in order / side stream), PMC traffic, tables
  rm -rf gpurun_out/prof
  ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python bench.py --steps 5 --warmup 2 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) > gpurun_out/prof.log
  find gpurun_out/prof -name "*kernel_trace*" -delete
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4y_kernel_stats_f32.csv
  rm -rf gpurun_out/prof
  ( SEGMI_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python bench.py --steps 5 --warmup 2 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) > gpurun_out/prof.log
  find gpurun_out/prof -name "*kernel_trace*" -delete
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4y_kernel_stats_f32_inorder.csv; head -3 gpurun_out/r4y_kernel_stats_f32_inorder.csv | cut -c1-200
  rm -rf gpurun_out/prof gpurun_out/pmc_f32
  ( timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f32/fetch -o r -- python bench.py --steps 1 --warmup 1 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) > gpurun_out/pmc_f32.log
  ( timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_f32/write -o r -- python bench.py --steps 1 --warmup 1 --no-cpu --no-roofline --no-alt 2>&1 | tail -2 ) >> gpurun_out/pmc_f32.log
  for d in fetch write; do f=$(find gpurun_out/pmc_f32/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/pmc_f32/$d/r_counter_collection.csv 2>/dev/null; done
  python tools/traffic_json.py gpurun_out/pmc_f32 gpurun_out/r4y_cfg2_conv_traffic_f32.json
  find gpurun_out/pmc_f32 -name "*kernel_trace*" -delete; find gpurun_out/pmc_f32 -name "*.csv" -size +8M -delete
  ( timeout 300 python tools/conv_layers.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r4y_conv_layers.txt; tail -1 gpurun_out/r4y_conv_layers.txt
  ( timeout 300 python tools/membound_ops.py cfg2 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r4y_membound_ops.txt; tail -2 gpurun_out/r4y_membound_ops.txt
  for c in cfg1 cfg5; do
    ( SEGMI_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$c -o r -- python bench.py --config $c --steps 7 --warmup 2 --no-cpu --no-roofline --no-alt 2>&1 | tail -1 ) > gpurun_out/r4y_prof_$c.log
    find gpurun_out/prof_$c -name "*kernel_trace*" -delete
    f=$(find gpurun_out/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4y_${c}_kernel_stats.csv
    rm -rf gpurun_out/prof_$c
  done ;;
esac
done
