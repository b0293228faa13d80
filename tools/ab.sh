#!/bin/bash
# A/B of an environment switch inside ONE gpurun call (boxes differ by +-1 %, the first bench of a call is ~0.7 ms slower: alternate and
# compare equal positions).   usage: tools/ab.sh <cfg> <ENVVAR> <valueA> <valueB> [rounds] [extra bench args...]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
cfg=$1; var=$2; a=$3; b=$4; n=${5:-2}; shift 5 2>/dev/null
out=gpurun_out/${TAG:-r06}_ab_${cfg}_${var}.txt
: > $out
for i in $(seq 1 $n); do
  for v in $a $b; do
    r=$(env $var=$v timeout 400 python bench.py --config $cfg --no-cpu --no-alt --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$cfg $var=$v round $i: $r" | tee -a $out
  done
done
