# tile-height rule re-checked on the new K loops: SEGMI_CONV_HALF_M unset (both rules) / 0 (never 64-row) / 3 (round-4 rule: small problems only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r06ae_half_m.txt
for i in 1 2; do for c in cfg3 cfg4 cfg5 cfg1; do for v in auto 0 3; do
 if [ $v = auto ]; then e=""; else e="SEGMI_CONV_HALF_M=$v"; fi
 r=$(env $e timeout 400 python bench.py --config $c --no-cpu --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
 echo "$c HALF_M=$v run $i: $r" | tee -a gpurun_out/r06ae_half_m.txt
done; done; done
