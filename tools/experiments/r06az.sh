cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/tr; SEGMI_SGD_TABLE_UPLOAD=throttled timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o r -- python bench.py --config cfg2 --steps 9 --warmup 5 --no-cpu --no-roofline --no-alt > /tmp/tr.full 2>&1
grep '^{' /tmp/tr.full | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['sgd_table_upload'])"
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python tools/gpu_gaps.py $f | sed -n '1,3p;/largest gaps/,+6p;/idle time by/,+4p'
python - <<'P'
import csv,glob
f=glob.glob('/tmp/tr/**/*kernel_trace.csv',recursive=True)[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:60],r.get('Queue_Id'),r.get('Stream_Id')) for r in csv.DictReader(open(f))]
rows.sort()
idx=[i for i,r in enumerate(rows) if 'sgd_multi' in r[2]]
i=idx[-3]
for r in rows[i-2:i+5]: print((r[0]-rows[i][0])/1e3,(r[1]-rows[i][0])/1e3,r[2],r[3],r[4])
P
