cd /tmp && export TMPDIR=/tmp
ONLY=logistic timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ro -o ro --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_ragged.py > /tmp/ro.jsonl 2>/tmp/ro.err
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_ro/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) > 0.5:
        print('%-100s calls %4s avg %10.3f ms total %10.3f ms %6s %%' % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e6, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
PY
cut -c60-240 /tmp/ro.jsonl
f=$(ls /tmp/prof_ro/*kernel_trace.csv | head -1); python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'fit_' in r['Kernel_Name']]
for r in rows: print(r['Kernel_Name'][:70], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, 'ms', 'grid', r.get('Grid_Size'), r.get('Workgroup_Size'))
PY
