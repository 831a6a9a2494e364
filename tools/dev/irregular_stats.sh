#!/bin/bash
# dev: the irregular panel's launch split into its one-wave kernel and the cooperative tail (rocprofv3 kernel stats)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_irr -o i --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_irregular.py > /tmp/irr.jsonl 2>/tmp/irr.err
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_irr/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) > 0.5:
        print('%-100s calls %4s avg %10.3f ms total %10.3f ms %6s %%' % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e6, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
PY
cut -c1-200 /tmp/irr.jsonl
