#!/bin/bash
# dev: which kernels the files -> files run of the reference's own model launches, and for how long
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_e2e -o e --output-format csv -- python $GRAFT_REPO_ROOT/tools/e2e_bench.py --kind reference > /tmp/e2e_ref.txt 2>/tmp/e2e_ref.err
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_e2e/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) > 0.3:
        print('%-100s calls %4s avg %10.3f ms total %10.3f ms %6s %%' % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e6, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
PY
tail -9 /tmp/e2e_ref.txt | cut -c1-200
