#!/bin/bash
# dev: where a 10 000-series launch of the reference's model and cfg4 spend their time -- one-wave kernel against the
# cooperative tail (rocprofv3 kernel stats of tools/bench_configs.py per configuration)
TAG=${1:-r05_g}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in cfg4 ref10k; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$c -o $c --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_configs.py $c > $OUT/$c.jsonl 2> $OUT/$c.err
  f=$(ls $OUT/prof_$c/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $c" | tee -a $OUT/summary.txt
  python - "$f" <<'PY' | tee -a $OUT/summary.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage']) > 0.2:
        print('%-110s calls %4s avg %10.3f ms total %10.3f ms %6s %%' % (r['Name'][:110], r['Calls'], float(r['AverageNs']) / 1e6, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
PY
  cut -c1-400 $OUT/$c.jsonl | tee -a $OUT/summary.txt
done
echo done
