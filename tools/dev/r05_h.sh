#!/bin/bash
# dev: Z^T Z of a ragged quadratic-form panel with a calendar per series -- three columns per pass from the rows' base
# pairs (default) against two columns per pass from the design tables (harm=0): tests, kernel time, FETCH_SIZE
TAG=${1:-r05_h}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "ragged or share or fixture" 2>&1 | tail -4 | tee -a $OUT/summary.txt
for h in -1 0 -1 0; do
  echo "== harm=$h" | tee -a $OUT/summary.txt
  ONLY=linear TSF_OPTIONS=harm=$h timeout 300 python tools/bench_ragged.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['grids'], d['growth'], ['%.2f' % v for v in d['fit_kernel_ms']], 'mean evals %.1f' % d['mean_evals'])" | tee -a $OUT/summary.txt
done
export TMPDIR=/tmp
for h in -1 0; do
( cd /tmp && ONLY=linear TSF_OPTIONS=harm=$h timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_fetch_harm$h -o f --output-format csv -- python $R/tools/bench_ragged.py > $OUT/run_harm$h.log 2>&1 ); echo "harm=$h rc=$?" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('$OUT/prof_fetch_harm$h/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:70]
        if 'fit_quad' in k or 'gram' in k:
            per[k][r['Dispatch_Id']] += float(r['Counter_Value'])
for k, d in per.items():
    v = list(d.values())
    print('harm=$h %-72s launches %d FETCH_SIZE KiB per launch: %s' % (k, len(v), ' '.join('%.0f' % x for x in v)))
PY
done
find $OUT -name '*.db' -delete 2>/dev/null
find $OUT -size +4M -delete 2>/dev/null
echo done
