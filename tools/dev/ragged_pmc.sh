#!/bin/bash
# dev: HBM read traffic (FETCH_SIZE) of the residual-form one-wave kernel on the ragged bench panel, every series on
# its own grid (TSF_GRID_SHARE=0) and on shared grids -- is the own-grid case bound by HBM?
# usage: gpurun -- 'bash tools/dev/ragged_pmc.sh r04_ragged_pmc'
TAG=${1:-ragged_pmc}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for share in 0 1; do
( cd /tmp && TSF_OPTIONS=grid_share=$share timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_fetch_share$share -o f --output-format csv -- python $OLDPWD/tools/bench_ragged.py > $OUT/run_share$share.log 2>&1 ); echo "share=$share rc=$?" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(dict)
for f in glob.glob('$OUT/prof_fetch_share$share/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:70]
        if 'fit_kernel' in k or 'fit_quad' in k:
            per[k][r['Dispatch_Id']] += float(r['Counter_Value'])
for k, d in per.items():
    v = sorted(d.values())
    print('share=$share %-72s launches %d FETCH_SIZE KiB median %.0f  (= %.1f MB)' % (k, len(v), v[len(v)//2], v[len(v)//2] * 1024 / 1e6))
PY
grep -h "grids" $OUT/run_share$share.log | cut -c60-230 | tee -a $OUT/summary.txt
done
find $OUT -name '*.db' -delete 2>/dev/null
find $OUT -size +4M -delete 2>/dev/null
