#!/bin/bash
# FETCH_SIZE calibration for the library's load shapes (tools/probes/fetch_calib.hip): factor = reported / streamed.
# usage: gpurun -- 'bash tools/dev/fetch_calib.sh r05_fetch_calib'
TAG=${1:-r05_fetch_calib}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof -o c --output-format csv -- $OLDPWD/tools/probes/bin/fetch_calib > $OUT/run.log 2>&1 ); echo "rc=$?" | tee $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import csv, glob, collections, json
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('$OUT/prof/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        per[r['Kernel_Name'].split('(')[0]][r['Dispatch_Id']] += float(r['Counter_Value'])
streamed = {'k_b8': 4 << 30, 'k_b16': 4 << 30, 'k_b2': 4 << 30, 'k_b8s': ((4 << 30) // (28 * 64 * 8)) * 28 * 64 * 8}
cal = {}
for k, d in sorted(per.items()):
    name = k.split()[-1] if ' ' in k else k
    v = sorted(d.values()); med = v[len(v) // 2] * 1024.0
    s = streamed.get(name)
    if s:
        cal[name] = med / s
        print('%-8s FETCH_SIZE %.4f GB for %.4f GB streamed: counter / bytes = %.4f  (launches %d)' % (name, med / 1e9, s / 1e9, med / s, len(v)))
json.dump({'what': 'FETCH_SIZE (KiB x 1024) / bytes streamed once from HBM, per load shape (tools/probes/fetch_calib.hip)', 'counter_per_byte': cal}, open('$OUT/calibration.json', 'w'), indent=1)
PY
find $OUT -name '*.db' -delete 2>/dev/null
echo done
