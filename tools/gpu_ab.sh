#!/bin/bash
# A/B of kernel variants in ONE gpurun call: for every library given (default build first) the bench line twice and
# the HBM traffic counters of the headline kernel.  usage: bash tools/gpu_ab.sh <tag> [variant-tag ...]
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for V in default "$@"; do
  if [ "$V" = default ]; then unset TSF_LIB_PATH; else export TSF_LIB_PATH=$PWD/tools/variants/libtsf_amd_$V.so; fi
  for i in 1 2 3; do
    timeout 600 python bench.py --steps 5 --warmup 2 --no-cfg3 --no-cpu-baseline > $OUT/bench_${V}_$i.json 2> $OUT/bench_${V}_$i.err
    python - <<PY | tee -a $OUT/summary.txt
import json
d=json.load(open('$OUT/bench_${V}_$i.json'))
print('$V: value %.0f ms_per_step %.3f kernel_ms %.3f hinted %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d.get('with_cost_hints',{}).get('ms_per_step',-1)))
PY
  done
  for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $OUT/prof_${V}_$C -o p --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --timed-only > $OUT/prof_${V}_$C.log 2>&1 )
  python - <<PY | tee -a $OUT/summary.txt
import csv, collections, glob
per = collections.defaultdict(float); n=set()
for f in glob.glob('$OUT/prof_${V}_$C/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'fit_quad_kernel' in r['Kernel_Name']:
            per[r['Counter_Name']] += float(r['Counter_Value']); n.add(r['Dispatch_Id'])
for c, v in per.items(): print('$V: fit_quad_kernel %s per launch %.1f KiB' % (c, v / max(len(n),1)))
PY
  done
done
find $OUT -name '*.db' -delete 2>/dev/null
