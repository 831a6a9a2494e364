#!/bin/bash
# rocprofv3 evidence for the matrix-core residual kernel next to the one-wave kernel on the same
# saturated workload (100 000 x 730, reference settings, max_iter 150): kernel stats, MFMA
# instruction counters, L1->L2 read requests.  usage: gpurun -- 'bash tools/gpu_mfma_profile.sh r02_mfma'
TAG=${1:-mfma}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o -E "Name:\s*[A-Za-z0-9_]*(MFMA|TCC_READ|TCP_TCC_READ|TCC_REQ|TCC_HIT|TCC_MISS)[A-Za-z0-9_]*" | sort -u > $OUT/avail.txt
for K in wave mfma; do
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_$K -o s --output-format csv -- python $OLDPWD/tools/bench_configs.py cap100000_$K > $OUT/stats_$K.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_mfma_$K -o p --output-format csv -- python $OLDPWD/tools/bench_configs.py cap100000_$K > $OUT/pmc_mfma_$K.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_l2_$K -o p --output-format csv -- python $OLDPWD/tools/bench_configs.py cap100000_$K > $OUT/pmc_l2_$K.log 2>&1 )
done
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for f in sorted(glob.glob(os.path.join(out, '*', '**', '*counter_collection.csv'), recursive=True)) + sorted(glob.glob(os.path.join(out, '*', '**', '*kernel_stats.csv'), recursive=True)):
    print('--', os.path.relpath(f, out))
    if f.endswith('kernel_stats.csv'):
        for i, l in enumerate(open(f)):
            if i < 4: print(l.strip()[:170])
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:48]
        if 'fit_' in k: per[(k, r['Counter_Name'])][r['Dispatch_Id']] += float(r['Counter_Value'])
    for (k, c), d in sorted(per.items()):
        print('%-50s %-30s n=%d mean=%.6g' % (k, c, len(d), sum(d.values()) / len(d)))
PY
find $OUT -name '*.db' -delete 2>/dev/null; find $OUT -size +4M -delete 2>/dev/null
echo done
