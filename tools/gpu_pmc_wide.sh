#!/bin/bash
# SQ counters of the quadratic-form kernel in the throughput regime (cfg2 model on 160 000 series: the 16-wave
# kernel) and on the headline panel (12-wave kernel): who is busy, who waits.  usage: bash tools/gpu_pmc_wide.sh <tag>
TAG=${1:-pmcw}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_IFETCH"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
for CFG in cfg2x16 cfg2; do
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    ( cd /tmp && timeout 600 rocprofv3 --pmc $P --kernel-trace -d $OUT/${CFG}_p$i -o p --output-format csv -- python $OLDPWD/tools/bench_configs.py $CFG > $OUT/${CFG}_p$i.log 2>&1 )
  done
  python - <<PY | tee -a $OUT/summary.txt
import csv, collections, glob
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('$OUT/${CFG}_p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'fit_quad_kernel' in r['Kernel_Name']:
            per[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
print('== $CFG')
for c, d in sorted(per.items()):
    print('%-28s per launch %.5g (n=%d)' % (c, sum(d.values()) / len(d), len(d)))
PY
done
find $OUT -name '*.db' -delete 2>/dev/null; find $OUT -size +4M -delete 2>/dev/null
