#!/bin/bash
# rocprofv3 kernel stats + HBM counters of the predict / interval kernels (tools/predict_profile.py)
TAG=${1:-r03_predict}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats --output-format csv -- python $R/tools/predict_profile.py > $OUT/prof_stats.log 2>&1 ); echo "stats rc=$?"
find $OUT/prof_stats -name '*kernel_stats.csv' | head -1 | xargs -r head -14
( cd /tmp && WITH_INTERVALS=1 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_fetch -o fetch --output-format csv -- python $R/tools/predict_profile.py > $OUT/prof_fetch.log 2>&1 ); echo "fetch rc=$?"
( cd /tmp && WITH_INTERVALS=1 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_write -o write --output-format csv -- python $R/tools/predict_profile.py > $OUT/prof_write.log 2>&1 ); echo "write rc=$?"
python tools/pmc_summary.py $OUT 2>/dev/null | grep -v "^wrote" | tee $OUT/pmc_summary.txt
find $OUT -name '*.db' -delete 2>/dev/null
find $OUT -size +2M -delete 2>/dev/null
