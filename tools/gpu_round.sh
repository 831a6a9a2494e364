#!/bin/bash
# One gpurun call: GPU parity tests, bench line, rocprofv3 kernel stats, and (separately) the
# PMC passes for HBM traffic.  Everything lands under gpurun_out/<tag>/.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r01_v2'
TAG=${1:-run}
STEPS=${STEPS:-3}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $OUT/summary.txt
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
fi
echo "== bench" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps $STEPS --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench.json | tee -a $OUT/summary.txt
if [ -z "$SKIP_PROF" ]; then
echo "== rocprofv3 kernel stats" | tee -a $OUT/summary.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats --output-format csv -- python $OLDPWD/bench.py --steps $STEPS --warmup 1 --timed-only > $OUT/prof_stats.log 2>&1 ); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof_stats -name '*kernel_stats.csv' | head -1 | xargs -r head -12 | tee -a $OUT/summary.txt
echo "== rocprofv3 pmc FETCH_SIZE" | tee -a $OUT/summary.txt
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_fetch -o fetch --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --timed-only > $OUT/prof_fetch.log 2>&1 ); echo "pmc fetch rc=$?" | tee -a $OUT/summary.txt
( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_write -o write --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --timed-only > $OUT/prof_write.log 2>&1 ); echo "pmc write rc=$?" | tee -a $OUT/summary.txt
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace -d $OUT/prof_sq -o sq --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --timed-only > $OUT/prof_sq.log 2>&1 ); echo "pmc sq rc=$?" | tee -a $OUT/summary.txt
python tools/pmc_summary.py $OUT | tee -a $OUT/summary.txt
# keep only the small csv files
find $OUT -name '*.db' -delete 2>/dev/null
find $OUT -size +4M -delete 2>/dev/null
fi
echo done
