#!/bin/bash
# One gpurun call for experiments on the headline kernel (round 3): parity of the kernel routes, A/B of
# environment switches on the quadratic-form configurations, phase cycles (timing variant), bench, FETCH / WRITE.
# usage: gpurun --timeout 900 -- 'bash tools/gpu_quad_round.sh <tag> "VAR=val,VAR2=val ..."'
#   each space-separated item of the second argument is one A/B leg (comma-separated assignments; "-" = defaults)
TAG=${1:-r03_quad}
LEGS=${2:--}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest (quadratic-form kernels)" | tee $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "routes or quadratic or full_size_panel or full_size_other or randomised_model or ragged or design_and or odd_shapes or fit_predict" > $OUT/pytest_quad.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -4 $OUT/pytest_quad.log | tee -a $OUT/summary.txt
for leg in $LEGS; do
  echo "-- leg $leg" | tee -a $OUT/summary.txt
  ( if [ "$leg" != "-" ]; then for kv in ${leg//,/ }; do export "$kv"; done; fi
    timeout 300 python tools/bench_configs.py ${CONFIGS:-cfg2 cfg3 cfg2x4 cfg2x16} 2>$OUT/configs_$leg.err | tee $OUT/configs_$leg.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], 'ms/step %.3f fit-kernel ms %.3f series/s %.0f' % (d['ms_per_step'], d['fit_kernel_ms'], d['series_per_s']))" ) | tee -a $OUT/summary.txt
done
if [ -f tools/variants/libtsf_amd_qtime.so ]; then
echo "== phase cycles (timing variant)" | tee -a $OUT/summary.txt
for leg in $LEGS; do
  echo "-- leg $leg" | tee -a $OUT/summary.txt
  ( if [ "$leg" != "-" ]; then for kv in ${leg//,/ }; do export "$kv"; done; fi
    TSF_LIB_PATH=$PWD/tools/variants/libtsf_amd_qtime.so timeout 300 python tools/bench_configs.py cfg2 cfg2x4 2>&1 >/dev/null | grep quad-timing | tail -4 ) | tee -a $OUT/summary.txt
done
fi
echo "== bench" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python -c "
import json; d = json.load(open('$OUT/bench.json')); print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms_avg'], 'oracle err', d.get('forecast_max_rel_err_vs_oracle'))" | tee -a $OUT/summary.txt
if [ -z "$SKIP_PROF" ]; then
echo "== rocprofv3 pmc FETCH_SIZE / WRITE_SIZE" | tee -a $OUT/summary.txt
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_fetch -o fetch --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --timed-only > $OUT/prof_fetch.log 2>&1 ); echo "pmc fetch rc=$?" | tee -a $OUT/summary.txt
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_write -o write --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --timed-only > $OUT/prof_write.log 2>&1 ); echo "pmc write rc=$?" | tee -a $OUT/summary.txt
python tools/pmc_summary.py $OUT | grep -i "quad\|--" | tee -a $OUT/summary.txt
fi
find $OUT -name '*.db' -delete 2>/dev/null
find $OUT -size +4M -delete 2>/dev/null
echo done
