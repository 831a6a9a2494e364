#!/bin/bash
# dev round for the cooperative tail: its parity tests, then timings with and without it
TAG=${1:-coop}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cooperative" > $OUT/pytest_coop.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
tail -15 $OUT/pytest_coop.log | tee -a $OUT/summary.txt
for c in ${CONFIGS:-cfg1 cfg1_wave ref10k ref10k_wave ref100k ref100k_wave cfg4 cfg4_wave}; do
  timeout 600 python tools/bench_configs.py $c >> $OUT/configs.jsonl 2>> $OUT/configs.err; echo "$c rc=$?" | tee -a $OUT/summary.txt
done
cat $OUT/configs.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['config'], 'fit_ms', round(d['fit_kernel_ms'], 2), 'series/s', round(d['series_per_s']), 'mean/max evals', round(d['mean_evals']), d['max_evals'], d['status_counts'])
" | tee -a $OUT/summary.txt
