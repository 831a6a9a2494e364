export TMPDIR=/tmp
mkdir -p gpurun_out/exp4
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
for leg in "-" "TSF_FIT_GROUPED=0"; do
  echo "-- leg $leg"
  ( if [ "$leg" != "-" ]; then export $leg; fi
    python tools/bench_configs.py cfg4 lin_hol_resid 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], 'fit-kernel ms %.3f series/s %.0f evals/s %.0f mean evals %.1f max %d' % (d['fit_kernel_ms'], d['series_per_s'], d['evals_per_s'], d['mean_evals'], d['max_evals']))" )
done
) 2>&1 | tee gpurun_out/exp4/summary.txt
