#!/bin/bash
OUT=$PWD/gpurun_out/r06_p; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "newton or cfg5 or Newton" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 4 $OUT/pytest_gpu.log
timeout 900 python tools/bench_configs.py cfg5_newton cfg5_newton_1m > $OUT/configs.jsonl 2> $OUT/configs.err; python - <<PY
import json
for l in open('$OUT/configs.jsonl'):
    try: d = json.loads(l)
    except Exception: continue
    print('%-16s fit-kernel %9.3f ms  %9.0f series/s  evals mean %.0f max %.0f  %s' % (d.get('config'), d.get('fit_kernel_ms', -1), d.get('series_per_s', -1), d.get('mean_evals', -1), d.get('max_evals', -1), d.get('status_counts')))
PY
