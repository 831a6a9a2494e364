#!/bin/bash
OUT=$PWD/gpurun_out/r06_r; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "newton or cfg5 or Newton" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest_gpu.log
for n in 20000 100000 1000000; do python tools/newton_timing.py $n 2>&1 | tail -1; done > $OUT/timing.txt
echo "phase timing (timing build)" >> $OUT/timing.txt
TSF_LIB_PATH=$PWD/tools/variants/libtsf_amd_nbtime.so python tools/newton_timing.py 100000 2>&1 | grep "newton-batch-timing\|newton-timing" >> $OUT/timing.txt
python - <<PY
import json
for l in open('$OUT/timing.txt'):
    if l.startswith('{'):
        d=json.loads(l); print(d['series'], 'newton %.0f series/s (%.2f s), lbfgs %.0f' % (d['newton']['series_per_s'], d['newton']['seconds_host_call'], d['lbfgs']['series_per_s']))
    else: print(l.strip()[:600])
PY
