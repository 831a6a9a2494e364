#!/bin/bash
OUT=$PWD/gpurun_out/r06_h; mkdir -p $OUT; export TMPDIR=/tmp
python bench.py --no-cfg3 --no-other-configs --no-boundary --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms_avg'])
print(json.dumps(d['parity_context'].get('map_mode'), indent=1))
PY
