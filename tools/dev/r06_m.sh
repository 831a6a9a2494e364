#!/bin/bash
OUT=$PWD/gpurun_out/r06_m; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python tools/dev/map_probe.py cfg2:256 ref:64 cfg5:64 cfg4:16 2>&1 | tee $OUT/map_probe.txt | tail -12
python bench.py --no-cfg3 --no-other-configs --no-boundary --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
m=d['parity_context']['map_mode']
print('value', d['value'], 'map ms', m['ms_per_step'], m['status_counts'], m.get('forecast_max_rel_err_over_horizon_vs_true_map_map_mode'))
PY
