#!/bin/bash
# round 6: the quadratic-form kernels on the caller's y rows (no scaled copy): parity subset, bench with the route on / off
OUT=$PWD/gpurun_out/r06_i; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "quad or full_size_panel or cfg3 or literal or odd_shapes or ragged or truncated or golden or cost_hints or edge or y_dtype or dtype" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest_gpu.log
for leg in "" "quad_raw_y=0" "" "quad_raw_y=0"; do
TSF_OPTIONS=$leg python bench.py --steps 10 --warmup 3 --no-cfg3 --no-other-configs --no-boundary --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print('[$leg] value %.0f ms_per_step %.3f kernel_ms %.3f hinted %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d.get('with_cost_hints',{}).get('ms_per_step',-1)))
PY
done
