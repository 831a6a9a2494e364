#!/bin/bash
# round 6: the quadratic-form kernels on the caller's y rows (no scaled copy): parity subset, bench with the route on / off
OUT=$PWD/gpurun_out/r06_k; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "quad or full_size_panel or cfg3 or literal or odd_shapes or ragged or truncated or golden or cost_hints or edge or y_dtype or dtype" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest_gpu.log
for leg in "" "quad_raw_y=0" "" "quad_raw_y=0"; do
TSF_OPTIONS=$leg python bench.py --steps 10 --warmup 3 --no-cfg3 --no-other-configs --no-boundary --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print('[$leg] value %.0f ms_per_step %.3f kernel_ms %.3f hinted %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d.get('with_cost_hints',{}).get('ms_per_step',-1)))
PY
done
# round 6: HBM counters of one step (every kernel) after setup_series reads y once and the quadratic-form fit reads the caller's rows
OUT=$PWD/gpurun_out/r06_k; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_fetch -o fetch --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --timed-only > $OUT/prof_fetch.log 2>&1 ); echo "pmc fetch rc=$?"
( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_write -o write --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --timed-only > $OUT/prof_write.log 2>&1 ); echo "pmc write rc=$?"
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1; tail -25 $OUT/pmc_summary.txt
python - <<PY
import json
d=json.load(open('$OUT/pmc_latest.json'))
tot=0
for k,v in d['step_kernels'].items():
    b=(2*v.get('FETCH_SIZE_KiB',0)+v.get('WRITE_SIZE_KiB',0))*1024; tot+=b
    print('%-28s fetch x2 %8.1f MB write %8.1f MB' % (k, 2*v.get('FETCH_SIZE_KiB',0)*1024/1e6, v.get('WRITE_SIZE_KiB',0)*1024/1e6))
print('step total %.1f MB = %.2f x algorithmic (69.92 MB)' % (tot/1e6, tot/69.92e6))
PY
find $OUT -name '*.db' -delete 2>/dev/null; find $OUT -size +4M -delete 2>/dev/null
