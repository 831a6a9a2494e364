#!/bin/bash
# round 5: time slicing of the quadratic-form kernel -- parity tests, then the bench line with the switch off / on / other quanta
TAG=${1:-r05_e}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "${KEXPR:-time_slicing or quad or full_size or golden or cost_hints or literal}" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
tail -4 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
for q in 0 -1 128 512 64; do
  for i in 1 2; do
    TSF_OPTIONS=quad_yield=$q timeout 600 python bench.py --steps 5 --warmup 2 --no-cfg3 --no-cpu-baseline --no-other-configs > $OUT/bench_q${q}_$i.json 2> $OUT/bench_q${q}_$i.err
    python - <<PY | tee -a $OUT/summary.txt
import json
try:
    d=json.load(open('$OUT/bench_q${q}_$i.json'))
    print('quad_yield=$q: value %.0f ms_per_step %.3f kernel_ms %.3f hinted %.3f ms host-pointer %.0f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d.get('with_cost_hints',{}).get('ms_per_step',-1), d.get('value_end_to_end_host_pointer',-1)))
except Exception as e:
    print('quad_yield=$q: failed', e)
PY
  done
done
TSF_OPTIONS=quad_yield=0 timeout 600 python tools/bench_configs.py cfg2x4 cfg3 cfg5 > $OUT/configs_off.jsonl 2> $OUT/configs_off.err
timeout 600 python tools/bench_configs.py cfg2x4 cfg3 cfg5 > $OUT/configs_on.jsonl 2> $OUT/configs_on.err
python - <<PY | tee -a $OUT/summary.txt
import json
for tag in ('off', 'on'):
    for l in open('$OUT/configs_%s.jsonl' % tag):
        try: d = json.loads(l)
        except Exception: continue
        print('yield %-3s %-8s fit-kernel %.3f ms  %.0f series/s' % (tag, d['config'], d['fit_kernel_ms'], d['series_per_s']))
PY
echo done
