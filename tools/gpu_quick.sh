#!/bin/bash
# One gpurun call while iterating on the headline kernel: the quadratic-form parity tests, the bench line and the
# SQ instruction counters (VALU / SALU / LDS instructions per launch).  Everything lands under gpurun_out/<tag>/.
# usage: gpurun --timeout 900 -- 'bash tools/gpu_quick.sh r04_x [pytest -k expression]'
TAG=${1:-quick}
KEXPR=${2:-"quad or full_size_panel or full_size_other or cfg3 or literal or odd_shapes or ragged or truncated or golden or cost_hints or edge"}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
for i in 1 2; do
timeout 600 python bench.py --steps 5 --warmup 2 --no-cfg3 --no-cpu-baseline --no-other-configs > $OUT/bench$i.json 2> $OUT/bench$i.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
d=json.load(open('$OUT/bench$i.json'))
print('value %.0f ms_per_step %.3f kernel_ms %.3f hinted %.3f ms host-pointer %.0f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d.get('with_cost_hints',{}).get('ms_per_step',-1), d.get('value_end_to_end_host_pointer',-1)))
PY
done
if [ -z "$SKIP_CFG" ]; then
timeout 900 python tools/bench_configs.py ${CFGS:-cfg2x16 cfg5} > $OUT/configs.jsonl 2> $OUT/configs.err
python - <<PY | tee -a $OUT/summary.txt
import json
for l in open('$OUT/configs.jsonl'):
    try:
        d = json.loads(l)
    except Exception:
        continue
    print('%-12s fit-kernel %.3f ms  %.0f series/s  evals mean %.0f max %.0f' % (d.get('config'), d.get('fit_kernel_ms', -1), d.get('series_per_s', -1), d.get('mean_evals', -1), d.get('max_evals', -1)))
PY
fi
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --kernel-trace -d $OUT/prof_sq -o sq --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --timed-only > $OUT/prof_sq.log 2>&1 ); echo "pmc sq rc=$?" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import csv, collections, glob
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('$OUT/prof_sq/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'fit_quad_kernel' in r['Kernel_Name']:
            per[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
for c, d in sorted(per.items()):
    print('fit_quad_kernel %-22s per launch %.4g (n=%d)' % (c, sum(d.values()) / len(d), len(d)))
PY
find $OUT -name '*.db' -delete 2>/dev/null
find $OUT -size +4M -delete 2>/dev/null
echo done
