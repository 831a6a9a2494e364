#!/bin/bash
# round 5, third GPU call: the async-scratch hunt, the cooperative kernel's phase counters, sparse base-pair kernel at 3 waves
TAG=${1:-r05_c}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/dev/async_scratch_hunt.sh $TAG
TSF_LIB_PATH=$PWD/tools/variants/libtsf_amd_ct.so timeout 300 python tools/bench_configs.py ref10k > $OUT/ct.jsonl 2> $OUT/ct.err; grep coop-timing $OUT/ct.err | tail -2 | tee -a $OUT/summary.txt
TSF_LIB_PATH=$PWD/tools/variants/libtsf_amd_sp3.so timeout 300 python tools/bench_configs.py cfg4 > $OUT/sp3.jsonl 2> $OUT/sp3.err; python - <<PY | tee -a $OUT/summary.txt
import json
for l in open('$OUT/sp3.jsonl'):
    d = json.loads(l); print('sp3', d['config'], 'fit-kernel %.1f ms' % d['fit_kernel_ms'], '%.1f M evals/s' % (d['evals_per_s'] / 1e6), 'max evals', d['max_evals'])
PY
echo done
