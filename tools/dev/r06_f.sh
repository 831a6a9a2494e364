#!/bin/bash
OUT=$PWD/gpurun_out/r06_f; mkdir -p $OUT; export TMPDIR=/tmp
( time python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/time.txt; echo "bench rc=$?"; tail -3 $OUT/time.txt; tail -5 $OUT/bench.err
python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'])
print(json.dumps(d.get('boundary'), indent=1)[:3000])
o=d.get('other_baseline_configs',{})
for k in o: print(k, {kk:o[k].get(kk) for kk in ('series_per_s','ms_per_step','fit_kernel_ms','max_evals','longest_fit_ms_over_launch_ms','error')})
print(d.get('ranks_seen')); print(d.get('in_process_device_split'))
PY
# 4 gloo ranks on the one GPU: the multi-rank path with the new fields
TSF_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 4 --steps 3 --warmup 1 --no-cfg3 > $OUT/bench_4ranks_gloo.json 2> $OUT/bench_4ranks.err; echo "4 ranks rc=$?"
python - <<PY
import json
d=json.load(open('$OUT/bench_4ranks_gloo.json'))
print('4 ranks value', d['value'], d['ranks_seen']['world_size'], d['ranks_seen']['backend'], d['ranks_seen']['distinct_gpus'], [r['uuid'] for r in d['ranks_seen']['ranks']])
PY
