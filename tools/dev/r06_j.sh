#!/bin/bash
# round 6: HBM counters of one step (every kernel) after setup_series reads y once and the quadratic-form fit reads the caller's rows
OUT=$PWD/gpurun_out/r06_j; mkdir -p $OUT; export TMPDIR=/tmp
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
