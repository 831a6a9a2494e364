#!/bin/bash
TAG=${1:-r05_f}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
for m in 16 16 18 1 1 1 3 0; do
  f=$OUT/probe_mode${m}_$RANDOM.txt
  timeout 300 tools/probes/bin/mallocasync_probe $m > $f 2>&1; rc=$?
  echo "mode $m rc=$rc: $(tail -1 $f) | corrupted calls: $(grep -v 'rounds 0, inside the rounds 0' $f | grep -c 'GB:') sizes: $(grep -v 'rounds 0, inside the rounds 0' $f | grep 'GB:' | sed 's/.*size *\([0-9.]*\) GB.*/\1/' | sort | uniq -c | tr '\n' ' ')" | tee -a $OUT/summary.txt
done
for v in nw12; do TSF_LIB_PATH=$PWD/tools/variants/libtsf_amd_$v.so timeout 300 python tools/bench_configs.py ref10k 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', d['config'], 'fit-kernel %.1f ms' % d['fit_kernel_ms'], 'max evals', d['max_evals'])" | tee -a $OUT/summary.txt; done
echo done
