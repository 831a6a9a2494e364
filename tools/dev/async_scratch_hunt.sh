#!/bin/bash
# Round 5: the hunt for round 3's wrong Newton fits behind stream-ordered scratch.  (1) the standalone probe (no library
# code), both pool modes; (2) tools/dev/nb_debug.py on the library with the old path re-enabled (TSF_OPT_DEBUG_ASYNC_SCRATCH):
# plain, + canaries, + synchronise before the free, + a pool that never releases; and on the cached block (default).
# usage: gpurun --timeout 900 -- 'bash tools/dev/async_scratch_hunt.sh r05_async'
TAG=${1:-r05_async}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for m in 0 1; do timeout 300 tools/probes/bin/mallocasync_probe $m > $OUT/probe_mode$m.txt 2>&1; echo "probe mode $m rc=$?" | tee -a $OUT/summary.txt; tail -1 $OUT/probe_mode$m.txt | tee -a $OUT/summary.txt; done
for opt in 0 1 5 3 9 1; do
  echo "== debug_async_scratch=$opt" | tee -a $OUT/summary.txt
  TSF_OPTIONS=debug_async_scratch=$opt BIG_N=${BIG_N:-100000} timeout 400 python tools/dev/nb_debug.py > $OUT/nb_$opt.txt 2> $OUT/nb_$opt.err; echo "rc=$?" | tee -a $OUT/summary.txt
  cat $OUT/nb_$opt.txt | tee -a $OUT/summary.txt; grep async-scratch $OUT/nb_$opt.err | sort | uniq -c | tee -a $OUT/summary.txt
done
echo done
