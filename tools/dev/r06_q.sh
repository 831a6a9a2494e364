#!/bin/bash
# round 6: phase cycles of the several-series-per-wave Newton kernel after the two reformulations (timing build), and the three sizes
OUT=$PWD/gpurun_out/r06_q; mkdir -p $OUT; export TMPDIR=/tmp
for n in 20000 100000 1000000; do python tools/newton_timing.py $n 2>&1 | tail -1; done > $OUT/timing.txt
echo "phase timing (timing build)" >> $OUT/timing.txt
TSF_LIB_PATH=$PWD/tools/variants/libtsf_amd_nbtime.so python tools/newton_timing.py 100000 2>&1 | grep "newton-batch-timing\|newton-timing" >> $OUT/timing.txt
cat $OUT/timing.txt | cut -c1-700
