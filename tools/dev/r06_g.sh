#!/bin/bash
OUT=$PWD/gpurun_out/r06_g; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python tools/dev/map_probe.py ${@:-cfg2:32 ref:16 cfg5:32 cfg4:8} 2>&1 | tee $OUT/map_probe.txt | tail -30
