#!/bin/bash
OUT=$PWD/gpurun_out/r06_l; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python tools/dev/coop_after_probe.py ${1:-ref10k,cfg4,cfg1} ${2:--1,300,600,1000,1500,2500,4000} 2>&1 | tee $OUT/coop_after.txt | tail -40
