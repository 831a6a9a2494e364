#!/bin/bash
OUT=$PWD/gpurun_out/r06_s; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python tools/dev/coop_live_probe.py cfg4,ref100k,ref10k 0,8,16 1500,2500 > $OUT/probe.txt 2>&1; echo "rc=$?"
cat $OUT/probe.txt | grep -v amdgpu.ids
