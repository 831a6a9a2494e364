#!/bin/bash
OUT=$PWD/gpurun_out/r06_u; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python tools/dev/map_probe.py cfg2:256 cfg5:64 > $OUT/map_probe.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $OUT/map_probe.txt | cut -c1-500
TSF_OPTIONS=map_direct=0 timeout 900 python tools/dev/map_probe.py cfg2:256 > $OUT/map_probe_cont.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $OUT/map_probe_cont.txt | cut -c1-500
