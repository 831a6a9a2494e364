#!/bin/bash
OUT=$PWD/gpurun_out/r06_u; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -8
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 5 $OUT/pytest_gpu.log
timeout 900 python tools/dev/map_direct_timing.py cfg5 > $OUT/map_direct_timing_cfg5.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $OUT/map_direct_timing_cfg5.txt | cut -c1-1300
