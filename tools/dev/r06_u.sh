#!/bin/bash
OUT=$PWD/gpurun_out/r06_u; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_literal.py -m gpu -x -q -k "degenerate or ragged_panels or map_mode" > $OUT/pytest_deg.log 2>&1; echo "pytest rc=$?"; tail -n 12 $OUT/pytest_deg.log | cut -c1-300
timeout 900 python tools/dev/map_direct_timing.py > $OUT/map_direct_timing3.txt 2>&1; grep -v amdgpu.ids $OUT/map_direct_timing3.txt | cut -c1-900
timeout 600 python tools/dev/route_stress.py 120 11 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-600
