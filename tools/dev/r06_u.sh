#!/bin/bash
OUT=$PWD/gpurun_out/r06_u; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python tools/dev/route_stress.py 400 7 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-1200
timeout 600 python -m pytest tests/test_gpu_literal.py -m gpu -x -q -k "degenerate or ragged_panels or map_mode" > $OUT/pytest_deg.log 2>&1; echo "pytest rc=$?"; tail -n 12 $OUT/pytest_deg.log | cut -c1-300
