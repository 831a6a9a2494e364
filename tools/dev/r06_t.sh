#!/bin/bash
# GPU suite (or -k subset) into gpurun_out/<tag>/pytest_gpu.log
TAG=${1:-r06_t}; K=${2:-}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ -n "$K" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$K" > $OUT/pytest_gpu.log 2>&1; else timeout 1700 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; fi
echo "pytest rc=$?"; tail -n 25 $OUT/pytest_gpu.log
