#!/bin/bash
OUT=$PWD/gpurun_out/r06_t2; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "lattice or ragged or cooperative_tail or fixture or prefetch or map_mode" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 15 $OUT/pytest_gpu.log
