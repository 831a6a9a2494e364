#!/bin/bash
OUT=$PWD/gpurun_out/r06_t2; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "lattice or ragged or cooperative_tail or fixture or prefetch or odd_shapes or randomised_model" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 15 $OUT/pytest_gpu.log
timeout 900 python tools/dev/lattice_probe.py 10000 slots > $OUT/probe_10k_slots.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $OUT/probe_10k_slots.txt | cut -c1-400
