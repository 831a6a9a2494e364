#!/bin/bash
OUT=$PWD/gpurun_out/r06_n; mkdir -p $OUT; export TMPDIR=/tmp
( time python __graft_entry__.py smoke ) > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; grep -v "^compiled\|^Model\|amdgpu.ids" $OUT/smoke.txt | tail -16
