#!/bin/bash
# round 6, first call: where the host time of the files -> files run goes on the GPU box (baseline before the pipeline work)
OUT=$PWD/gpurun_out/r06_a
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt; free -g >> $OUT/nproc.txt
python tools/e2e_bench.py > $OUT/e2e_cfg2.txt 2>&1
python tools/e2e_bench.py --kind reference > $OUT/e2e_ref.txt 2>&1
TSF_HOST_TIMING=1 python tools/dev/model_stage_profile.py > $OUT/model_stage.txt 2>&1
python tools/dev/e2e_profile.py > $OUT/e2e_profile.txt 2>&1
tail -3 $OUT/e2e_cfg2.txt $OUT/e2e_ref.txt
head -40 $OUT/model_stage.txt
