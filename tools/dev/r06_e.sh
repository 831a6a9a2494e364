#!/bin/bash
OUT=$PWD/gpurun_out/r06_e; mkdir -p $OUT; export TMPDIR=/tmp
for kind in cfg2 reference; do for c in 1 auto; do python tools/e2e_bench.py --n 40000 --passes 3 --chunks $c --kind $kind > $OUT/e2e_${kind}_40k_$c.txt 2>&1; tail -n 1 $OUT/e2e_${kind}_40k_$c.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k!='stages_s'})"; done; done
