#!/bin/bash
OUT=$PWD/gpurun_out/r06_d; mkdir -p $OUT; export TMPDIR=/tmp
python tools/dev/read_probe2.py 2>&1 | grep -v timing | tee $OUT/read_probe2.txt
for c in 1 2 auto; do python tools/e2e_bench.py --passes 5 --chunks $c > $OUT/e2e_cfg2_$c.txt 2>&1; tail -n 1 $OUT/e2e_cfg2_$c.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k!='stages_s'})"; done
python tools/e2e_bench.py --passes 5 --chunks 1 --kind reference > $OUT/e2e_ref_1.txt 2>&1; tail -n 1 $OUT/e2e_ref_1.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k!='stages_s'})"
