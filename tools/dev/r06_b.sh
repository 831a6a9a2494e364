#!/bin/bash
# round 6: files -> files with the chunked pipelines (modeler: read | fit | parquet; scorer: read | predict | sink)
TAG=${1:-r06_b}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python tools/e2e_bench.py --stages --passes 5 > $OUT/e2e_cfg2.txt 2>&1
python tools/e2e_bench.py --passes 5 --chunks 1 > $OUT/e2e_cfg2_chunks1.txt 2>&1
python tools/e2e_bench.py --passes 5 --chunks 8 > $OUT/e2e_cfg2_chunks8.txt 2>&1
python tools/e2e_bench.py --kind reference --passes 5 --stages > $OUT/e2e_ref.txt 2>&1
python tools/e2e_bench.py --kind reference --passes 5 --chunks 1 > $OUT/e2e_ref_chunks1.txt 2>&1
for f in e2e_cfg2 e2e_cfg2_chunks1 e2e_cfg2_chunks8 e2e_ref e2e_ref_chunks1; do echo "== $f"; grep -v "^{" $OUT/$f.txt | grep -v warm | tail -n 20; tail -n 1 $OUT/$f.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k!='stages_s'})"; done
