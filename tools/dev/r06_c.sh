#!/bin/bash
# round 6: when every stage of the two pipelines has every chunk (TSF_PIPELINE_TIMING), host-side timers of the fit entry
TAG=${1:-r06_c}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for c in ${CHUNKS:-auto 1}; do
TSF_JOB_TIMING=1 TSF_PIPELINE_TIMING=1 TSF_HOST_TIMING=1 TSF_CSV_TIMING=1 python tools/e2e_bench.py --passes 3 --chunks $c ${KIND:-} > $OUT/e2e_$c.txt 2>&1
echo "== chunks $c"; grep "pipeline\|timing\|tsf_pack" $OUT/e2e_$c.txt | grep -v releasing | tail -n ${LINES_:-40}
tail -n 1 $OUT/e2e_$c.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k!='stages_s'})"
done
