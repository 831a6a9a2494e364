#!/bin/bash
# dev (round 6): t and y of a series' rows staged in LDS by the prefetch form of the base-pair kernel (option stage_ty): kernel time
# and FETCH_SIZE of tools/bench_irregular.py's panels (no lattice; lattice), off and on
OUT=$PWD/gpurun_out/${1:-r06_stage}; mkdir -p $OUT; export TMPDIR=/tmp
HERE=$PWD
timeout 900 python -m pytest tests -m gpu -x -q -k "prefetch or lattice or cooperative_tail or ragged or fixture or full_size_reference" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
for v in 0 1; do
( cd /tmp && TSF_OPTIONS=stage_ty=$v timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_$v -o f --output-format csv -- python $HERE/tools/bench_irregular.py > $OUT/irregular_$v.jsonl 2> $OUT/irregular_$v.err ); echo "stage_ty=$v rc=$?"
python tools/irregular_pmc_summary.py $OUT/prof_$v $OUT/pmc_$v.json | cut -c1-700
python - <<PY
import json
for l in open('$OUT/irregular_$v.jsonl'):
    d = json.loads(l)
    print('stage_ty=$v', d['panel'][:60], d['growth'], 'fit ms', [round(x, 2) for x in d['fit_kernel_ms']])
PY
done
find $OUT -name '*.db' -delete 2>/dev/null; find $OUT -size +4M -delete 2>/dev/null
