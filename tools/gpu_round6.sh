#!/bin/bash
# Round 6, the closing GPU call (ONE per round: DESIGN.md section 8 quotes THIS run): the whole GPU suite, rocprofv3 kernel
# stats and the PMC passes of the bench command (-> profiles/pmc_latest.json via tools/pmc_summary.py), the PMC pass of the
# irregular panel (-> profiles/irregular_pmc_latest.json), the bench line with those profiles in place, the table of
# configurations, ragged panels, files -> files (both models, 10 000 and 40 000 series), the MAP-mode probe.
# usage: gpurun --timeout 2700 -- 'bash tools/gpu_round6.sh r06_final'
TAG=${1:-r06_final}
STEPS=${STEPS:-3}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt
tail -4 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
fi
echo "== rocprofv3 kernel stats" | tee -a $OUT/summary.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats --output-format csv -- python $OLDPWD/bench.py --steps $STEPS --warmup 1 --timed-only > $OUT/prof_stats.log 2>&1 ); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof_stats -name '*kernel_stats.csv' | head -1 | xargs -r head -12 | tee -a $OUT/summary.txt
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_fetch -o fetch --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --timed-only > $OUT/prof_fetch.log 2>&1 ); echo "pmc fetch rc=$?" | tee -a $OUT/summary.txt
( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_write -o write --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --timed-only > $OUT/prof_write.log 2>&1 ); echo "pmc write rc=$?" | tee -a $OUT/summary.txt
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace -d $OUT/prof_sq -o sq --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --timed-only > $OUT/prof_sq.log 2>&1 ); echo "pmc sq rc=$?" | tee -a $OUT/summary.txt
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1; tail -3 $OUT/pmc_summary.txt | tee -a $OUT/summary.txt
cp $OUT/pmc_latest.json profiles/pmc_latest.json 2>/dev/null
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_irregular -o f --output-format csv -- python $OLDPWD/tools/bench_irregular.py > $OUT/irregular.jsonl 2> $OUT/irregular.err ); echo "pmc irregular rc=$?" | tee -a $OUT/summary.txt
python tools/irregular_pmc_summary.py $OUT/prof_irregular $OUT/irregular_pmc_latest.json | tee -a $OUT/summary.txt
cp $OUT/irregular_pmc_latest.json profiles/irregular_pmc_latest.json 2>/dev/null
echo "== bench (with the profiles of THIS call in place)" | tee -a $OUT/summary.txt
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench_time.txt; echo "bench rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/bench_time.txt | tee -a $OUT/summary.txt
cut -c1-1500 $OUT/bench.json | tee -a $OUT/summary.txt
echo "== configurations" | tee -a $OUT/summary.txt
timeout 900 python tools/bench_configs.py ${CFGS:-cfg1 cfg3 cfg3_full cfg4 cfg5 cfg2_resid ref10k ref100k cfg2x4 cfg2x16 lin_hol} > $OUT/configs.jsonl 2> $OUT/configs.err
python - <<PY | tee -a $OUT/summary.txt
import json
for l in open('$OUT/configs.jsonl'):
    try: d = json.loads(l)
    except Exception: continue
    print('%-12s fit-kernel %9.3f ms  %9.0f series/s  evals mean %.0f max %.0f  %.1f M evals/s' % (d.get('config'), d.get('fit_kernel_ms', -1), d.get('series_per_s', -1), d.get('mean_evals', -1), d.get('max_evals', -1), d.get('evals_per_s', 0) / 1e6))
PY
timeout 300 python tools/bench_ragged.py > $OUT/ragged.txt 2>&1; tail -6 $OUT/ragged.txt | cut -c1-300 | tee -a $OUT/summary.txt
if [ -z "$SKIP_E2E" ]; then
echo "== files -> files" | tee -a $OUT/summary.txt
for kind in cfg2 reference; do
timeout 600 python tools/e2e_bench.py --kind $kind --passes 5 --stages > $OUT/e2e_$kind.txt 2>&1; tail -1 $OUT/e2e_$kind.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k!='stages_s'})" | tee -a $OUT/summary.txt
timeout 600 python tools/e2e_bench.py --kind $kind --n 40000 --passes 3 > $OUT/e2e_${kind}_40k.txt 2>&1; tail -1 $OUT/e2e_${kind}_40k.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k!='stages_s'})" | tee -a $OUT/summary.txt
done
fi
echo "== converge = MAP against the independent solver" | tee -a $OUT/summary.txt
timeout 900 python tools/dev/map_probe.py cfg2:256 ref:64 cfg5:64 cfg4:16 > $OUT/map_probe.txt 2>&1; cat $OUT/map_probe.txt | tee -a $OUT/summary.txt
timeout 600 python tools/dev/map_direct_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/map_direct_timing.txt; cut -c1-700 $OUT/map_direct_timing.txt | tee -a $OUT/summary.txt
echo "== lattice panels: a table per series / lattice points / gathered rows" | tee -a $OUT/summary.txt
timeout 600 python tools/dev/lattice_probe.py 10000 daily 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee $OUT/lattice_probe.txt | tee -a $OUT/summary.txt
echo "== Stan's Newton, 1 000 000 x 90" | tee -a $OUT/summary.txt
timeout 600 python tools/newton_timing.py 1000000 2>&1 | tail -1 | tee $OUT/newton_1m.txt | cut -c1-600 | tee -a $OUT/summary.txt
find $OUT -name '*.db' -delete 2>/dev/null
find $OUT -size +4M -delete 2>/dev/null
echo done
