#!/bin/bash
# Round 5, residual-form kernel with the Fourier columns expanded from the rows' base pairs (eval_fg HARM): the GPU suite,
# then the reference-model configurations with the route on / off and on the variant libraries given, the irregular
# panel, and the per-phase cycle counters of the timing build.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_r05_harm.sh r05_a [variant ...]'
TAG=${1:-r05_a}; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -x tools/probes/bin/scan_probe ]; then tools/probes/bin/scan_probe 2>&1 | tail -4 | tee $OUT/scan_probe.txt; fi
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests -m gpu -q --tb=short ${KEXPR:+-k "$KEXPR"} > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -15 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
fi
CFGS=${CFGS:-ref10k ref100k cfg4 cfg1 cfg2_resid}
show() { python - "$1" "$2" <<'PY' | tee -a $OUT/summary.txt
import json, sys
for l in open(sys.argv[1]):
    try:
        d = json.loads(l)
    except Exception:
        continue
    print('%-10s %-12s fit-kernel %9.3f ms  %9.0f series/s  evals mean %.0f max %.0f  %.1f M evals/s  status %s' % (sys.argv[2], d.get('config'), d.get('fit_kernel_ms', -1), d.get('series_per_s', -1), d.get('mean_evals', -1), d.get('max_evals', -1), d.get('evals_per_s', 0) / 1e6, d.get('status_counts')))
PY
}
timeout 600 python tools/bench_configs.py $CFGS > $OUT/configs.jsonl 2> $OUT/configs.err; show $OUT/configs.jsonl harm
TSF_OPTIONS=harm=0 timeout 600 python tools/bench_configs.py ${CFGS_OFF:-ref10k ref100k cfg4} > $OUT/configs_off.jsonl 2> $OUT/configs_off.err; show $OUT/configs_off.jsonl table
for V in "$@"; do
  [ "$V" = ft ] && continue
  TSF_LIB_PATH=$PWD/tools/variants/libtsf_amd_$V.so timeout 600 python tools/bench_configs.py ${CFGS_VAR:-ref10k ref100k} > $OUT/configs_$V.jsonl 2> $OUT/configs_$V.err; show $OUT/configs_$V.jsonl $V
done
timeout 600 python tools/bench_irregular.py > $OUT/irregular.jsonl 2> $OUT/irregular.err; echo "irregular (harm):" | tee -a $OUT/summary.txt; cut -c1-700 $OUT/irregular.jsonl | tee -a $OUT/summary.txt
TSF_OPTIONS=harm=0 timeout 600 python tools/bench_irregular.py > $OUT/irregular_off.jsonl 2> $OUT/irregular_off.err; echo "irregular (table):" | tee -a $OUT/summary.txt; cut -c1-700 $OUT/irregular_off.jsonl | tee -a $OUT/summary.txt
if [ -f tools/variants/libtsf_amd_ft.so ]; then
  TSF_LIB_PATH=$PWD/tools/variants/libtsf_amd_ft.so timeout 600 python tools/bench_configs.py ref100k > /dev/null 2> $OUT/ft_harm.err; grep fit-timing $OUT/ft_harm.err | tail -2 | tee -a $OUT/summary.txt
  TSF_OPTIONS=harm=0 TSF_LIB_PATH=$PWD/tools/variants/libtsf_amd_ft.so timeout 600 python tools/bench_configs.py ref100k > /dev/null 2> $OUT/ft_table.err; grep fit-timing $OUT/ft_table.err | tail -2 | tee -a $OUT/summary.txt
fi
if [ -n "$PMC_IRREGULAR" ]; then
for leg in harm table; do
  if [ $leg = table ]; then export TSF_OPTIONS=harm=0; else unset TSF_OPTIONS; fi
  ( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_fetch_$leg -o f --output-format csv -- python $OLDPWD/tools/bench_irregular.py > $OUT/pmc_$leg.log 2>&1 ); echo "pmc $leg rc=$?" | tee -a $OUT/summary.txt
  python - <<PY | tee -a $OUT/summary.txt
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('$OUT/prof_fetch_$leg/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:90]
        if 'fit_kernel' in k or 'fit_quad' in k or 'fit_coop' in k:
            per[k][r['Dispatch_Id']] += float(r['Counter_Value'])
for k, d in per.items():
    v = sorted(d.values())
    print('$leg %-92s launches %d FETCH_SIZE KiB median %.0f  (= %.2f GB at face value)' % (k, len(v), v[len(v)//2], v[len(v)//2] * 1024 / 1e9))
PY
done
unset TSF_OPTIONS
fi
find $OUT -name '*.db' -delete 2>/dev/null
find $OUT -size +4M -delete 2>/dev/null
echo done
