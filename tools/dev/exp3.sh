export TMPDIR=/tmp
mkdir -p gpurun_out/exp3
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "routes or quadratic or full_size_panel or full_size_other or randomised_model or ragged or design_and or odd_shapes or fit_predict" 2>&1 | tail -15
python tools/bench_configs.py cfg2 cfg3 cfg2x4 cfg2x16 cfg5 lin_hol 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], 'fit-kernel ms %.3f series/s %.0f evals/s %.0f mean evals %.1f max %d' % (d['fit_kernel_ms'], d['series_per_s'], d['evals_per_s'], d['mean_evals'], d['max_evals']))"
echo "== phase cycles"
TSF_QUAD_W4=0 TSF_LIB_PATH=$PWD/tools/variants/libtsf_amd_qtime.so python tools/bench_configs.py cfg2 cfg2x4 2>&1 >/dev/null | grep quad-timing | tail -2
) 2>&1 | tee gpurun_out/exp3/summary.txt
