export TMPDIR=/tmp
mkdir -p gpurun_out/exp1
( for leg in "TSF_QUAD_REG=1" "TSF_QUAD_W4=0" "-"; do
  echo "-- leg $leg"
  ( if [ "$leg" != "-" ]; then export $leg; fi
    python tools/bench_configs.py cfg2x16 cfg2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], 'fit-kernel ms %.3f series/s %.0f evals/s %.0f' % (d['fit_kernel_ms'], d['series_per_s'], d['evals_per_s']))" )
done
echo "== phase cycles: LDS kernel"
TSF_QUAD_W4=0 TSF_LIB_PATH=$PWD/tools/variants/libtsf_amd_qtime.so python tools/bench_configs.py cfg2 cfg2x4 2>&1 >/dev/null | grep quad-timing | tail -2
echo "== phase cycles: lone-ish (300 series)"
TSF_QUAD_REG=0 TSF_LIB_PATH=$PWD/tools/variants/libtsf_amd_qtime.so python tools/quad_lone_timing.py 2>&1 | grep "quad-timing\|^N " | tail -9; echo "== register-M kernel"; TSF_QUAD_REG=1 TSF_LIB_PATH=$PWD/tools/variants/libtsf_amd_qtime4.so python tools/quad_lone_timing.py 2>&1 | grep "quad-timing\|^N " | tail -9
) 2>&1 | tee gpurun_out/exp1/summary.txt
