"""dev tool: per-phase cycles of fit_kernel (a -DTSF_FIT_TIMING build of tsf_inst_g1m1.hip, see
tools/build_variant.sh) for waves that run ALONE (one series per CU) and for a saturated launch.
  TSF_LIB_PATH=tools/variants/libtsf_amd_ft.so python tools/lone_wave_timing.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from time_series_spark_amd import forecaster as fc, synth  # noqa: E402

YEARLY = {'name': 'yearly', 'period': 365.25, 'fourier_order': 10}
WEEKLY = {'name': 'weekly', 'period': 7, 'fourier_order': 3}
for N in (128, 20000):
    ds, y = synth.make_panel(N, 730, 'logistic', seed=751)
    spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=[YEARLY, WEEKLY], max_iter=150)
    fc.fit_aligned(spec, ds, y, floor=np.zeros(N), cap=y.max(axis=1) * 1.1)
    r = fc.fit_aligned(spec, ds, y, floor=np.zeros(N), cap=y.max(axis=1) * 1.1)
    print('N', N, 'mean evals', r.n_eval.mean(), flush=True)
