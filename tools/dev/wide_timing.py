"""dev tool: per-phase cycles of fit_kernel on the cfg4 model (logistic, multiplicative, 26 seasonal + 30 holiday
columns, P = 84: two parameters per lane) -- a -DTSF_FIT_TIMING build of tsf_inst_g1m1.hip
(tools/build_variant.sh ft tsf_inst_g1m1.hip -DTSF_FIT_TIMING), waves alone and saturated.
  TSF_LIB_PATH=tools/variants/libtsf_amd_ft.so python tools/dev/wide_timing.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from time_series_spark_amd import _lib, forecaster as fc, synth  # noqa: E402

YEARLY = {'name': 'yearly', 'period': 365.25, 'fourier_order': 10}
WEEKLY = {'name': 'weekly', 'period': 7, 'fourier_order': 3}
T = 730
ds = synth.daily_grid(T)
extra, names = synth.holiday_matrix(ds, 10)
for N in (128, 20000):
    _, y = synth.make_panel(N, T, 'logistic', seed=751, holidays=extra)
    spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=[YEARLY, WEEKLY],
                        extra=[{'name': n} for n in names], max_iter=150)
    fc.fit_aligned(spec, ds, y, floor=np.zeros(N), cap=y.max(axis=1) * 1.1, extra=extra)
    r = fc.fit_aligned(spec, ds, y, floor=np.zeros(N), cap=y.max(axis=1) * 1.1, extra=extra)
    print('N', N, 'mean evals', r.n_eval.mean(), flush=True)
