"""dev tool: owner-wave cycles per phase of fit_coop_kernel (a -DTSF_COOP_TIMING build) next to the
lone-wave phases of fit_kernel (-DTSF_FIT_TIMING), same series, one per CU.
  TSF_LIB_PATH=tools/variants/libtsf_amd_ct.so python tools/coop_timing.py coop
  TSF_LIB_PATH=tools/variants/libtsf_amd_ft.so python tools/coop_timing.py wave"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from time_series_spark_amd import _lib, forecaster as fc, synth  # noqa: E402

YEARLY = {'name': 'yearly', 'period': 365.25, 'fourier_order': 10}
WEEKLY = {'name': 'weekly', 'period': 7, 'fourier_order': 3}
mode = sys.argv[1] if len(sys.argv) > 1 else 'coop'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rk = {'coop': _lib.RK_COOP, 'wave': _lib.RK_WAVE, 'auto': _lib.RK_AUTO}[mode]
ds, y = synth.make_panel(N, 730, 'logistic', seed=751)
spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=[YEARLY, WEEKLY], max_iter=150,
                    residual_kernel=rk)
fc.fit_aligned(spec, ds, y, floor=np.zeros(N), cap=y.max(axis=1) * 1.1)
t0 = time.perf_counter()
r = fc.fit_aligned(spec, ds, y, floor=np.zeros(N), cap=y.max(axis=1) * 1.1)
dt = time.perf_counter() - t0
print(mode, 'N', N, 'evals total', int(r.n_eval.sum()), 'max', int(r.n_eval.max()), 'wall ms', round(1e3 * dt, 2),
      'us per eval of the longest', round(1e6 * dt / r.n_eval.max(), 2), flush=True)
