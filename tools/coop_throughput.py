"""dev tool: saturated throughput (evaluations per second) of the one-wave kernel against the cooperative
kernel on a panel that fits the checkpoint slots (every series on the chosen kernel).
  python tools/coop_throughput.py [ref|cfg4] [N]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from time_series_spark_amd import _lib, forecaster as fc, synth  # noqa: E402

YEARLY = {'name': 'yearly', 'period': 365.25, 'fourier_order': 10}
WEEKLY = {'name': 'weekly', 'period': 7, 'fourier_order': 3}
which = sys.argv[1] if len(sys.argv) > 1 else 'ref'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = 730
extra = None
ex_spec = []
if which == 'cfg4':
    ds = synth.daily_grid(T)
    allm, names = synth.holiday_matrix(ds, 10)
    extra = np.ascontiguousarray(allm)
    ex_spec = [{'name': n} for n in names]
ds, y = synth.make_panel(N, T, 'logistic', seed=751, holidays=extra)
for mode, rk in (('wave', _lib.RK_WAVE), ('coop', _lib.RK_COOP)):
    spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=[YEARLY, WEEKLY], extra=ex_spec,
                        max_iter=100, residual_kernel=rk)
    fc.fit_aligned(spec, ds, y, floor=np.zeros(N), cap=y.max(axis=1) * 1.1, extra=extra)
    t0 = time.perf_counter()
    r = fc.fit_aligned(spec, ds, y, floor=np.zeros(N), cap=y.max(axis=1) * 1.1, extra=extra)
    dt = time.perf_counter() - t0
    print(which, mode, 'N', N, 'K', spec.K, 'evals', int(r.n_eval.sum()), 'wall ms', round(1e3 * dt, 1), 'M evals/s', round(r.n_eval.sum() / dt / 1e6, 2), flush=True)
