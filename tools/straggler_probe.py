"""dev tool: the longest series of the reference-settings 10 000 x 730 panel fitted ALONE (one
workgroup / one wave on an otherwise idle GPU): wall time per evaluation, for the cooperative and the
one-wave kernel.  With a -DTSF_COOP_TIMING library the owner's cycle counts are printed too, which
gives the frequency of the cycle counter in this regime."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from time_series_spark_amd import _lib, forecaster as fc, synth  # noqa: E402

YEARLY = {'name': 'yearly', 'period': 365.25, 'fourier_order': 10}
WEEKLY = {'name': 'weekly', 'period': 7, 'fourier_order': 3}
N = 10000
ds, y = synth.make_panel(N, 730, 'logistic', seed=751)
cap = y.max(axis=1) * 1.1
kw = dict(growth='logistic', seasonality_mode='multiplicative', seasonalities=[YEARLY, WEEKLY])
r = fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_WAVE, **kw), ds, y, floor=np.zeros(N), cap=cap)
order = np.argsort(-r.n_eval)
print('longest', r.n_eval[order[:5]], flush=True)
for k in (0, 1):
    i = int(order[k])
    for mode, rk in (('coop', _lib.RK_COOP), ('wave', _lib.RK_WAVE)):
        spec = fc.ModelSpec(residual_kernel=rk, **kw)
        fc.fit_aligned(spec, ds, y[i:i + 1], floor=np.zeros(1), cap=cap[i:i + 1])
        t0 = time.perf_counter()
        r1 = fc.fit_aligned(spec, ds, y[i:i + 1], floor=np.zeros(1), cap=cap[i:i + 1])
        dt = time.perf_counter() - t0
        print(mode, 'series', i, 'evals', int(r1.n_eval[0]), 'wall ms', round(1e3 * dt, 2), 'us/eval', round(1e6 * dt / r1.n_eval[0], 2), flush=True)
# does the per-evaluation time of a lone series drift when the GPU has been nearly idle for a while?
i = int(order[0])
for mode, rk in (('wave', _lib.RK_WAVE), ('coop', _lib.RK_COOP)):
    spec = fc.ModelSpec(residual_kernel=rk, **kw)
    out = []
    for rep in range(12):
        t0 = time.perf_counter()
        r1 = fc.fit_aligned(spec, ds, y[i:i + 1], floor=np.zeros(1), cap=cap[i:i + 1])
        out.append(round(1e6 * (time.perf_counter() - t0) / r1.n_eval[0], 2))
    print(mode, 'back-to-back lone fits, us/eval:', out, flush=True)
    # the same after a busy phase
    fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_WAVE, max_iter=150, **kw), ds, y, floor=np.zeros(N), cap=cap)
    t0 = time.perf_counter()
    r1 = fc.fit_aligned(spec, ds, y[i:i + 1], floor=np.zeros(1), cap=cap[i:i + 1])
    print(mode, 'right after a busy launch:', round(1e6 * (time.perf_counter() - t0) / r1.n_eval[0], 2), flush=True)
