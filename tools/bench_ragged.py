"""Times the fit kernel on a RAGGED panel (every series carries its own timestamps and design
tables): the cfg2 series with lengths 640..730, linear/additive (quadratic form, per-series
Z^T Z built in-kernel) and the reference's logistic/multiplicative settings (residual form).
The panel's 10 000 series share 91 distinct timestamp vectors (lengths 640..730 of one daily calendar), which since
round 4 share grid tables and one prebuilt Z^T Z per vector: every (model) is timed that way ("grids": "shared", the
default) and with the context option grid_share = 0 ("own": a grid per series, what a panel of unrelated calendars costs).
Host entry point (PCIe copies outside the kernel time, which comes from the library's HIP events)."""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from time_series_spark_amd import _lib, forecaster as fc, synth  # noqa: E402

N, T = 10000, 730
for growth, mode in (('linear', 'additive'), ('logistic', 'multiplicative')):
    if os.environ.get('ONLY') not in (None, '', growth):      # ONLY=linear | logistic: one model (A/B runs)
        continue
    ds, y = synth.make_panel(N, T, growth, seed=751)
    lens = 640 + (np.arange(N) * 37) % 91
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    dsr = np.concatenate([ds[:c] for c in lens])
    yr = np.concatenate([y[i][:c] for i, c in enumerate(lens)])
    # RK=wave: the one-wave residual kernel without the cooperative tail (for A/B runs)
    rk = {'wave': _lib.RK_WAVE, 'coop': _lib.RK_COOP}.get(os.environ.get('RK', ''), _lib.RK_AUTO)
    spec = fc.ModelSpec(growth=growth, seasonality_mode=mode, residual_kernel=rk,
                        seasonalities=[{'name': 'yearly', 'period': 365.25, 'fourier_order': 10},
                                       {'name': 'weekly', 'period': 7, 'fourier_order': 3}])
    cap = np.array([y[i][:c].max() * 1.1 for i, c in enumerate(lens)])
    ctx = fc.get_context()
    L = _lib.load()
    ms = ctypes.c_float(0.0)
    for grids in ('shared', 'own'):
        if grids == 'own':
            ctx.set_option('grid_share', 0)
        out, wall = [], []
        import time
        for rep in range(3):
            ctx.check(L.tsf_set_profiling(ctx.handle, 1))
            t0 = time.perf_counter()
            r = fc.fit_ragged(spec, off, dsr, yr, floor=np.zeros(N), cap=cap)
            wall.append((time.perf_counter() - t0) * 1e3)
            ctx.check(L.tsf_last_fit_kernel_ms(ctx.handle, ctypes.byref(ms)))
            out.append(float(ms.value))
        ctx.set_option('grid_share', -1)
        print(json.dumps({'panel': 'ragged 10000 series, 640..730 rows, 91 distinct timestamp vectors', 'grids': grids,
                          'growth': growth, 'mode': mode, 'residual_kernel': os.environ.get('RK', 'auto'),
                          'fit_kernel_ms': out, 'host_call_ms': wall, 'series_per_s_kernel': N / (min(out) * 1e-3),
                          'mean_evals': float(r.n_eval.mean()),
                          'status_ok': int((r.status > 0).sum())}), flush=True)
