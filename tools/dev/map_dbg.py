import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from time_series_spark_amd import _lib, forecaster as fc, synth
W3 = {'name': 'weekly', 'period': 7, 'fourier_order': 3}
M5 = {'name': 'monthly', 'period': 30.5, 'fourier_order': 5}
Y10 = {'name': 'yearly', 'period': 365.25, 'fourier_order': 10}
for seas in ([W3, M5], [W3], [Y10, W3]):
  for cps in (0.05, 0.5, 0.005):
    for n_cp in (25, 5, 0):
        T, N = 275, 6
        ds, y = synth.make_panel(N, T, 'linear', seed=5)
        spec = fc.ModelSpec(growth='linear', seasonalities=seas, n_changepoints=n_cp, converge=_lib.CONVERGE_MAP, changepoint_prior_scale=cps)
        lens = np.array([T, T - 10, T, T - 33, T, T - 50])
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        args = (off, np.concatenate([ds[:c] for c in lens]), np.concatenate([y[i][:c] for i, c in enumerate(lens)]))
        res = {}
        for tag, o, f in (('al_direct', {}, lambda: fc.fit_aligned(spec, ds, y)), ('al_cont', dict(map_direct=0), lambda: fc.fit_aligned(spec, ds, y)),
                          ('rg_direct', {}, lambda: fc.fit_ragged(spec, *args)), ('rg_cont', dict(map_direct=0), lambda: fc.fit_ragged(spec, *args))):
            with fc.get_context().options(**o):
                res[tag] = f()
        print([s['name'] for s in seas], 'cps', cps, 'n_cp', n_cp,
              'aligned d-c max %.2e' % np.max((res['al_direct'].fval - res['al_cont'].fval) / np.abs(res['al_cont'].fval)),
              'ragged d-c max %.2e' % np.max((res['rg_direct'].fval - res['rg_cont'].fval) / np.abs(res['rg_cont'].fval)),
              'status', res['al_direct'].status[:3], res['rg_direct'].status[:3], res['al_cont'].status[:3], flush=True)
