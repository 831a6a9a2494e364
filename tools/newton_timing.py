"""One timed Newton fit of N x 90 series (fbprophet's optimiser for T < 100) and the L-BFGS fit of
the same panel beside it.  python tools/newton_timing.py [N]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from time_series_spark_amd import _lib, forecaster as fc, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ds, y = synth.make_panel(N, 90, 'linear', seed=751)
seas = fc.ModelSpec.auto_seasonalities(ds)
out = {'series': N, 'points': 90}
for name, algo in (('lbfgs', _lib.ALGO_LBFGS), ('newton', _lib.ALGO_NEWTON)):
    spec = fc.ModelSpec(growth='linear', seasonalities=seas, algorithm=algo)
    if name == 'lbfgs':
        fc.fit_aligned(spec, ds, y[:64])                 # context + first-launch cost outside the timing
    t0 = time.perf_counter()
    r = fc.fit_aligned(spec, ds, y)
    dt = time.perf_counter() - t0
    out[name] = {'seconds_host_call': dt, 'series_per_s': N / dt, 'mean_iters': float(r.n_iter.mean()),
                 'mean_evals': float(r.n_eval.mean()), 'max_evals': int(r.n_eval.max()),
                 'status_counts': {str(int(k)): int(v) for k, v in zip(*np.unique(r.status, return_counts=True))}}
    print(json.dumps({name: out[name]}), flush=True)
print(json.dumps(out))
