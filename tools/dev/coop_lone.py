"""dev: one series of the reference-settings 10 000 x 730 panel (default: 8779, the one that reaches Stan's iteration
limit: 29 394 evaluations) fitted ALONE by the cooperative kernel, microseconds per evaluation; with a -DTSF_COOP_TIMING
library (TSF_LIB_PATH) the owner's cycles per phase are printed by the library.
  python tools/dev/coop_lone.py [series ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from time_series_spark_amd import _lib, forecaster as fc, synth  # noqa: E402

YEARLY = {'name': 'yearly', 'period': 365.25, 'fourier_order': 10}
WEEKLY = {'name': 'weekly', 'period': 7, 'fourier_order': 3}
N = 10000
ds, y = synth.make_panel(N, 730, 'logistic', seed=751)
cap = y.max(axis=1) * 1.1
kw = dict(growth='logistic', seasonality_mode='multiplicative', seasonalities=[YEARLY, WEEKLY])
spec = fc.ModelSpec(residual_kernel=_lib.RK_COOP, **kw)
for i in [int(a) for a in sys.argv[1:]] or [8779]:
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        r1 = fc.fit_aligned(spec, ds, y[i:i + 1], floor=np.zeros(1), cap=cap[i:i + 1])
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    print('coop series', i, 'evals', int(r1.n_eval[0]), 'iters', int(r1.n_iter[0]), 'status', int(r1.status[0]),
          'wall ms %.2f' % (1e3 * best), 'us/eval %.3f' % (1e6 * best / r1.n_eval[0]),
          'theta checksum %.17g' % float(np.sum(r1.theta[0] * np.arange(1, r1.theta.shape[1] + 1))), flush=True)
