"""dev: cProfile of the modeler's middle stage (model_arrays: pack + group + H2D + fit + D2H + blobs + frame) on 10 000 x 730."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from time_series_spark_amd import synth  # noqa: E402
from time_series_spark_amd.jobs import prophet_modeler as pm  # noqa: E402

N, T = 10000, 730
ds, y = synth.make_panel(N, T, 'linear', seed=2)
sid = np.repeat(np.arange(N, dtype=np.int64), T)
did = np.ones(N * T, dtype=np.int64)
dsr = np.tile(ds, N)
yr = np.ascontiguousarray(y.reshape(-1))
cfg = {'io': {'input': 'x', 'models': 'y'}, 'model': {'floor': 0, 'cap_multiplier': 1.1,
       'prophet': {'growth': 'linear', 'seasonality_mode': 'additive', 'yearly_seasonality': True}}}
out, sys.stdout = sys.stdout, open(os.devnull, 'w')
f = pm.model_arrays(cfg)
f(sid, did, dsr, yr)
t0 = time.time()
pr = cProfile.Profile()
pr.enable()
m = f(sid, did, dsr, yr)
pr.disable()
dt = time.time() - t0
sys.stdout = out
print('model_arrays %.1f ms for %d models' % (dt * 1e3, len(m)))
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
