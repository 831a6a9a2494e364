"""Dev tool: fit the BASELINE cfg5 panel (1 000 000 x 90, float32 y) and list the series whose
fit ended abnormally and the ones with the most evaluations (how the never-settling line search
behind TSF_ST_EVAL_LIMIT was found).  `python tools/cfg5_scan.py` on the GPU box."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from time_series_spark_amd import synth, forecaster as fc
ds,y = synth.make_panel(1000000,90,'linear',seed=751,dtype=np.float32)
spec = fc.ModelSpec(growth='linear', seasonalities=fc.ModelSpec.auto_seasonalities(ds))
r = fc.fit_aligned(spec, ds, y)
bad = np.flatnonzero(r.status < 0)
print('bad', [(int(n), int(r.status[n]), int(r.n_iter[n]), int(r.n_eval[n])) for n in bad])
big = np.argsort(r.n_eval)[-5:]
print('largest n_eval', [(int(n), int(r.status[n]), int(r.n_iter[n]), int(r.n_eval[n])) for n in big])
