"""dev tool: what the job layer's grouping of series by timestamp vector costs on a moderately ragged panel -- 10 000
series on 91 distinct grids (the panel of tools/bench_ragged.py) through model_arrays with groups of >= 2 series as
aligned launches (the policy until round 3) and with the default policy (groups of < 4 096 series join the ragged call).

  python tools/group_policy_probe.py
"""
import io
import contextlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from time_series_spark_amd import synth  # noqa: E402
from time_series_spark_amd.jobs import prophet_modeler as pm  # noqa: E402

N, T = 10000, 730
for growth, mode in (('linear', 'additive'), ('logistic', 'multiplicative')):
    ds, y = synth.make_panel(N, T, growth, seed=751)
    lens = 640 + (np.arange(N) * 37) % 91
    sid = np.repeat(np.arange(N, dtype=np.int64), lens)
    did = np.ones(len(sid), dtype=np.int64)
    dsr = np.concatenate([ds[:c] for c in lens])
    yr = np.concatenate([y[i][:c] for i, c in enumerate(lens)]).astype(np.float64)
    out = {}
    blobs = {}
    for tag, mg in (('groups_of_2_or_more_aligned', 2), ('default_policy', None)):
        pr = {'growth': growth, 'seasonality_mode': mode, 'algorithm': 'lbfgs'}
        if mg:
            pr['min_aligned_group'] = mg
        cfg = {'model': {'floor': 0, 'cap_multiplier': 1.1, 'prophet': pr}}
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                m = pm.model_arrays(cfg)(sid, did, dsr, yr)
            ts.append(round(time.perf_counter() - t0, 4))
        out[tag] = ts
        blobs[tag] = [bytes(b) for b in m['model']]
    print(json.dumps({'panel': '10000 series on 91 grids (640..730 rows)', 'growth': growth, 'mode': mode,
                      'model_arrays_seconds': out,
                      'same_models': blobs['groups_of_2_or_more_aligned'] == blobs['default_policy']}), flush=True)
