"""Times the fit path on a panel shaped like the reference's own fixture (tests/fixtures/model-input: every
(series_id, dim_id) observed at its OWN irregular timestamps): N series of 600..730 rows, each with its own random
gaps -- no two series share a timestamp vector, the timestamps lie on no lattice, so every series carries its own
design table (172 KB for 26 columns) and the residual-form kernel streams it from HBM at every evaluation.
Reference settings (logistic growth, multiplicative seasonality) and cfg2's model (quadratic form).
Round 6, third line: the fixture's shape taken literally -- every series at its OWN subset of the slots of one time lattice
(the fixture: Thu-Sun at 11:15 and 21:45; here 600..730 of 730 days) -- where a row is 22 bytes (t, y, segment word, lattice
point) and the base pairs are the lattice points', from one table all series share.
    python tools/bench_irregular.py [N]"""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from time_series_spark_amd import _lib, forecaster as fc, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
T = 730
rng = np.random.default_rng(11)
lens = rng.integers(600, T + 1, N)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
YEARLY = {'name': 'yearly', 'period': 365.25, 'fourier_order': 10}
WEEKLY = {'name': 'weekly', 'period': 7, 'fourier_order': 3}
for growth, mode in (('logistic', 'multiplicative'), ('linear', 'additive')):
    ds, y = synth.make_panel(N, T, growth, seed=751)
    # own timestamps: the daily grid plus a per-row jitter of up to +-6 h (order kept), different for every series
    dsr = np.concatenate([ds[:c] + rng.integers(-6 * 3600, 6 * 3600, c) * 1_000_000_000 for c in lens])
    yr = np.concatenate([y[i][:c] for i, c in enumerate(lens)])
    cap = np.array([y[i][:c].max() * 1.1 for i, c in enumerate(lens)])
    spec = fc.ModelSpec(growth=growth, seasonality_mode=mode, seasonalities=[YEARLY, WEEKLY])
    ctx = fc.get_context()
    L = _lib.load()
    ms = ctypes.c_float(0.0)
    out = []
    for rep in range(3):
        ctx.check(L.tsf_set_profiling(ctx.handle, 1))
        r = fc.fit_ragged(spec, off, dsr, yr, floor=np.zeros(N), cap=cap)
        ctx.check(L.tsf_last_fit_kernel_ms(ctx.handle, ctypes.byref(ms)))
        out.append(float(ms.value))
    evals = float(r.n_eval.sum())
    rows = float(lens.sum())
    k = min(out) * 1e-3
    # what an evaluation of the residual form reads per row: round 4 the design row ([KP = 28] doubles) + t, y, the
    # segment word; round 5 (base-pair kernel) two doubles per seasonality + t, y, the segment word
    harm = ctx.get_option('harm') != 0
    row_bytes = (2 * 2 * 8 if harm else 28 * 8) + 8 + 8 + 2
    x_bytes = float(np.sum(np.ceil(lens / 64) * 64 * row_bytes * r.n_eval)) if growth == 'logistic' else None
    # SURVEY 8d's algorithmic bytes: ds + y in, theta out, once per series (no forecast in this tool)
    alg = float(np.sum(lens * 16 + 54 * 8))
    print(json.dumps({'panel': '%d series of 600..730 rows at their own irregular timestamps' % N, 'growth': growth, 'mode': mode,
                      'fit_kernel_ms': out, 'series_per_s_kernel': N / k, 'mean_evals': evals / N, 'max_evals': int(r.n_eval.max()),
                      'status_counts': {str(int(a)): int(b) for a, b in zip(*np.unique(r.status, return_counts=True))},
                      'base_pair_kernel': bool(harm and growth == 'logistic'),
                      'row_bytes_read_by_the_evaluations': x_bytes,
                      'row_bytes_GBps': None if x_bytes is None else x_bytes / k / 1e9,
                      'algorithmic_bytes': alg, 'algorithmic_GBps': alg / k / 1e9,
                      'status_ok': int((r.status > 0).sum())}), flush=True)

# the fixture's shape taken literally: own subsets of one lattice's slots (rows dropped at random from the daily grid)
rng2 = np.random.default_rng(12)
lens2 = rng2.integers(600, T + 1, N)
off2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.int64)
ds, y = synth.make_panel(N, T, 'logistic', seed=751)
keep = [np.sort(rng2.choice(T, size=c, replace=False)) for c in lens2]
dsr = np.concatenate([ds[k] for k in keep])
yr = np.concatenate([y[i][k] for i, k in enumerate(keep)])
cap = np.array([y[i][k].max() * 1.1 for i, k in enumerate(keep)])
spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=[YEARLY, WEEKLY])
out = []
for rep in range(3):
    ctx.check(L.tsf_set_profiling(ctx.handle, 1))
    r = fc.fit_ragged(spec, off2, dsr, yr, floor=np.zeros(N), cap=cap)
    ctx.check(L.tsf_last_fit_kernel_ms(ctx.handle, ctypes.byref(ms)))
    out.append(float(ms.value))
k = min(out) * 1e-3
alg = float(np.sum(lens2 * 16 + 54 * 8))
x_bytes = float(np.sum(np.ceil(lens2 / 64) * 64 * (8 + 8 + 2 + 4) * r.n_eval))
print(json.dumps({'panel': '%d series at their own 600..730 of the 730 slots of a daily lattice' % N, 'growth': 'logistic', 'mode': 'multiplicative',
                  'fit_kernel_ms': out, 'series_per_s_kernel': N / k, 'mean_evals': float(r.n_eval.mean()), 'max_evals': int(r.n_eval.max()),
                  'lattice_point_kernel': bool(ctx.get_option('harm') != 0 and ctx.get_option('lattice') != 0),
                  'row_bytes_read_by_the_evaluations': x_bytes, 'row_bytes_GBps': x_bytes / k / 1e9,
                  'algorithmic_bytes': alg, 'algorithmic_GBps': alg / k / 1e9, 'status_ok': int((r.status > 0).sum())}), flush=True)
