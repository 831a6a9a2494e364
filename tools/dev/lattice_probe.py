"""dev (round 6): the reference's model on a panel shaped like the reference's fixture -- every series observed at its OWN
subset of the slots of one time lattice (the fixture: Thu-Sun at 11:15 and 21:45, a 1.5 h lattice) -- through the three
routes such a panel can take:
  tables   a base-pair table per series (50 bytes per row and evaluation; round 5's route)
  lattice  rows of 22 bytes (t, y, segment word, lattice point), base pairs of the POINT from one shared table (round 6)
  dense    the shared table of whole design rows, gathered (round 3's route; option harm = 0)
Prints kernel time per route and whether a bit differs.  python tools/dev/lattice_probe.py [N] [kind: daily|slots]"""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from time_series_spark_amd import _lib, forecaster as fc, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
kind = sys.argv[2] if len(sys.argv) > 2 else 'daily'
T = 730
rng = np.random.default_rng(11)
lens = rng.integers(600, T + 1, N)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
ds, y = synth.make_panel(N, T, 'logistic', seed=751)
if kind == 'slots':
    # two slots a day (11:15 and 21:45), 365 days: the fixture's lattice (1.5 h)
    day = synth.DAY_NS
    ds = np.sort(np.concatenate([ds[:365] + 11 * 3600 * 10**9 + 15 * 60 * 10**9, ds[:365] + 21 * 3600 * 10**9 + 45 * 60 * 10**9]))
keep = [np.sort(rng.choice(T, size=c, replace=False)) for c in lens]
dsr = np.concatenate([ds[k] for k in keep])
yr = np.concatenate([y[i][k] for i, k in enumerate(keep)])
cap = np.array([y[i][k].max() * 1.1 for i, k in enumerate(keep)])
spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative',
                    seasonalities=[{'name': 'yearly', 'period': 365.25, 'fourier_order': 10}, {'name': 'weekly', 'period': 7, 'fourier_order': 3}])
ctx = fc.get_context()
L = _lib.load()
ms = ctypes.c_float(0.0)
base = None
for name, opts in (('tables', dict(lattice=0)), ('lattice', dict(lattice=1)), ('dense', dict(lattice=1, harm=0))):
    for k in ('lattice', 'harm'):
        ctx.set_option(k, opts.get(k, -1))
    out = []
    for rep in range(3):
        ctx.check(L.tsf_set_profiling(ctx.handle, 1))
        r = fc.fit_ragged(spec, off, dsr, yr, floor=np.zeros(N), cap=cap)
        ctx.check(L.tsf_last_fit_kernel_ms(ctx.handle, ctypes.byref(ms)))
        out.append(float(ms.value))
    if base is None:
        base = r
    same = all(np.array_equal(getattr(r, f), getattr(base, f)) for f in ('theta', 'status', 'n_iter', 'n_eval', 'fval'))
    print(json.dumps({'route': name, 'panel': '%d series of 600..730 of %d %s lattice slots' % (N, T, kind), 'fit_kernel_ms': out,
                      'series_per_s_kernel': N / (min(out) * 1e-3), 'mean_evals': float(r.n_eval.mean()), 'max_evals': int(r.n_eval.max()),
                      'bits': 'same' if same else 'DIFFER'}), flush=True)
for k in ('lattice', 'harm'):
    ctx.set_option(k, -1)
