"""dev (round 6): randomised differential runs of the routes added this round.
  lattice: ragged panels on random time lattices (own subsets of the slots), the three models with a compiled expansion and a
           few others: option lattice = default against lattice = 0 (tables per series) -- every output bit for bit;
  map:     converge = MAP computed directly against the continuation (option map_direct = 0) on random linear / additive
           panels, aligned and ragged -- finite estimates, and a tally of which route ended at the lower objective (short
           noisy histories have several local minima).
    python tools/dev/route_stress.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from time_series_spark_amd import _lib, forecaster as fc, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
Y10 = {'name': 'yearly', 'period': 365.25, 'fourier_order': 10}
W3 = {'name': 'weekly', 'period': 7, 'fourier_order': 3}
D4 = {'name': 'daily', 'period': 1, 'fourier_order': 4}
M5 = {'name': 'monthly', 'period': 30.5, 'fourier_order': 5}
t_end = time.time() + budget
n_lat = n_map = 0
tally = [0, 0, 0]
worst = 0.0
while time.time() < t_end:
    # ---- lattice route
    step_h = float(rng.choice([24, 12, 6, 1.5]))
    n_slots = int(rng.integers(60, 900))
    slots = synth.START_NS + (np.arange(n_slots) * step_h * 3600e9).astype(np.int64)
    N = int(rng.integers(1, 30))
    growth = str(rng.choice(['linear', 'logistic']))
    mode = str(rng.choice(['additive', 'multiplicative']))
    seas = [[Y10, W3], [W3, D4], [W3], [W3, M5], [Y10, W3, M5]][int(rng.integers(0, 5))]
    n_cp = int(rng.choice([0, 5, 25]))
    _, ym = synth.make_panel(N, n_slots, growth, seed=int(rng.integers(1, 1 << 30)))
    keep = [np.sort(rng.choice(n_slots, size=int(rng.integers(max(4, n_slots // 2), n_slots + 1)), replace=False)) for _ in range(N)]
    off = np.concatenate([[0], np.cumsum([len(k) for k in keep])]).astype(np.int64)
    dsr = np.concatenate([slots[k] for k in keep])
    yr = np.concatenate([ym[i][k] for i, k in enumerate(keep)])
    kw = dict(floor=np.zeros(N), cap=np.array([ym[i][k].max() * 1.1 for i, k in enumerate(keep)])) if growth == 'logistic' else {}
    spec = fc.ModelSpec(growth=growth, seasonality_mode=mode, seasonalities=seas, n_changepoints=n_cp, eval_form=_lib.EVAL_RESIDUAL,
                        max_iter=int(rng.choice([30, 200, 10000])))
    a = fc.fit_ragged(spec, off, dsr, yr, **kw)
    with fc.get_context().options(lattice=0):
        b = fc.fit_ragged(spec, off, dsr, yr, **kw)
    for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status', 'y_scale'):
        if not np.array_equal(getattr(a, name), getattr(b, name), equal_nan=True):
            print('LATTICE MISMATCH', name, dict(step_h=step_h, n_slots=n_slots, N=N, growth=growth, mode=mode, seas=[s['name'] for s in seas], n_cp=n_cp), flush=True)
            sys.exit(1)
    n_lat += 1
    # ---- direct MAP
    T = int(rng.integers(30, 800))
    N = int(rng.integers(1, 40))
    ds, y = synth.make_panel(N, T, 'linear', seed=int(rng.integers(1, 1 << 30)))
    seas = [[W3], [Y10, W3], [W3, M5]][int(rng.integers(0, 3))]
    spec = fc.ModelSpec(growth='linear', seasonalities=seas, n_changepoints=int(rng.choice([0, 5, 25])), converge=_lib.CONVERGE_MAP,
                        changepoint_prior_scale=float(rng.choice([0.05, 0.5, 0.005])))
    ragged = bool(rng.integers(0, 2))
    if ragged:
        lens = rng.integers(max(10, T // 2), T + 1, N)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        args = (off, np.concatenate([ds[:c] for c in lens]), np.concatenate([y[i][:c] for i, c in enumerate(lens)]))
        fit = lambda: fc.fit_ragged(spec, *args)
    else:
        fit = lambda: fc.fit_aligned(spec, ds, y)
    d = fit()
    with fc.get_context().options(map_direct=0):
        c = fit()
    ok = (c.status >= _lib.ST_MAP_KKT) & (d.status >= _lib.ST_MAP_KKT)
    # (fewer rows than parameters: sigma -> 0 and the posterior is unbounded below; both routes then chase minus infinity)
    ok &= (np.exp(d.theta[:, 2]) > 1e-4) & (np.exp(c.theta[:, 2]) > 1e-4)
    if ok.any():
        rel = (d.fval[ok] - c.fval[ok]) / np.maximum(1.0, np.abs(c.fval[ok]))
        if not np.isfinite(d.theta[ok]).all():
            print('MAP: non-finite estimate', dict(T=T, N=N, ragged=ragged), flush=True)
            sys.exit(1)
        # the posterior of a short noisy history can have several local minima (T/2 log sigma^2 is concave): either route
        # ends in one of them; count who found the lower one
        tally[0] += int((np.abs(rel) <= 1e-7).sum()); tally[1] += int((rel < -1e-7).sum()); tally[2] += int((rel > 1e-7).sum())
        if (rel > 1e-7).any():
            worst = max(worst, float(rel.max()))
    n_map += 1
print('route stress ok: %d lattice panels bit for bit; %d MAP panels: %d series where both routes end at the same objective (1e-7), %d where the direct solver ends lower, %d where the continuation does (worst %.2e relative)' % (n_lat, n_map, tally[0], tally[1], tally[2], worst))
