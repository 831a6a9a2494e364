import os, sys, numpy as np
sys.path.insert(0, '.')
from time_series_spark_amd import _lib, synth, forecaster as fc
N, T = 12000, 90
ds, y = synth.make_panel(N, T, 'linear', seed=99)
spec = fc.ModelSpec(growth='linear', seasonalities=fc.ModelSpec.auto_seasonalities(ds), algorithm=_lib.ALGO_NEWTON)
sub = np.arange(0, N, 5)
def one_per_wave():
    small = [fc.fit_aligned(spec, ds, y[sub[i::3]]) for i in range(3)]
    ref = np.zeros((len(sub), small[0].theta.shape[1]))
    for i in range(3):
        ref[i::3] = small[i].theta
    return ref
ref = one_per_wave()
def nd(r):
    return int((~(r.theta[sub] == ref).all(axis=1)).sum())
big_n = int(os.environ.get('BIG_N', 100000))
for rep in range(6):
    if rep == 2:
        ds0, y0 = synth.make_panel(big_n, 90, 'linear', seed=751, dtype=np.float32)
        spec0 = fc.ModelSpec(growth='linear', seasonalities=fc.ModelSpec.auto_seasonalities(ds0), algorithm=_lib.ALGO_AUTO)
        fc.fit_aligned(spec0, ds0, y0)
    big = fc.fit_aligned(spec, ds, y)
    r2 = one_per_wave()
    bad = np.nonzero(~(big.theta[sub] == ref).all(axis=1))[0]
    print('rep', rep, 'slots differ from one-per-wave in', nd(big), 'of', len(sub), '| one-per-wave rerun differs in', int((~(r2 == ref).all(axis=1)).sum()),
          '| n_iter of bad', big.n_iter[sub[bad[:6]]], flush=True)
