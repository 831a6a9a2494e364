"""The TRUE MAP (oracle/true_map.py) of the first series of a BASELINE-shaped panel, in processes of its own -- CPU only,
TEST INFRASTRUCTURE: what tests/test_gpu_literal.py compares the library's converge = MAP fits with, and bench.py's
parity_context.map_mode.  Writes an .npz: theta_map [n][stride] in the library's layout, kkt [n], f_map [n].

    python tools/true_map_solve.py <cfg2|ref|cfg4|cfg5> <n_series> <out.npz> [--from-stan theta.npy] [--panel panel.npz]

--panel: ds [T], y [n][T] (and cap [n] for logistic growth) of the series to solve, instead of the kind's own synthetic
panel (make_panel draws a panel from ONE random stream: the first n series of a 10 000-series panel are not the n-series
panel; bench.py hands over the rows it fitted).

The solver starts from fbprophet's initial values (and, with --from-stan, also from the given stopped fits: the lower of
the two optima is kept; they agree to 1e-10 in objective).  The literal model runs on the canonical design values
(tests/helpers.literal_on_canonical_design), as in tools/true_map_report.py."""
import multiprocessing as mp
import os
import sys

for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
    os.environ.setdefault(_v, '1')

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
YEARLY = {'name': 'yearly', 'period': 365.25, 'fourier_order': 10}
WEEKLY = {'name': 'weekly', 'period': 7, 'fourier_order': 3}


def panel(kind, n):
    """-> ds, y [n][T] float64, cap [n] (0 for linear growth), ProphetOracle keyword arguments, holidays frame or None"""
    from time_series_spark_amd import synth
    import pandas as pd
    if kind == 'cfg2':
        ds, y = synth.make_panel(n, 730, 'linear', seed=751)
        return ds, y.astype(np.float64), np.zeros(n), dict(growth='linear', seasonality_mode='additive', yearly_seasonality=True,
                                                           weekly_seasonality=True, daily_seasonality=False), None
    if kind == 'ref':
        ds, y = synth.make_panel(n, 730, 'logistic', seed=751)
        return ds, y.astype(np.float64), y.max(axis=1) * 1.1, dict(growth='logistic', seasonality_mode='multiplicative',
                                                                   yearly_seasonality=True, weekly_seasonality=True,
                                                                   daily_seasonality=False), None
    if kind == 'cfg5':
        ds, y = synth.make_panel(n, 90, 'linear', seed=751, dtype=np.float32)
        return ds, y.astype(np.float64), np.zeros(n), dict(growth='linear', seasonality_mode='additive', yearly_seasonality=False,
                                                           weekly_seasonality=True, daily_seasonality=False), None
    if kind == 'cfg4':
        ds = synth.daily_grid(730)
        fut = ds[-1] + synth.DAY_NS * np.arange(1, 91)
        allm, names = synth.holiday_matrix(np.concatenate([ds, fut]), 10)
        ex = np.ascontiguousarray(allm[:, :730])
        _, y = synth.make_panel(n, 730, 'logistic', seed=751, holidays=ex)
        # the holidays frame whose make_holiday_features columns are synth.holiday_matrix's (name "h<k>", window [-1, +1])
        hol = synth.holiday_frame(np.concatenate([ds, fut]), 10)
        return ds, y.astype(np.float64), y.max(axis=1) * 1.1, dict(growth='logistic', seasonality_mode='multiplicative',
                                                                   yearly_seasonality=True, weekly_seasonality=True,
                                                                   daily_seasonality=False), (ex, names, hol)
    raise SystemExit('unknown kind ' + kind)


def _one(args):
    kind, n, ds, yn, capn, kw, hol, th_stan = args
    import pandas as pd
    from tests import helpers
    from oracle import true_map
    from oracle.fbprophet_restated import ProphetOracle
    with helpers.literal_on_canonical_design():
        m = ProphetOracle(**kw) if hol is None else ProphetOracle(holidays=hol[2], **kw)
        df = pd.DataFrame({'ds': pd.to_datetime(ds), 'y': yn})
        if kw['growth'] == 'logistic':
            df['floor'], df['cap'] = 0.0, capn
        dat, th0 = m.stan_data(df)
    best, info = true_map.solve(dat, th0)
    if th_stan is not None:
        S, K = int(dat['S']), int(dat['K'])
        t2 = np.concatenate([th_stan[:3 + S], th_stan[len(th_stan) - K:]])
        b2, i2 = true_map.solve(dat, t2)
        if i2['f'] < info['f']:
            best, info = b2, i2
    return n, best, info['kkt'], info['f'], int(dat['S']), int(dat['K'])


def main():
    kind, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    opts = dict(zip(sys.argv[4::2], sys.argv[5::2]))
    stan = np.load(opts['--from-stan']) if '--from-stan' in opts else None
    ds, y, cap, kw, hol = panel(kind, 2 if '--panel' in opts else n)
    if '--panel' in opts:
        z = np.load(opts['--panel'])
        ds, y = z['ds'], np.asarray(z['y'], dtype=np.float64)[:n]
        cap = np.asarray(z['cap'], dtype=np.float64)[:n] if 'cap' in z.files else np.zeros(n)
    jobs = [(kind, i, ds, y[i], float(cap[i]), kw, hol, None if stan is None else stan[i]) for i in range(n)]
    with mp.Pool(min(os.cpu_count(), 64, n)) as pool:
        rows = pool.map(_one, jobs, chunksize=1)
    S, K = rows[0][4], rows[0][5]
    n_cp = 25
    theta = np.zeros((n, 3 + n_cp + K))
    for i, th, _k, _f, s_, _kk in rows:
        theta[i, :3 + s_] = th[:3 + s_]
        theta[i, 3 + n_cp:] = th[3 + s_:]
    np.savez(out, theta_map=theta, kkt=np.array([r[2] for r in rows]), f_map=np.array([r[3] for r in rows]), S=S, K=K)


if __name__ == '__main__':
    main()
