"""Pins the oracle to REAL fbprophet -- wherever fbprophet==0.5 (+ pystan==2.19.1.1, the
reference's environment.yml:12-13) is installed.  It cannot run in the build container (no
network, no wheels), which is why parity is labelled "unpinned" in oracle/ and DESIGN.md; this
script makes pinning one command away:

    conda env create -f /path/to/reference/environment.yml && conda activate <env>
    python tests/golden/make_fbprophet_goldens.py            # writes tests/golden/fbprophet_goldens.npz
    python -m pytest tests/test_oracle.py -k real_fbprophet   # reads it, reports the error distribution

It needs only fbprophet, pandas and numpy plus two dependency-free modules of this repo
(time_series_spark_amd/synth.py, tests/helpers.py: the seeded inputs).  For every case of
tests/helpers.CASES (N = 6 series each) and, when the reference checkout is given with
--reference, its own fixture (series_id=751, two dim_ids, the reference's settings) it stores what
`Prophet(...).fit(df)` / `.predict(future)` return: params k, m, delta, sigma_obs, beta, the scaling
(y_scale, start, t_scale), changepoints_t, the seasonality table, and yhat / trend on the horizon
the oracle tests use.  Reference call sites mirrored: prophet_modeler.py:56-66 (floor, cap,
constructor, fit), prophet_scorer.py:64-70 (make_future_dataframe, floor/cap, predict).
"""
import argparse
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def holidays_frame(names, dates, matrix):
    """fbprophet-style holidays frame from the 0/1 indicator columns of synth.holiday_matrix
    (names '<holiday>_delim_<+|-><offset>', window [-1, +1])."""
    rows = []
    for e, nm in enumerate(names):
        name, off = nm.split('_delim_')
        if int(off) == 0:
            rows += [(name, d) for d in dates[matrix[e] == 1.0]]
    h = pd.DataFrame(rows, columns=['holiday', 'ds'])
    h['lower_window'], h['upper_window'] = -1, 1
    return h


def fit_predict(Prophet, df, fut, growth, mode, yearly, weekly, daily, holidays=None):
    m = Prophet(growth=growth, seasonality_mode=mode, yearly_seasonality=yearly,
                weekly_seasonality=weekly, daily_seasonality=daily, holidays=holidays)
    m.fit(df)
    fc = m.predict(fut)
    tcc = getattr(m, 'train_component_cols', None)      # fbprophet 0.5: a DataFrame indexed by design column
    cols = [str(c) for c in tcc.index] if isinstance(tcc, pd.DataFrame) else []
    return {'k': float(m.params['k'][0]), 'm': float(m.params['m'][0]),
            'sigma_obs': float(m.params['sigma_obs'][0]),
            'delta': np.asarray(m.params['delta'][0], dtype=np.float64),
            'beta': np.asarray(m.params['beta'][0], dtype=np.float64),
            'y_scale': float(m.y_scale), 'start_ns': int(pd.Timestamp(m.start).value),
            't_scale_ns': int(pd.Timedelta(m.t_scale).value),
            'changepoints_t': np.asarray(m.changepoints_t, dtype=np.float64),
            'columns': np.array(cols, dtype=str),
            'yhat': fc['yhat'].values.astype(np.float64), 'trend': fc['trend'].values.astype(np.float64)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default=None, help='checkout of mageky/time-series-spark (for its fixture)')
    ap.add_argument('--out', default=os.path.join(HERE, 'fbprophet_goldens.npz'))
    args = ap.parse_args()
    import fbprophet
    from fbprophet import Prophet
    from tests import helpers
    from time_series_spark_amd import synth
    out = {'fbprophet_version': np.array(fbprophet.__version__)}
    for case, (growth, mode, T, seas, nh) in helpers.CASES.items():
        N, H = 6, 30
        ds = synth.daily_grid(T)
        fut_ns = ds[-1] + helpers.DAY_NS * np.arange(1, H + 1)
        hol = hmat = None
        if nh:
            hmat, names = synth.holiday_matrix(np.concatenate([ds, fut_ns]), nh)
            hol = holidays_frame(names, pd.to_datetime(np.concatenate([ds, fut_ns])), hmat)
            hmat = hmat[:, :T]
        ds, y = synth.make_panel(N, T, 'linear' if growth == 'linear' else 'logistic', seed=21, holidays=hmat)
        orders = {s['name']: s['fourier_order'] for s in seas}
        yearly = orders.get('yearly', False)
        yearly = True if yearly == 10 else yearly
        for n in range(N):
            df = pd.DataFrame({'ds': pd.to_datetime(ds), 'y': y[n]})
            fut = pd.DataFrame({'ds': pd.to_datetime(fut_ns)})
            if growth == 'logistic':
                df['floor'], df['cap'] = 0.0, float(y[n].max() * 1.1)
                fut['floor'], fut['cap'] = 0.0, float(y[n].max() * 1.1)
            r = fit_predict(Prophet, df, fut, growth, mode, yearly, 'weekly' in orders, False, hol)
            for k, v in r.items():
                out['%s/%d/%s' % (case, n, k)] = v
            print(case, n, 'k %.6g m %.6g yhat[:2]' % (r['k'], r['m']), r['yhat'][:2], flush=True)
    if args.reference:
        # the reference's own fixture with the reference's own settings and test horizon
        # (prophet_modeler.py:56-66; prophet_scorer_test.py:38-39: 40 periods of 15min)
        path = os.path.join(args.reference, 'tests/fixtures/model-input/series_id=751/sample-model-input.csv')
        raw = pd.read_csv(path, header=None, names=['dim_id', 'ds', 'y'])
        raw['ds'] = pd.to_datetime(raw['ds'])
        for dim, df in raw.groupby('dim_id'):
            df = df[['ds', 'y']].copy()
            cap = float(df['y'].max() * 1.1)
            df['floor'], df['cap'] = 0.0, cap
            m = Prophet(growth='logistic', seasonality_mode='multiplicative')
            m.fit(df)
            fut = m.make_future_dataframe(periods=40, freq='15min', include_history=False)
            fut['floor'], fut['cap'] = float(np.float32(0.0)), float(np.float32(cap))
            fcst = m.predict(fut)
            out['fixture_751/%d/yhat' % dim] = fcst['yhat'].values.astype(np.float64)
            out['fixture_751/%d/k' % dim] = float(m.params['k'][0])
            out['fixture_751/%d/m' % dim] = float(m.params['m'][0])
            out['fixture_751/%d/delta' % dim] = np.asarray(m.params['delta'][0])
            out['fixture_751/%d/beta' % dim] = np.asarray(m.params['beta'][0])
            out['fixture_751/%d/seasonalities' % dim] = np.array(list(m.seasonalities), dtype=str)
            print('fixture dim', dim, list(m.seasonalities), fcst['yhat'].values[:3], flush=True)
    np.savez_compressed(args.out, **out)
    print('wrote', args.out)


if __name__ == '__main__':
    main()
