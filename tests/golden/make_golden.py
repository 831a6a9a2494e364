"""Generates the committed golden vectors under tests/golden/ (run in the build container):

  fixture_751.npz     inputs = the reference's only data fixture
                      (/root/reference/tests/fixtures/model-input/series_id=751/sample-model-input.csv,
                      816 rows, two dim_ids, irregular timestamps with duplicates) + outputs of the
                      canonical CPU oracle for the reference's own settings (logistic + floor 0,
                      cap = 1.1 max(y), multiplicative, auto seasonalities; 40 x 15min horizon as in
                      /root/reference/tests/unit/prophet_scorer_test.py:38-39).
  synthetic_cases.npz canonical-oracle outputs for the small synthetic cases of tests/helpers.CASES.
  newton_cases.npz    canonical-oracle outputs of Stan's Newton optimiser on two short panels.

PARITY UNPINNED: outputs come from the restated oracle, NOT from fbprophet/pystan (which cannot
be installed here); the reference's tests pin no numeric value.
"""
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import helpers  # noqa: E402
from oracle import canon_lib as cl  # noqa: E402
from time_series_spark_amd import forecaster as fc, panel as pk  # noqa: E402

REF_FIXTURE = '/root/reference/tests/fixtures/model-input/series_id=751/sample-model-input.csv'


def fixture():
    df = pd.read_csv(REF_FIXTURE, header=None, names=['dim_id', 'ds', 'y'])
    df['series_id'] = 751
    df['ds'] = pd.to_datetime(df['ds'])
    raw = {'raw_dim_id': df['dim_id'].values.astype(np.int32),
           'raw_ds_ns': df['ds'].values.astype('datetime64[ns]').astype(np.int64),
           'raw_y': df['y'].values.astype(np.int32)}
    p = pk.pack_long_frame(df)
    span, min_dt, ymax = pk.per_series_stats(p)
    out = dict(raw)
    out['offsets'] = p.offsets
    out['dim_ids'] = p.keys['dim_id'].values.astype(np.int32)
    cap = ymax * 1.1
    out['cap'] = cap
    thetas, yhats, iters, stats, futs = [], [], [], [], []
    for n in range(p.N):
        a, b = p.offsets[n], p.offsets[n + 1]
        seas = fc.ModelSpec.auto_from_stats(int(span[n]), int(min_dt[n]),
                                            seasonality_mode='multiplicative')
        spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=seas)
        csp = helpers.oracle_spec(spec)
        r = cl.fit(csp, p.ds_ns[a:b], p.y[a:b], 0.0, cap[n])
        fut = pk.future_dates(p.ds_ns[b - 1:b], 40, '15min')[0]
        # the scorer reads floor/cap back from float32 columns (prophet_scorer.py:46-47)
        yh, _ = cl.predict(csp, r, fut, float(np.float32(0.0)), float(np.float32(cap[n])))
        thetas.append(r['theta']); yhats.append(yh); iters.append(r['n_iter']); stats.append(r['status'])
        futs.append(fut)
        print('fixture dim', out['dim_ids'][n], 'K', len(r['theta']) - 28, r['status_name'], r['n_iter'], yh[:3])
    out['theta'] = np.array(thetas); out['yhat'] = np.array(yhats)
    out['n_iter'] = np.array(iters); out['status'] = np.array(stats); out['fut'] = np.array(futs)
    np.savez_compressed(os.path.join(HERE, 'fixture_751.npz'), **out)


def synthetic():
    out = {}
    for name in list(helpers.CASES) + helpers.RESID_VARIANTS:
        spec, ds, y, floor, cap, extra, fut, extra_future = helpers.make_case(name)
        csp = helpers.oracle_spec(spec)
        th, yh, it, st, ev = [], [], [], [], []
        for n in range(y.shape[0]):
            r = cl.fit(csp, ds, y[n], floor[n], cap[n], extra)
            yo, _ = cl.predict(csp, r, fut, floor[n], cap[n], extra_future)
            pad = np.zeros(spec.theta_stride); pad[:3 + r['info'].S] = r['theta'][:3 + r['info'].S]
            pad[3 + spec.n_changepoints:] = r['theta'][3 + r['info'].S:]
            th.append(pad); yh.append(yo); it.append(r['n_iter']); st.append(r['status']); ev.append(r['n_eval'])
        out[name + '/theta'] = np.array(th); out[name + '/yhat'] = np.array(yh)
        out[name + '/n_iter'] = np.array(it); out[name + '/status'] = np.array(st)
        out[name + '/n_eval'] = np.array(ev)
        print(name, 'iters', it, 'status', st)
    np.savez_compressed(os.path.join(HERE, 'synthetic_cases.npz'), **out)


def newton_cases():
    """Stan's Newton optimiser (fbprophet's choice below 100 rows) on two short panels:
    newton_cases.npz = canonical-oracle (cn_newton) outputs.  Inputs are regenerated from the seed
    by the tests (synth.make_panel(4, T, growth, seed=751))."""
    from time_series_spark_amd import synth
    out = {}
    for growth, mode, T in (('linear', 'additive', 60), ('logistic', 'multiplicative', 90)):
        ds, y = synth.make_panel(4, T, growth, seed=751)
        seas = fc.ModelSpec.auto_seasonalities(ds, seasonality_mode=mode)
        spec = fc.ModelSpec(growth=growth, seasonality_mode=mode, seasonalities=seas)
        csp = helpers.oracle_spec(spec)
        fut = ds[-1] + helpers.DAY_NS * np.arange(1, 31)
        cap = y.max(axis=1) * 1.1
        th, yh, it, ev, st = [], [], [], [], []
        for n in range(4):
            r = cl.fit_newton(csp, ds, y[n], 0.0, cap[n])
            yo, _ = cl.predict(csp, r, fut, 0.0, cap[n])
            pad = np.zeros(spec.theta_stride); S = r['info'].S
            pad[:3 + S] = r['theta'][:3 + S]; pad[3 + spec.n_changepoints:] = r['theta'][3 + S:]
            th.append(pad); yh.append(yo); it.append(r['n_iter']); ev.append(r['n_eval']); st.append(r['status'])
        key = '%s_%d' % (growth, T)
        out[key + '/theta'] = np.array(th); out[key + '/yhat'] = np.array(yh)
        out[key + '/n_iter'] = np.array(it); out[key + '/n_eval'] = np.array(ev); out[key + '/status'] = np.array(st)
        print('newton', key, 'iters', it, 'status', st)
    np.savez_compressed(os.path.join(HERE, 'newton_cases.npz'), **out)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'newton':
        newton_cases()
    else:
        fixture()
        synthetic()
        newton_cases()
