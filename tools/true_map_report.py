"""Three-way distance to the TRUE MAP (round-4 review, item 3).  CPU only (the canonical oracle IS the GPU's arithmetic,
bit for bit: tests/test_gpu_parity.py), so this runs anywhere.

For every sampled series: the true MAP of the literal posterior (oracle/true_map.py: delta split into its positive and
negative parts, L-BFGS-B with bounds, from TWO starting points that must agree), and the 90-step forecasts of
  (a) the headline arithmetic -- Stan's L-BFGS on the quadratic form of the likelihood (what fit_quad_kernel runs),
  (b) Stan's L-BFGS in Stan's own order of operations (the residual form),
  (c) (a) with ONE input value moved by one ulp,
each against the forecast at the MAP: per series the median over the horizon of |yhat - yhat_MAP| / |yhat_MAP|, then
median / p90 over the series; and the objective gaps f(stopped) - f(MAP).

    python tools/true_map_report.py [n_cfg2=256] [n_ref=64] > profiles/r05_true_map/report.json
"""
import json
import multiprocessing as mp
import os
import sys

for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):      # one series per process: no BLAS threads under the pool
    os.environ.setdefault(_v, '1')

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H = 90
DAY = 86400 * 10 ** 9
YEARLY = {'name': 'yearly', 'period': 365.25, 'fourier_order': 10}
WEEKLY = {'name': 'weekly', 'period': 7, 'fourier_order': 3}


def _one(args):
    kind, n, ds, yn, capn = args
    import pandas as pd
    from tests import helpers
    from oracle import canon_lib as cl, true_map
    from oracle.fbprophet_restated import ProphetOracle
    from time_series_spark_amd import forecaster as fc
    growth, mode = ('linear', 'additive') if kind == 'cfg2' else ('logistic', 'multiplicative')
    spec = fc.ModelSpec(growth=growth, seasonality_mode=mode, seasonalities=[dict(YEARLY), dict(WEEKLY)])
    fut = ds[-1] + DAY * np.arange(1, H + 1)
    with helpers.literal_on_canonical_design():
        m = ProphetOracle(growth=growth, seasonality_mode=mode, yearly_seasonality=True, weekly_seasonality=True,
                          daily_seasonality=False)
        df = pd.DataFrame({'ds': pd.to_datetime(ds), 'y': yn})
        if growth == 'logistic':
            df['floor'], df['cap'] = 0.0, capn
        dat, th0 = m.stan_data(df)
    fits = {}
    forms = (('quadratic', 1, yn), ('residual', 0, yn), ('one_ulp', 1, None)) if kind == 'cfg2' else \
            (('residual', 0, yn), ('one_ulp', 0, None))
    for tag, em, yy in forms:
        if yy is None:
            yy = yn.copy()
            yy[len(yy) // 2] = np.nextafter(yy[len(yy) // 2], np.inf)
        csp = helpers.oracle_spec(spec)
        csp.eval_mode = em
        fits[tag] = cl.fit(csp, ds, yy, 0.0, capn)
    base = fits['quadratic' if kind == 'cfg2' else 'residual']
    th_a, info_a = true_map.solve(dat, base['theta'])
    th_b, info_b = true_map.solve(dat, th0)
    csp = helpers.oracle_spec(spec)

    def yhat(theta):
        o = dict(base)
        o['theta'] = theta
        return cl.predict(csp, o, fut, 0.0, capn)[0]
    best, info = (th_a, info_a) if info_a['f'] <= info_b['f'] else (th_b, info_b)
    ym = yhat(best)
    out = {'kind': kind, 'n': int(n), 'f_map': info['f'], 'kkt': info['kkt'],
           'two_starts_forecast_rel': float(np.max(np.abs(yhat(th_a) - yhat(th_b)) / np.abs(ym))),
           'two_starts_f_gap': abs(info_a['f'] - info_b['f'])}
    for tag, o in fits.items():
        from oracle import oracle_lib
        f_stop = oracle_lib.neg_log_prob_grad(dat, o['theta'])[0]
        out[tag] = {'rel': float(np.median(np.abs(yhat(o['theta']) - ym) / np.abs(ym))), 'gap': float(f_stop - info['f']),
                    'status': int(o['status']), 'n_iter': int(o['n_iter'])}
    return out


def main():
    n_cfg2 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n_ref = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    from time_series_spark_amd import synth
    jobs = []
    ds, y = synth.make_panel(n_cfg2, 730, 'linear', seed=751)          # (a panel of bench.py's distribution; make_panel draws from ONE stream, so not its first n series)
    jobs += [('cfg2', n, ds, y[n].astype(np.float64), 0.0) for n in range(n_cfg2)]
    ds2, y2 = synth.make_panel(n_ref, 730, 'logistic', seed=751)       # tools/bench_configs.py ref10k: its first n series
    jobs += [('ref', n, ds2, y2[n].astype(np.float64), float(y2[n].max() * 1.1)) for n in range(n_ref)]
    with mp.Pool(min(os.cpu_count(), 64, len(jobs))) as pool:
        rows = pool.map(_one, jobs, chunksize=2)
    rep = {'what': 'distance of stopped fits to the TRUE MAP (oracle/true_map.py); per series the median over a 90-day horizon of '
                   '|yhat - yhat_MAP| / |yhat_MAP|, then median / p90 over series; gap = f(stopped) - f(MAP) >= 0',
           'parity': 'restated oracle, NOT real fbprophet (unpinned)'}
    for kind, label in (('cfg2', 'cfg2_linear_additive_730'), ('ref', 'reference_model_logistic_multiplicative_730')):
        rs = [r for r in rows if r['kind'] == kind]
        if not rs:
            continue
        blk = {'series': len(rs),
               'map_solver': {'kkt_max': max(r['kkt'] for r in rs),
                              'two_starts_forecast_rel_max': max(r['two_starts_forecast_rel'] for r in rs),
                              'two_starts_forecast_rel_median': float(np.median([r['two_starts_forecast_rel'] for r in rs])),
                              'two_starts_f_gap_max': max(r['two_starts_f_gap'] for r in rs)}}
        for tag in ('quadratic', 'residual', 'one_ulp'):
            if tag not in rs[0]:
                continue
            rel = np.array([r[tag]['rel'] for r in rs])
            gap = np.array([r[tag]['gap'] for r in rs])
            blk[tag] = {'forecast_rel_median': float(np.median(rel)), 'forecast_rel_p90': float(np.quantile(rel, 0.9)),
                        'forecast_rel_max': float(rel.max()),
                        'objective_gap_min': float(gap.min()), 'objective_gap_median': float(np.median(gap)),
                        'objective_gap_p90': float(np.quantile(gap, 0.9)), 'objective_gap_max': float(gap.max())}
        rep[label] = blk
    print(json.dumps(rep, indent=1))


if __name__ == '__main__':
    main()
