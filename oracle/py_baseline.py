"""CPU baseline SHAPED like the reference's path -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's
`cpu_baseline_python` leg; the product never imports this).

The reference fits one series per grouped-map UDF call inside a Python worker process, one worker per
core under Spark `local[*]` (/root/reference/tests/unit/prophet_modeler_test.py:20;
/root/reference/src/jobs/prophet_modeler.py:56-75: pandas frame in -> `Prophet(...)` -> `.fit(pdf)` ->
one row out; /root/reference/src/jobs/prophet_scorer.py:64-70: `make_future_dataframe` + `predict`).
fbprophet 0.5 / pystan 2.19 cannot be installed here (PARITY UNPINNED, see oracle/fbprophet_restated.py),
so the stand-in is the literal restatement in the same shape: per series a pandas DataFrame ->
`ProphetOracle(...)` (setup_dataframe, Fourier features, changepoints: pandas / numpy, method by method
as fbprophet does them) -> `.fit(df)` (Stan's L-BFGS restated in C, oracle/stan_lbfgs.c, on the literal
dense-A log-posterior: compiled code, as Stan's optimiser is) -> `make_future_dataframe` -> `predict`,
under `multiprocessing.Pool(n_procs)` with one series per task.

What it is NOT: fbprophet + Stan.  Stan evaluates the model through its autodiff tape and pystan
marshals the data per call; both are slower than this.  Label: "restated python -- not fbprophet+Stan;
a reported baseline, not the optimisation target".

Run as its own process (bench.py starts it with subprocess: no fork of a process that holds a HIP
context): python -m oracle.py_baseline --series 512 --procs 256 [--points 730 --horizon 90 --seed 751]
prints one JSON line.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_G = {}


def _setup(points, seed, n_total, horizon, model, series):
    """In the PARENT, before the pool forks: the panel (the workers share its pages) and the imports."""
    import pandas as pd
    from time_series_spark_amd import synth
    from oracle import fbprophet_restated  # noqa: F401
    ds, y = synth.make_panel(n_total, points, 'linear' if model == 'cfg2' else 'logistic', seed=seed)
    _G.update(ds=pd.to_datetime(ds), y=np.ascontiguousarray(y[:series]), horizon=horizon, model=model)


def _one(n):
    """One series, the way the reference's two UDFs treat it: frame in, fitted model, 90-step forecast out."""
    import pandas as pd
    from oracle.fbprophet_restated import ProphetOracle
    y = _G['y'][n]
    df = pd.DataFrame({'ds': _G['ds'], 'y': y})
    if _G['model'] == 'cfg2':
        # BASELINE cfg2: linear trend, additive weekly + yearly (forced on: 730 daily points span 729 d)
        m = ProphetOracle(growth='linear', seasonality_mode='additive', yearly_seasonality=True,
                          weekly_seasonality=True, daily_seasonality=False)
    else:
        # the reference's own settings (prophet_modeler.py:56-65): floor 0, cap = 1.1 max(y), logistic, multiplicative
        df['floor'] = 0.0
        df['cap'] = float(y.max()) * 1.1
        m = ProphetOracle(growth='logistic', seasonality_mode='multiplicative')
    try:
        m.fit(df)
    except RuntimeError:            # the reference prints and drops the series (prophet_modeler.py:81-85)
        return 0, 0.0
    fut = m.make_future_dataframe(periods=_G['horizon'], freq='D', include_history=False)
    if _G['model'] != 'cfg2':
        fut['floor'] = 0.0
        fut['cap'] = float(y.max()) * 1.1
    yhat = m.predict(fut)['yhat'].values
    return int(m.fit_info.get('n_eval', 0)), float(yhat[-1])


def run(series, procs, points=730, horizon=90, seed=751, n_total=10000, model='cfg2'):
    from oracle import oracle_lib
    oracle_lib.build()              # before the fork: the workers only load it
    series = min(series, n_total)
    _setup(points, seed, n_total, horizon, model, series)
    with mp.get_context('fork').Pool(procs) as pool:
        pool.map(_one, range(min(procs, series)), chunksize=1)      # warm-up: library load in every worker
        t0 = time.perf_counter()
        res = pool.map(_one, range(series), chunksize=1)
        dt = time.perf_counter() - t0
    ev = [r[0] for r in res]
    return {'value': series / dt, 'unit': 'series/s', 'cores': procs, 'kind': 'port',
            'label': 'restated python -- not fbprophet+Stan; a reported baseline, not the optimisation target',
            'sample': '%d of %d series of the same panel, one per task on multiprocessing.Pool(%d): pandas frame -> '
                      'ProphetOracle(...).fit (oracle/fbprophet_restated.py + oracle/stan_lbfgs.c, residual form) -> '
                      'make_future_dataframe -> predict (%d steps); %.2f s wall' % (series, n_total, procs, horizon, dt),
            'mean_evals': float(np.mean(ev)) if ev else None, 'seconds': dt, 'series': series}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--series', type=int, default=512)
    ap.add_argument('--procs', type=int, default=os.cpu_count() or 1)
    ap.add_argument('--points', type=int, default=730)
    ap.add_argument('--horizon', type=int, default=90)
    ap.add_argument('--seed', type=int, default=751)
    ap.add_argument('--total', type=int, default=10000)
    ap.add_argument('--model', default='cfg2', choices=['cfg2', 'reference'])
    a = ap.parse_args()
    print(json.dumps(run(a.series, a.procs, a.points, a.horizon, a.seed, a.total, a.model)))


if __name__ == '__main__':
    main()
