"""End-to-end run of the reference's two jobs through the drop-in entry points, from files to
files: Hive-partitioned model-input CSV -> ProphetModeler.model's steps (read, pack, GPU fit, model
parquet) -> ProphetScorer.score (read models, GPU predict, convert, forecast CSV).  Prints the
wall time of every stage and the series/s of the whole thing (host IO included -- this is NOT
bench.py's `value`, which times the hot path with inputs resident in HBM).

    python tools/e2e_bench.py [--n 10000] [--t 730] [--kind cfg2|reference] [--fake-gpu]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from time_series_spark_amd import synth  # noqa: E402
from time_series_spark_amd.jobs import prophet_modeler as pm, prophet_scorer as ps  # noqa: E402


def write_input(root, ds, y):
    stamps = pd.DatetimeIndex(ds.astype('datetime64[ns]')).strftime('%Y-%m-%d %H:%M:%S').values
    pre = np.array(['1,' + s + ',' for s in stamps], dtype=object)
    for n in range(y.shape[0]):
        d = os.path.join(root, 'series_id=%d' % n)
        os.makedirs(d)
        with open(os.path.join(d, 'part-00000.csv'), 'w') as f:
            f.write('\n'.join(pre + y[n].astype(np.int64).astype(str).astype(object)))
            f.write('\n')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=10000)
    ap.add_argument('--t', type=int, default=730)
    ap.add_argument('--kind', default='cfg2', choices=['cfg2', 'reference'])
    ap.add_argument('--no-hints', action='store_true', help='do not schedule the second pass from the first pass\'s models')
    ap.add_argument('--fake-gpu', action='store_true', help='stand-ins for the GPU calls (host cost only)')
    ap.add_argument('--keep', action='store_true')
    a = ap.parse_args()
    if a.fake_gpu:
        import host_profile as hp
        from time_series_spark_amd import forecaster as fc
        fc.fit_aligned, fc.fit_ragged, fc.predict = hp.fake_fit_aligned, hp.fake_fit_ragged, hp.fake_predict
    linear = a.kind == 'cfg2'
    ds, y = synth.make_panel(a.n, a.t, 'linear' if linear else 'logistic', seed=2)
    work = tempfile.mkdtemp(prefix='tsf_e2e_')
    t = {}
    t0 = time.time()
    write_input(os.path.join(work, 'model-input'), ds, y)
    t['(write synthetic input)'] = time.time() - t0
    mcfg = {'io': {'input': os.path.join(work, 'model-input'), 'models': os.path.join(work, 'models')},
            'model': {'floor': 0, 'cap_multiplier': 1.1}}
    if linear:
        mcfg['model']['prophet'] = {'growth': 'linear', 'seasonality_mode': 'additive',
                                    'yearly_seasonality': True}
    scfg = {'io': {'models': mcfg['io']['models'], 'forecasts': os.path.join(work, 'forecasts')},
            'forecast': {'periods': 90, 'frequency': 'D'}}

    def stage(name, f, *args):
        t0 = time.time()
        r = f(*args)
        t[name] = time.time() - t0
        return r

    devnull = open(os.devnull, 'w')
    out, sys.stdout = sys.stdout, devnull          # the jobs print one line per dropped series
    try:
        for rep in ('warm-up ', ''):              # first pass pays library load + HIP init
            mo = pm.ProphetModeler(mcfg)
            # what ProphetModeler.model does: the previous run's models (none in the first pass) give this run its
            # scheduling hints before they are overwritten
            prev = stage(rep + 'previous_run_cost (scheduling hints)', pm.previous_run_cost, mcfg['io']['models']) \
                if not a.no_hints else None
            cols = stage(rep + 'read_input_columns', mo.read_input_columns)
            models = stage(rep + 'model_arrays (pack + fit + blobs)', lambda c: pm.model_arrays(mcfg, previous=prev)(*c), cols)
            stage(rep + 'persist_models', mo.persist_models, models)
            sc = ps.ProphetScorer(scfg)
            mdf = stage(rep + 'read_model_dataframe', sc.read_model_dataframe)
            fdf = stage(rep + 'forecast_panel (predict)', ps.forecast_panel(scfg), mdf)
            # (ProphetScorer.score: convert_forecasts is a lazy plan in the reference; the native sink formats the
            # converted rows from the forecast columns, the converted frame is never built)
            stage(rep + 'write_converted (convert + native sink)', sc.write_converted, fdf)
    finally:
        sys.stdout = out
    total = sum(v for k, v in t.items() if not k.startswith(('warm-up', '(')))
    for k, v in t.items():
        print('%-44s %8.3f s' % (k, v))
    res = {'n_series': a.n, 'T': a.t, 'kind': a.kind, 'fake_gpu': a.fake_gpu, 'models': int(len(models)),
           'forecast_rows': int(len(fdf)), 'total_s': round(total, 3),
           'series_per_s_files_to_files': round(a.n / total, 1),
           'stages_s': {k: round(v, 4) for k, v in t.items()}}
    print(json.dumps(res))
    if not a.keep:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    main()
