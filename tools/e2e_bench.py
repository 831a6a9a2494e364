"""End-to-end run of the reference's two jobs through the drop-in entry points, from files to
files: Hive-partitioned model-input CSV -> ProphetModeler.model (read, pack, GPU fit, model parquet)
-> ProphetScorer.score (read models, GPU predict, convert, forecast CSV).  Prints the wall time of the
two jobs as their drivers run them (round 6: each a pipeline over chunks, time_series_spark_amd/pipeline.py)
and the series/s of the whole thing (host IO included -- this is NOT bench.py's `value`, which times
the hot path with inputs resident in HBM); with --stages also the jobs' steps called one after the
other, each timed (the round-5 arrangement: no overlap).

    python tools/e2e_bench.py [--n 10000] [--t 730] [--kind cfg2|reference] [--fake-gpu] [--stages] [--chunks K]
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


def run(n=10000, t=730, kind='cfg2', passes=3, chunks=None, no_hints=False, stages=False, keep=False, fake_gpu=False,
        quiet=True):
    """Writes the synthetic Hive-partitioned input, runs the two jobs `passes` times after a warm-up pass, returns the
    result dict (times per pass, median / best series per second from files to files)."""
    if fake_gpu:
        import host_profile as hp
        from time_series_spark_amd import forecaster as fc
        fc.fit_aligned, fc.fit_ragged, fc.predict = hp.fake_fit_aligned, hp.fake_fit_ragged, hp.fake_predict
    linear = kind == 'cfg2'
    ds, y = synth.make_panel(n, t, 'linear' if linear else 'logistic', seed=2)
    work = tempfile.mkdtemp(prefix='tsf_e2e_')
    tm = {}
    t0 = time.time()
    write_input(os.path.join(work, 'model-input'), ds, y)
    tm['(write synthetic input)'] = time.time() - t0
    mcfg = {'io': {'input': os.path.join(work, 'model-input'), 'models': os.path.join(work, 'models')},
            'model': {'floor': 0, 'cap_multiplier': 1.1}}
    if chunks is not None:
        mcfg['io']['chunks'] = chunks if chunks == 'auto' else int(chunks)
    if no_hints:
        mcfg['model']['schedule_from_previous_models'] = False
    if linear:
        mcfg['model']['prophet'] = {'growth': 'linear', 'seasonality_mode': 'additive',
                                    'yearly_seasonality': True}
    scfg = {'io': {'models': mcfg['io']['models'], 'forecasts': os.path.join(work, 'forecasts')},
            'forecast': {'periods': 90, 'frequency': 'D'}}

    def stage(name, f, *args):
        t0 = time.time()
        r = f(*args)
        tm[name] = time.time() - t0
        return r

    devnull = open(os.devnull, 'w')
    out = sys.stdout
    if quiet:
        sys.stdout = devnull          # the jobs print one line per chunk and per dropped series
    jobs = []
    try:
        # the two jobs as their drivers run them (modeler_driver / scorer_driver): first pass pays library load + HIP init
        for rep in ['warm-up '] + ['pass %d ' % i for i in range(passes)]:
            t0 = time.time()
            pm.ProphetModeler.model(None, mcfg, return_frame=False)
            t1 = time.time()
            ps.ProphetScorer.score(None, scfg)
            t2 = time.time()
            tm[rep + 'ProphetModeler.model'] = t1 - t0
            tm[rep + 'ProphetScorer.score'] = t2 - t1
            if not rep.startswith('warm'):
                jobs.append((t1 - t0, t2 - t1))
        models = pd.read_parquet(mcfg['io']['models'])
        n_fc = sum(sum(1 for _ in open(os.path.join(scfg['io']['forecasts'], f))) - 1
                   for f in os.listdir(scfg['io']['forecasts']) if f.endswith('.csv'))
        for rep in (('warm-up ', '') if stages else ()):
            mo = pm.ProphetModeler(mcfg)
            # what ProphetModeler.model does: the previous run's models (none in the first pass) give this run its
            # scheduling hints before they are overwritten
            prev = stage(rep + 'previous_run_cost (scheduling hints)', pm.previous_run_cost, mcfg['io']['models']) \
                if not no_hints else None
            cols = stage(rep + 'read_input_columns', mo.read_input_columns)
            mdl = stage(rep + 'model_arrays (pack + fit + blobs)', lambda c: pm.model_arrays(mcfg, previous=prev)(*c), cols)
            stage(rep + 'persist_models', mo.persist_models, mdl)
            sc = ps.ProphetScorer(scfg)
            mdf = stage(rep + 'read_model_dataframe', sc.read_model_dataframe)
            fdf = stage(rep + 'forecast_panel (predict)', ps.forecast_panel(scfg), mdf)
            # (ProphetScorer.score: convert_forecasts is a lazy plan in the reference; the native sink formats the
            # converted rows from the forecast columns, the converted frame is never built)
            stage(rep + 'write_converted (convert + native sink)', sc.write_converted, fdf)
    finally:
        sys.stdout = out
        if not keep:
            shutil.rmtree(work, ignore_errors=True)
    tot = sorted(m + s for m, s in jobs)
    best, med = tot[0], tot[len(tot) // 2]
    res = {'n_series': n, 'T': t, 'kind': kind, 'fake_gpu': fake_gpu, 'models': int(len(models)),
           'forecast_rows': int(n_fc), 'total_s': round(med, 4), 'total_s_best': round(best, 4),
           'modeler_s': round(sorted(m for m, _ in jobs)[len(jobs) // 2], 4),
           'scorer_s': round(sorted(s for _, s in jobs)[len(jobs) // 2], 4),
           'series_per_s_files_to_files': round(n / med, 1), 'series_per_s_best_pass': round(n / best, 1),
           'passes': passes, 'chunks': mcfg['io'].get('chunks', 'auto'),
           'stages_s': {k: round(v, 4) for k, v in tm.items()}}
    if stages:
        res['total_s_stages_one_after_the_other'] = round(sum(v for k, v in tm.items() if not k.startswith(('warm-up', '(', 'pass')) and 'Prophet' not in k), 4)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=10000)
    ap.add_argument('--t', type=int, default=730)
    ap.add_argument('--kind', default='cfg2', choices=['cfg2', 'reference'])
    ap.add_argument('--no-hints', action='store_true', help='do not schedule the second pass from the first pass\'s models')
    ap.add_argument('--fake-gpu', action='store_true', help='stand-ins for the GPU calls (host cost only)')
    ap.add_argument('--keep', action='store_true')
    ap.add_argument('--stages', action='store_true', help='also time the steps of the two jobs one after the other')
    ap.add_argument('--chunks', default=None, help="io.chunks of the modeler config ('auto' by default; 1 = no pipeline)")
    ap.add_argument('--passes', type=int, default=3, help='timed passes of the two jobs (the best and the median are reported)')
    a = ap.parse_args()
    res = run(a.n, a.t, a.kind, a.passes, a.chunks, a.no_hints, a.stages, a.keep, a.fake_gpu)
    for k, v in res['stages_s'].items():
        print('%-44s %8.3f s' % (k, v))
    print(json.dumps(res))


if __name__ == '__main__':
    main()
