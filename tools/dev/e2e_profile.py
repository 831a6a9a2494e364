"""dev: cProfile of the files -> files run (tools/e2e_bench.py's second pass): where the host time of the two jobs goes."""
import cProfile
import os
import pstats
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from time_series_spark_amd import synth  # noqa: E402
from time_series_spark_amd.jobs import prophet_modeler as pm, prophet_scorer as ps  # noqa: E402
import e2e_bench  # noqa: E402

N, T = 10000, 730
ds, y = synth.make_panel(N, T, 'linear', seed=2)
work = tempfile.mkdtemp(prefix='tsf_e2e_')
e2e_bench.write_input(os.path.join(work, 'model-input'), ds, y)
mcfg = {'io': {'input': os.path.join(work, 'model-input'), 'models': os.path.join(work, 'models')},
        'model': {'floor': 0, 'cap_multiplier': 1.1, 'prophet': {'growth': 'linear', 'seasonality_mode': 'additive', 'yearly_seasonality': True}}}
scfg = {'io': {'models': mcfg['io']['models'], 'forecasts': os.path.join(work, 'forecasts')}, 'forecast': {'periods': 90, 'frequency': 'D'}}


def one_pass():
    mo = pm.ProphetModeler(mcfg)
    prev = pm.previous_run_cost(mcfg['io']['models'])
    cols = mo.read_input_columns()
    models = pm.model_arrays(mcfg, previous=prev)(*cols)
    mo.persist_models(models)
    sc = ps.ProphetScorer(scfg)
    mdf = sc.read_model_dataframe()
    fdf = ps.forecast_panel(scfg)(mdf)
    sc.write_converted(fdf)


out, sys.stdout = sys.stdout, open(os.devnull, 'w')
one_pass()
pr = cProfile.Profile()
pr.enable()
one_pass()
pr.disable()
sys.stdout = out
pstats.Stats(pr).sort_stats('tottime').print_stats(32)
shutil.rmtree(work, ignore_errors=True)
