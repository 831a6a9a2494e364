"""dev tool: per-phase cycles of the quadratic-form fit kernel (a -DTSF_QUAD_TIMING build of tsf_inst_quad3.hip /
tsf_inst_quad4.hip, tools/build_variant.sh) for waves that run ALONE (one series per CU) and under load.
  TSF_OPTIONS=quad_reg=0 TSF_LIB_PATH=tools/variants/libtsf_amd_qtime.so python tools/quad_lone_timing.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from time_series_spark_amd import forecaster as fc, synth  # noqa: E402

YEARLY = {'name': 'yearly', 'period': 365.25, 'fourier_order': 10}
WEEKLY = {'name': 'weekly', 'period': 7, 'fourier_order': 3}
for N in (200, 3000, 10000):
    ds, y = synth.make_panel(N, 730, 'linear', seed=751)
    spec = fc.ModelSpec(growth='linear', seasonalities=[YEARLY, WEEKLY])
    fc.fit_aligned(spec, ds, y)
    r = fc.fit_aligned(spec, ds, y)
    print('N', N, 'mean evals', r.n_eval.mean(), 'mean iters', r.n_iter.mean(), flush=True)
