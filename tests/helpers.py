"""Shared test helpers: oracle <-> product spec conversion, bit comparison, case definitions."""
import contextlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
DAY_NS = 86400 * 10 ** 9


def canonical_fourier(dates, period, series_order):
    """The CANONICAL design values of one seasonality (oracle/prophet_canon.c fourier_row, round 5): the first
    harmonic from the deterministic sincos at fbprophet's argument, the others by the three-term recurrence --
    in fbprophet's column order [sin 1, cos 1, sin 2, ...].  Drop-in for fbprophet_restated.fourier_series."""
    import pandas as pd
    from oracle import canon_lib as cl
    ns = np.ascontiguousarray(pd.DatetimeIndex(dates).asi8, dtype=np.int64)
    pad = np.concatenate([ns, [ns.max() + DAY_NS]])       # (the oracle wants >= 2 rows and a non-zero time span)
    sp = cl.make_spec(seasonalities=[(period, series_order, 'additive', 10.0)], n_changepoints=0)
    return cl.design(sp, pad, np.arange(float(pad.size)))['X'][:ns.size].copy()


@contextlib.contextmanager
def literal_on_canonical_design():
    """Inside: the LITERAL restatement (oracle/fbprophet_restated.py) builds its seasonal features from the canonical
    design values instead of numpy's sin / cos of every harmonic's argument.  The parity statements are split
    (round-4 review, item 1): canonical X against literal X to 1e-9 (test_canonical_eval_matches_literal_stan), and
    log-posterior / gradient / predict on the SAME X to the tolerances they always had (1e-12 / 1e-11 / 16 ulp)."""
    import oracle.fbprophet_restated as fr
    orig = fr.fourier_series
    fr.fourier_series = canonical_fourier
    try:
        yield
    finally:
        fr.fourier_series = orig


class _Routes(object):
    """The route switches of the default context (tsf_set_option; `Context.set_option`), addressed by the names the
    GPU tests used while these were process-wide environment variables read inside the library (until round 4):
    `helpers.routes['TSF_SPARSE_EXTRA'] = '0'` ... `helpers.routes.pop('TSF_SPARSE_EXTRA')` (back to the default)."""
    NAMES = {'TSF_HARM': 'harm', 'TSF_LATTICE': 'lattice', 'TSF_SPARSE_EXTRA': 'sparse_extra', 'TSF_FIT_GROUPED': 'fit_grouped',
             'TSF_GRAM_SHARE': 'gram_share', 'TSF_GRID_ORDER': 'grid_order', 'TSF_GRID_SHARE': 'grid_share',
             'TSF_RAGGED_SPLIT': 'ragged_split', 'TSF_QUAD_REG': 'quad_reg', 'TSF_QUAD_M2_LDS': 'quad_m2_lds',
             'TSF_QUAD_W4': 'quad_w4', 'TSF_QUAD_RREG': 'quad_rreg', 'TSF_NEWTON_BATCH': 'newton_batch',
             'TSF_NEWTON_LCAP': 'newton_lcap'}

    @staticmethod
    def _ctx():
        from time_series_spark_amd import forecaster
        return forecaster.get_context()

    def __setitem__(self, k, v):
        self._ctx().set_option(self.NAMES[k], int(v))

    def pop(self, k, default=None):
        self._ctx().set_option(self.NAMES[k], -1)

    __delitem__ = pop

    def update(self, d):
        for k, v in d.items():
            self[k] = v


routes = _Routes()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def n_bit_diff(a, b):
    return int(np.sum(bits(a) != bits(b)))


def uses_quadratic_form(spec):
    """Whether the product evaluates the data term in quadratic (Gram) form for this spec
    (include/tsf.h eval_form; tsf_api.hip run_fit): linear growth, every column additive,
    L-BFGS history 5, eval_form not forced to RESIDUAL (1).  A property of the MODEL only:
    aligned and ragged panels take the same form."""
    modes = [s.get('mode', spec.seasonality_mode) for s in spec.seasonalities]
    modes += [e.get('mode', spec.seasonality_mode) for e in spec.extra]
    return (spec.growth == 'linear' and all(m == 'additive' for m in modes)
            and spec.lbfgs.get('history', 5) == 5 and spec.lbfgs.get('eval_form', 0) != 1)


def oracle_spec(spec):
    """time_series_spark_amd.forecaster.ModelSpec -> oracle.canon_lib spec."""
    from oracle import canon_lib as cl
    opt = {k: v for k, v in spec.lbfgs.items() if k not in ('eval_form', 'algorithm', 'residual_kernel', 'coop_after')}
    opt['eval_mode'] = int(uses_quadratic_form(spec))
    seas = [(s['period'], s['fourier_order'], s.get('mode', spec.seasonality_mode),
             s.get('prior_scale', spec.seasonality_prior_scale)) for s in spec.seasonalities]
    ex = [(e.get('mode', spec.seasonality_mode), e.get('prior_scale', spec.holidays_prior_scale))
          for e in spec.extra]
    return cl.make_spec(growth=spec.growth, n_changepoints=spec.n_changepoints,
                        changepoint_range=spec.changepoint_range,
                        changepoint_prior_scale=spec.changepoint_prior_scale,
                        seasonalities=seas, extra=ex, **opt)


YEARLY = {'name': 'yearly', 'period': 365.25, 'fourier_order': 10}
WEEKLY = {'name': 'weekly', 'period': 7, 'fourier_order': 3}
DAILY = {'name': 'daily', 'period': 1, 'fourier_order': 4}
YEARLY5 = {'name': 'yearly', 'period': 365.25, 'fourier_order': 5}

# name -> (growth, mode, T, seasonalities, n_holidays)
CASES = {
    'cfg2_linear_additive': ('linear', 'additive', 730, [YEARLY, WEEKLY], 0),
    'ref_logistic_multiplicative': ('logistic', 'multiplicative', 730, [YEARLY, WEEKLY], 0),
    'linear_multiplicative_365': ('linear', 'multiplicative', 365, [WEEKLY], 0),
    'logistic_additive_400': ('logistic', 'additive', 400, [WEEKLY], 0),
    'short_90': ('linear', 'additive', 90, [WEEKLY], 0),
    'cfg4_holidays': ('logistic', 'multiplicative', 730, [YEARLY, WEEKLY], 10),
    # quadratic-form kernel variants: NT = 18 (cfg3 length), residual staging in global memory
    # (T too long for LDS), the 16-column kernel, and the two-slot (P > 64) kernel
    'cfg3_linear_1095': ('linear', 'additive', 1095, [YEARLY, WEEKLY], 0),
    'long_linear_1400': ('linear', 'additive', 1400, [YEARLY, WEEKLY], 0),
    'kp16_linear_400': ('linear', 'additive', 400, [YEARLY5, WEEKLY], 0),
    'linear_additive_holidays': ('linear', 'additive', 730, [YEARLY, WEEKLY], 10),
}


# cases whose default evaluation form is quadratic also run with the residual form forced
RESID_VARIANTS = ['cfg2_linear_additive@resid', 'short_90@resid']


def make_case(name, N=6, seed=21):
    """Returns (spec, ds, y, floor, cap, extra, fut, extra_future)."""
    from time_series_spark_amd import forecaster as fc, synth
    lb = {}
    if name.endswith('@resid'):
        name, lb = name[:-6], {'eval_form': 1}
    growth, mode, T, seas, nh = CASES[name]
    H = 30
    ds = synth.daily_grid(T)
    fut = ds[-1] + DAY_NS * np.arange(1, H + 1)
    extra = extra_future = None
    extra_spec = []
    hol = None
    if nh:
        allm, names = synth.holiday_matrix(np.concatenate([ds, fut]), nh)
        extra, extra_future = np.ascontiguousarray(allm[:, :T]), np.ascontiguousarray(allm[:, T:])
        extra_spec = [{'name': n} for n in names]
        hol = extra
    ds, y = synth.make_panel(N, T, 'linear' if growth == 'linear' else 'logistic', seed=seed,
                             holidays=hol)
    spec = fc.ModelSpec(growth=growth, seasonality_mode=mode, seasonalities=[dict(s) for s in seas],
                        extra=extra_spec, **lb)
    floor = np.zeros(N)
    cap = y.max(axis=1) * 1.1
    return spec, ds, y, floor, cap, extra, fut, extra_future


def pack_reference(sid, did, ds_ns, y):
    """numpy statement of the packing order contract (include/tsf.h, tsf_pack_rows): NaN-y rows
    dropped, stable lexsort by (series_id, dim_id, ds).  Checker for the native packer."""
    keep = ~np.isnan(y)
    sid, did, ds_ns, y = sid[keep], did[keep], ds_ns[keep], y[keep]
    order = np.lexsort((ds_ns, did, sid))
    sid, did, ds_ns, y = sid[order], did[order], ds_ns[order], y[order]
    new = np.ones(len(sid), dtype=bool)
    new[1:] = (sid[1:] != sid[:-1]) | (did[1:] != did[:-1])
    starts = np.flatnonzero(new)
    offsets = np.concatenate([starts, [len(sid)]]).astype(np.int64)
    return sid[starts], did[starts], offsets, ds_ns, y
