"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C-ABI,
against the canonical CPU oracle on the same seeded inputs and against the committed golden
vectors.  Bar: floating-point forecasts within the north-star 1e-4 relative tolerance
(BASELINE.json); because product and oracle share one canonical arithmetic the tests also
assert the much sharper "identical bits" wherever both are run.  Parity is vs the restated
oracle, NOT real fbprophet (parity unpinned, see oracle/ headers)."""
import os

import numpy as np
import pandas as pd
import pytest

from tests import helpers
from tests.helpers import n_bit_diff

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4          # BASELINE.json north_star


@pytest.fixture(scope='module')
def env(built):
    from time_series_spark_amd import _lib, forecaster as fc
    if _lib.load().tsf_device_count() < 1:
        pytest.fail('no GPU visible: GPU parity tests cannot run (product has no CPU fallback)')
    from oracle import canon_lib as cl
    cl.lib()
    return fc, cl


def test_device_arithmetic_is_ieee_and_detmath_matches(env):
    fc, cl = env
    import ctypes
    L = cl.lib()
    rng = np.random.default_rng(0)
    n = 100000
    a = rng.normal(0, 1, n) * np.exp(rng.uniform(-30, 30, n))
    b = rng.normal(0, 1, n) * np.exp(rng.uniform(-30, 30, n))
    assert n_bit_diff(fc.selftest_math(0, a, b), a / b) == 0
    assert n_bit_diff(fc.selftest_math(1, np.abs(a)), np.sqrt(np.abs(a))) == 0
    x = rng.uniform(-60, 60, 5000)
    assert n_bit_diff(fc.selftest_math(2, x), [L.cn_det_exp(v) for v in x]) == 0
    x = np.exp(rng.uniform(-30, 30, 5000))
    assert n_bit_diff(fc.selftest_math(3, x), [L.cn_det_log(v) for v in x]) == 0
    x = rng.uniform(-5000, 5000, 5000)
    s, c = ctypes.c_double(), ctypes.c_double()
    S, C = [], []
    for v in x:
        L.cn_det_sincos(v, ctypes.byref(s), ctypes.byref(c))
        S.append(s.value)
        C.append(c.value)
    assert n_bit_diff(fc.selftest_math(4, x), S) == 0 and n_bit_diff(fc.selftest_math(5, x), C) == 0


@pytest.mark.parametrize('case', list(helpers.CASES))
def test_design_and_single_evaluation(env, case):
    fc, cl = env
    spec, ds, y, floor, cap, extra, fut, exf = helpers.make_case(case)
    csp = helpers.oracle_spec(spec)
    X, t, grid = fc.design(spec, ds, extra)
    des = cl.design(csp, ds, y[0], floor[0], cap[0], extra)
    assert n_bit_diff(X, des['X']) == 0 and n_bit_diff(t, des['t']) == 0
    assert grid['S'][0] == des['info'].S
    assert n_bit_diff(grid['t_change'][0][:des['info'].S], des['t_change']) == 0
    N = y.shape[0]
    rng = np.random.default_rng(5)
    th = np.zeros((N, spec.theta_stride))
    for n in range(N):
        d = cl.design(csp, ds, y[n], floor[n], cap[n], extra)
        th[n, 0], th[n, 1] = d['k0'], d['m0']
    th += rng.normal(0, 0.01, th.shape)
    f, g = fc.eval_aligned(spec, ds, y, th, floor=floor, cap=cap, extra=extra)
    for n in range(N):
        fo, go, rc = cl.eval_at(csp, ds, y[n], th[n], floor[n], cap[n], extra)
        assert abs(f[n] - fo) <= 1e-12 * abs(fo)
        assert np.max(np.abs(g[n] - go) / (1 + np.abs(go))) <= 1e-11
        assert n_bit_diff(f[n], fo) == 0 and n_bit_diff(g[n], go) == 0


@pytest.mark.parametrize('case', list(helpers.CASES) + helpers.RESID_VARIANTS)
def test_fit_predict_against_oracle_and_golden(env, case):
    """case 'name@resid' = the same case with eval_form forced to RESIDUAL (cases whose default
    is the quadratic form: linear growth + additive columns)."""
    fc, cl = env
    spec, ds, y, floor, cap, extra, fut, exf = helpers.make_case(case)
    csp = helpers.oracle_spec(spec)
    g = np.load(helpers.GOLDEN + '/synthetic_cases.npz')
    r = fc.fit_aligned(spec, ds, y, floor=floor, cap=cap, extra=extra)
    yhat, yint = fc.predict(spec, r.theta, r.y_scale, r.grid, fut, floor=floor, cap=cap,
                            extra_future=exf, want_int=True)
    # committed golden vectors (oracle outputs generated in the build container)
    assert np.array_equal(r.n_iter, g[case + '/n_iter']) and np.array_equal(r.status, g[case + '/status'])
    assert np.max(np.abs(yhat - g[case + '/yhat']) / np.abs(g[case + '/yhat'])) <= REL_TOL
    assert np.array_equal(yhat, g[case + '/yhat'])
    assert np.array_equal(r.theta, g[case + '/theta'])
    # live oracle on the same inputs
    for n in range(y.shape[0]):
        o = cl.fit(csp, ds, y[n], floor[n], cap[n], extra)
        yo, _ = cl.predict(csp, o, fut, floor[n], cap[n], exf)
        assert r.n_eval[n] == o['n_eval'] and n_bit_diff(r.fval[n], o['f']) == 0
        assert np.max(np.abs(yhat[n] - yo) / np.abs(yo)) <= REL_TOL
    # the reference's post-step: int truncation, clamp to floor (prophet_scorer.py:73-84)
    assert np.array_equal(yint, np.maximum(np.trunc(yhat), floor[:, None]).astype(np.int32))


@pytest.mark.parametrize('case', ['ref_logistic_multiplicative', 'cfg2_linear_additive@resid',
                                  'linear_multiplicative_365', 'logistic_additive_400', 'short_90@resid'])
def test_matrix_core_kernel_is_bit_identical_to_the_one_wave_kernel(env, case):
    """Residual-form fits of an aligned panel run 16 series per workgroup on the matrix cores
    (csrc/tsf_mfma_kernels.h: v_mfma_f64_16x16x4_f64 chains in the canonical order); the one-wave
    kernel (residual_kernel = WAVE) and the oracle must give the same bits.  N = 37: two full tiles,
    one partial, slots refilled from the queue."""
    fc, cl = env
    from time_series_spark_amd import _lib
    spec, ds, y, floor, cap, extra, fut, exf = helpers.make_case(case, N=37)
    kw = dict(growth=spec.growth, seasonality_mode=spec.seasonality_mode, seasonalities=spec.seasonalities,
              **spec.lbfgs)
    r_m = fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_MFMA, **kw), ds, y, floor=floor, cap=cap)
    r_w = fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_WAVE, **kw), ds, y, floor=floor, cap=cap)
    assert np.array_equal(r_m.status, r_w.status) and (r_m.status > 0).all()
    assert np.array_equal(r_m.n_iter, r_w.n_iter) and np.array_equal(r_m.n_eval, r_w.n_eval)
    assert np.array_equal(r_m.theta, r_w.theta) and np.array_equal(r_m.fval, r_w.fval)
    assert np.array_equal(r_m.y_scale, r_w.y_scale)
    csp = helpers.oracle_spec(spec)
    for n in (0, 17, 36):
        o = cl.fit(csp, ds, y[n], floor[n], cap[n])
        S = o['info'].S
        assert (r_m.n_iter[n], r_m.n_eval[n], r_m.status[n]) == (o['n_iter'], o['n_eval'], o['status'])
        assert n_bit_diff(r_m.theta[n][:3 + S], o['theta'][:3 + S]) == 0 and n_bit_diff(r_m.fval[n], o['f']) == 0
    # a single series (what a per-group UDF call hands over) and an odd truncation
    r1 = fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_MFMA, **kw), ds, y[5:6], floor=floor[5:6], cap=cap[5:6])
    assert np.array_equal(r1.theta[0], r_w.theta[5]) and r1.n_eval[0] == r_w.n_eval[5]
    kw7 = dict(kw, max_iter=7)
    r7m = fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_MFMA, **kw7), ds, y[:20], floor=floor[:20], cap=cap[:20])
    r7w = fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_WAVE, **kw7), ds, y[:20], floor=floor[:20], cap=cap[:20])
    assert np.array_equal(r7m.theta, r7w.theta) and np.array_equal(r7m.n_eval, r7w.n_eval)


@pytest.mark.parametrize('case', ['ref_logistic_multiplicative', 'cfg2_linear_additive@resid',
                                  'linear_multiplicative_365', 'logistic_additive_400', 'short_90@resid',
                                  'cfg4_holidays'])
def test_cooperative_tail_is_bit_identical_to_the_one_wave_kernel(env, case):
    """A residual-form fit can be suspended at any line-search evaluation and finished by a whole
    workgroup (csrc/tsf_coop_kernels.h: 16 waves share every evaluation in eval_fg's order).  The
    one-wave kernel alone (residual_kernel = WAVE), every series handed over after its first
    evaluation (COOP), hand-overs after 2 / 7 / 40 / 300 evaluations (coop_after) and the default
    rule (AUTO: whatever still runs when the launch has started its last series) must give the same
    bits -- and those of the oracle."""
    fc, cl = env
    from time_series_spark_amd import _lib
    spec, ds, y, floor, cap, extra, fut, exf = helpers.make_case(case, N=21)
    kw = dict(growth=spec.growth, seasonality_mode=spec.seasonality_mode, seasonalities=spec.seasonalities,
              extra=spec.extra, **spec.lbfgs)
    fit = lambda **o: fc.fit_aligned(fc.ModelSpec(**dict(kw, **o)), ds, y, floor=floor, cap=cap, extra=extra)  # noqa: E731
    r_w = fit(residual_kernel=_lib.RK_WAVE)
    assert (r_w.status > 0).all()
    variants = [dict(residual_kernel=_lib.RK_COOP), dict(residual_kernel=_lib.RK_AUTO)]
    variants += [dict(coop_after=k) for k in (2, 7, 40, 300)]
    for o in variants:
        r = fit(**o)
        assert np.array_equal(r.status, r_w.status) and np.array_equal(r.n_iter, r_w.n_iter), o
        assert np.array_equal(r.n_eval, r_w.n_eval) and np.array_equal(r.fval, r_w.fval), o
        assert np.array_equal(r.theta, r_w.theta) and np.array_equal(r.y_scale, r_w.y_scale), o
    csp = helpers.oracle_spec(spec)
    r_c = fit(residual_kernel=_lib.RK_COOP)
    for n in (0, 9, 20):
        o = cl.fit(csp, ds, y[n], floor[n], cap[n], extra)
        S = o['info'].S
        assert (r_c.n_iter[n], r_c.n_eval[n], r_c.status[n]) == (o['n_iter'], o['n_eval'], o['status'])
        assert n_bit_diff(r_c.theta[n][:3 + S], o['theta'][:3 + S]) == 0 and n_bit_diff(r_c.fval[n], o['f']) == 0
    # truncated runs: the iteration cap is hit inside the cooperative kernel
    r7w, r7c = fit(residual_kernel=_lib.RK_WAVE, max_iter=7), fit(residual_kernel=_lib.RK_COOP, max_iter=7)
    assert np.array_equal(r7w.theta, r7c.theta) and np.array_equal(r7w.n_eval, r7c.n_eval)
    assert np.array_equal(r7w.status, r7c.status)


def test_cooperative_tail_other_shapes(env):
    """The cooperative kernel beyond the standard cases: ragged panels on a timestamp lattice (shared
    design table) and with private grids, mixed additive / multiplicative columns (two parameters per
    lane), no changepoints (the dummy changepoint), T = 1 400 (22 steps per chunk: two steps per wave),
    a single series, more series than checkpoint waves can hold at once -- against the one-wave kernel."""
    fc, cl = env
    from time_series_spark_amd import _lib, synth
    rng = np.random.default_rng(11)

    def same(a, b, ctx):
        assert np.array_equal(a.status, b.status) and np.array_equal(a.n_iter, b.n_iter), ctx
        assert np.array_equal(a.n_eval, b.n_eval) and np.array_equal(a.fval, b.fval), ctx
        assert np.array_equal(a.theta, b.theta), ctx

    # aligned shapes
    shapes = [
        dict(growth='logistic', seasonality_mode='multiplicative', T=1400, seas=[helpers.YEARLY, helpers.WEEKLY], n_cp=25),
        dict(growth='linear', seasonality_mode='multiplicative', T=200, seas=[helpers.WEEKLY], n_cp=0),
        dict(growth='logistic', seasonality_mode='additive', T=64, seas=[helpers.WEEKLY], n_cp=10),
        dict(growth='logistic', seasonality_mode='multiplicative', T=365, n_cp=25,       # mixed modes, P = 68
             seas=[dict(helpers.WEEKLY, mode='additive'), {'name': 'yearly', 'period': 365.25, 'fourier_order': 12},
                   {'name': 'monthly', 'period': 30.5, 'fourier_order': 5, 'mode': 'additive'}]),
        dict(growth='linear', seasonality_mode='additive', T=3000, seas=[helpers.YEARLY5, helpers.WEEKLY], n_cp=25,
             eval_form=_lib.EVAL_RESIDUAL, max_iter=60),
    ]
    for sh in shapes:
        sh = dict(sh)
        T, seas, n_cp = sh.pop('T'), sh.pop('seas'), sh.pop('n_cp')
        N = 5
        ds, y = synth.make_panel(N, T, sh['growth'], seed=300 + T)
        fit_kw = dict(floor=np.zeros(N), cap=y.max(axis=1) * 1.2) if sh['growth'] == 'logistic' else {}
        kw = dict(sh, seasonalities=seas, n_changepoints=n_cp)
        r_w = fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_WAVE, **kw), ds, y, **fit_kw)
        for o in (dict(residual_kernel=_lib.RK_COOP), dict(coop_after=int(rng.integers(1, 60)))):
            same(fc.fit_aligned(fc.ModelSpec(**dict(kw, **o)), ds, y, **fit_kw), r_w, (T, o))
        same(fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_COOP, **kw), ds, y[2:3],
                            **{k: v[2:3] for k, v in fit_kw.items()}),
             fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_WAVE, **kw), ds, y[2:3],
                            **{k: v[2:3] for k, v in fit_kw.items()}), (T, 'single'))
    # ragged: one sampling lattice with different starts / lengths (shared design table), and
    # private timestamps (rows dropped at random: no common lattice step beyond the day ... still a
    # lattice; irregular seconds break it)
    spec_kw = dict(growth='logistic', seasonality_mode='multiplicative', seasonalities=[helpers.YEARLY, helpers.WEEKLY])
    N, T = 9, 730
    ds, y = synth.make_panel(N, T, 'logistic', seed=77)
    for jitter in (False, True):
        keep = [np.sort(rng.choice(T, size=T - int(rng.integers(0, 90)), replace=False)) for _ in range(N)]
        offs = np.concatenate([[0], np.cumsum([len(k) for k in keep])]).astype(np.int64)
        dsr = np.concatenate([ds[k] for k in keep])
        if jitter:
            dsr = dsr + rng.integers(0, 3600, size=dsr.shape) * 1000000007     # irregular: no lattice
            dsr = np.concatenate([np.sort(dsr[offs[i]:offs[i + 1]]) for i in range(N)])
        yr = np.concatenate([y[n][k] for n, k in enumerate(keep)])
        fit_kw = dict(floor=np.zeros(N), cap=y.max(axis=1) * 1.1)
        r_w = fc.fit_ragged(fc.ModelSpec(residual_kernel=_lib.RK_WAVE, **spec_kw), offs, dsr, yr, **fit_kw)
        for o in (dict(residual_kernel=_lib.RK_COOP), dict(coop_after=25)):
            same(fc.fit_ragged(fc.ModelSpec(**dict(spec_kw, **o)), offs, dsr, yr, **fit_kw), r_w, ('ragged', jitter, o))
    # direct mode (every series on a workgroup from its initial values) with series that never reach the
    # optimiser: constant y (fbprophet skips the fit), cap <= floor (fbprophet raises), too few rows
    ds, y = synth.make_panel(5, 200, 'linear', seed=9)
    y[1] = 7.0
    lin = dict(growth='linear', seasonality_mode='multiplicative', seasonalities=[helpers.WEEKLY])
    same(fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_COOP, **lin), ds, y),
         fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_WAVE, **lin), ds, y), 'constant')
    assert fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_COOP, **lin), ds, y).status[1] == _lib.ST_CONSTANT
    ds, y = synth.make_panel(4, 200, 'logistic', seed=10)
    capb = y.max(axis=1) * 1.1
    capb[2] = -1.0
    lg = dict(growth='logistic', seasonality_mode='multiplicative', seasonalities=[helpers.WEEKLY])
    r_c = fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_COOP, **lg), ds, y, floor=np.zeros(4), cap=capb)
    same(r_c, fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_WAVE, **lg), ds, y, floor=np.zeros(4), cap=capb), 'cap')
    assert r_c.status[2] == _lib.ST_CAP
    offs = np.array([0, 200, 201, 401], dtype=np.int64)             # a one-row series in a ragged call
    dsr = np.concatenate([ds, ds[:1], ds])
    yr = np.concatenate([y[0], y[1][:1], y[3]])
    fit_kw = dict(floor=np.zeros(3), cap=np.array([capb[0], capb[1], capb[3]]))
    r_c = fc.fit_ragged(fc.ModelSpec(residual_kernel=_lib.RK_COOP, **lg), offs, dsr, yr, **fit_kw)
    same(r_c, fc.fit_ragged(fc.ModelSpec(residual_kernel=_lib.RK_WAVE, **lg), offs, dsr, yr, **fit_kw), 'too few')
    assert r_c.status[1] == _lib.ST_TOO_FEW
    # series longer than the cooperative kernel stages in LDS stay on the one-wave kernel (AUTO), and
    # asking for COOP there is an error
    ds, y = synth.make_panel(2, 4200, 'linear', seed=5)
    long_kw = dict(growth='linear', seasonality_mode='multiplicative', seasonalities=[helpers.WEEKLY], max_iter=30)
    same(fc.fit_aligned(fc.ModelSpec(**long_kw), ds, y),
         fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_WAVE, **long_kw), ds, y), 'long')
    with pytest.raises(_lib.TsfError, match='residual_kernel COOP'):
        fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_COOP, **long_kw), ds, y)


def test_matrix_core_kernel_other_shapes(env):
    """fit_mfma_kernel beyond the five standard cases: two row groups per chunk (T = 1 400: 22 rows per
    chunk), 16 design columns, no changepoints (the dummy changepoint), 28 changepoints, a panel of
    3 series, duplicate timestamps around changepoints -- against the one-wave kernel, bit for bit."""
    fc, cl = env
    from time_series_spark_amd import _lib, synth

    def both(kw, ds, y, **fit_kw):
        r_m = fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_MFMA, eval_form=_lib.EVAL_RESIDUAL, **kw), ds, y, **fit_kw)
        r_w = fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_WAVE, eval_form=_lib.EVAL_RESIDUAL, **kw), ds, y, **fit_kw)
        assert np.array_equal(r_m.status, r_w.status) and np.array_equal(r_m.n_eval, r_w.n_eval)
        assert np.array_equal(r_m.theta, r_w.theta) and np.array_equal(r_m.fval, r_w.fval)
        assert np.array_equal(r_m.grid['S'], r_w.grid['S'])
        return r_m

    ds, y = synth.make_panel(19, 1400, 'logistic', seed=31)
    lg = dict(floor=np.zeros(19), cap=y.max(axis=1) * 1.1)
    both(dict(growth='logistic', seasonality_mode='multiplicative',
              seasonalities=[helpers.YEARLY, helpers.WEEKLY]), ds, y, **lg)                  # NG = 2, KP = 28
    both(dict(growth='linear', seasonalities=[helpers.YEARLY5, helpers.WEEKLY]), ds, y)       # KP = 16, NG = 2
    ds4, y4 = synth.make_panel(3, 400, 'logistic', seed=32)
    lg4 = dict(floor=np.zeros(3), cap=y4.max(axis=1) * 1.2)
    both(dict(growth='logistic', seasonalities=[helpers.YEARLY5, helpers.WEEKLY]), ds4, y4, **lg4)   # KP = 16, 3 series
    r0 = both(dict(growth='logistic', seasonality_mode='multiplicative', seasonalities=[helpers.WEEKLY],
                   n_changepoints=0), ds4, y4, **lg4)                                       # dummy changepoint
    assert r0.grid['S'][0] == 0
    both(dict(growth='linear', seasonalities=[helpers.WEEKLY], n_changepoints=28), ds4, y4)   # S = 28 (the kernel's limit)
    both(dict(growth='linear', seasonality_mode='multiplicative', seasonalities=[helpers.WEEKLY], max_iter=25),
         ds4[:100], y4[:, :100])                                                            # 2 rows per chunk
    # duplicate timestamps (the reference's fixture has them): runs of equal ds across changepoint rows
    dsd = np.sort(np.concatenate([ds4, ds4[40:44], ds4[200:203]]))
    yd = np.concatenate([y4, y4[:, 40:44], y4[:, 200:203]], axis=1)
    both(dict(growth='logistic', seasonality_mode='multiplicative', seasonalities=[helpers.WEEKLY]), dsd, yd, **lg4)
    with pytest.raises(_lib.TsfError, match='residual_kernel MFMA'):
        fc.fit_aligned(fc.ModelSpec(residual_kernel=_lib.RK_MFMA, growth='linear', n_changepoints=40,
                                    seasonalities=[helpers.WEEKLY], eval_form=_lib.EVAL_RESIDUAL), ds4, y4)


def test_truncated_trajectories_match(env):
    """Same iterate after 1, 3, 10, 40 L-BFGS iterations: checks line search, two-loop
    recursion and the history ring step by step rather than only at the end."""
    fc, cl = env
    for case in ('cfg2_linear_additive', 'ref_logistic_multiplicative'):
        spec0, ds, y, floor, cap, extra, fut, exf = helpers.make_case(case)
        for mi in (1, 3, 10, 40):
            spec = fc.ModelSpec.from_dict(dict(spec0.to_dict(), lbfgs={'max_iter': mi}))
            csp = helpers.oracle_spec(spec)
            r = fc.fit_aligned(spec, ds, y, floor=floor, cap=cap)
            for n in range(y.shape[0]):
                o = cl.fit(csp, ds, y[n], floor[n], cap[n])
                S = o['info'].S
                assert r.n_iter[n] == o['n_iter'] and r.n_eval[n] == o['n_eval']
                assert n_bit_diff(r.theta[n][:3 + S], o['theta'][:3 + S]) == 0


def test_reference_fixture_ragged_irregular_timestamps(env):
    """The reference's own fixture (two dim_ids, 410/406 rows, Thu-Sun 11:15/21:45 observations,
    four duplicate timestamps) with the reference's settings, through the ragged entry point;
    golden = canonical oracle outputs committed in tests/golden/fixture_751.npz."""
    fc, cl = env
    from time_series_spark_amd import panel as pk
    g = np.load(helpers.GOLDEN + '/fixture_751.npz')
    df = pd.DataFrame({'series_id': 751, 'dim_id': g['raw_dim_id'],
                       'ds': g['raw_ds_ns'].astype('datetime64[ns]'), 'y': g['raw_y']})
    p = pk.pack_long_frame(df)
    span, min_dt, ymax = pk.per_series_stats(p)
    seas = fc.ModelSpec.auto_from_stats(int(span[0]), int(min_dt[0]), seasonality_mode='multiplicative')
    spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=seas)
    cap = ymax * 1.1
    r = fc.fit_ragged(spec, p.offsets, p.ds_ns, p.y, floor=np.zeros(2), cap=cap)
    assert np.array_equal(r.n_iter, g['n_iter']) and np.array_equal(r.status, g['status'])
    assert np.array_equal(r.theta, g['theta'])
    yhat = fc.predict(spec, r.theta, r.y_scale, r.grid, g['fut'], floor=np.zeros(2, dtype=np.float32).astype(np.float64),
                      cap=cap.astype(np.float32).astype(np.float64))
    assert np.max(np.abs(yhat - g['yhat']) / np.abs(g['yhat'])) <= REL_TOL
    assert np.array_equal(yhat, g['yhat'])


def test_edge_cases(env):
    fc, cl = env
    from time_series_spark_amd import _lib
    spec, ds, y, floor, cap, extra, fut, exf = helpers.make_case('cfg2_linear_additive')
    csp = helpers.oracle_spec(spec)
    yy = y.copy()
    yy[0] = 7.0                                   # constant series, linear growth
    r = fc.fit_aligned(spec, ds, yy)
    assert r.status[0] == _lib.ST_CONSTANT and r.n_iter[0] == 0
    o = cl.fit(csp, ds, yy[0])
    assert np.array_equal(r.theta[0], np.concatenate([o['theta'][:3 + 25], o['theta'][28:]]))
    yh = fc.predict(spec, r.theta, r.y_scale, r.grid, fut)
    assert np.allclose(yh[0], 7.0)
    # int32 / float32 inputs give the same fit as float64 (values are integers)
    r64 = fc.fit_aligned(spec, ds, y)
    r32 = fc.fit_aligned(spec, ds, y.astype(np.int32))
    rf32 = fc.fit_aligned(spec, ds, y.astype(np.float32))
    assert np.array_equal(r64.theta, r32.theta) and np.array_equal(r64.theta, rf32.theta)
    # cap <= floor is reported per series (fbprophet raises ValueError)
    lspec = helpers.make_case('ref_logistic_multiplicative')[0]
    capbad = cap.copy()
    capbad[2] = -1.0
    rl = fc.fit_aligned(lspec, ds, y, floor=floor, cap=capbad)
    assert rl.status[2] == _lib.ST_CAP and (rl.status[[0, 1, 3]] > 0).all()
    # ragged: one-row series -> TOO_FEW; short series -> fewer changepoints; duplicates fine
    off = np.array([0, 1, 21, 21 + 90, 21 + 90 + 730], dtype=np.int64)
    dsr = np.concatenate([ds[:1], ds[:20], np.sort(np.concatenate([ds[:88], ds[10:12]])), ds])
    yr = np.concatenate([y[0][:1], y[0][:20], y[1][:90], y[2]])
    sp90 = helpers.make_case('short_90')[0]
    rr = fc.fit_ragged(sp90, off, dsr, yr)
    assert rr.status[0] == _lib.ST_TOO_FEW and (rr.status[1:] > 0).all()
    assert list(rr.grid['S']) == [0, 15, 25, 25]
    c90 = helpers.oracle_spec(sp90)      # linear + additive: quadratic form, ragged or not
    for n in (1, 2, 3):
        o = cl.fit(c90, dsr[off[n]:off[n + 1]], yr[off[n]:off[n + 1]])
        S = o['info'].S
        assert rr.n_iter[n] == o['n_iter']
        assert n_bit_diff(rr.theta[n][:3 + S], o['theta'][:3 + S]) == 0
        assert n_bit_diff(rr.theta[n][28:], o['theta'][3 + S:]) == 0


def test_ragged_and_aligned_entry_points_agree_bit_for_bit(env):
    """The evaluation form depends on the model only: the same series fitted through the aligned
    entry point, or as members of a ragged panel next to unrelated series of other lengths, gives
    identical bits (quadratic form: per-series Z^T Z built in-kernel vs one shared Z^T Z; residual
    form likewise)."""
    fc, cl = env
    for case in ('cfg2_linear_additive', 'kp16_linear_400', 'ref_logistic_multiplicative',
                 'linear_additive_holidays'):
        spec, ds, y, floor, cap, extra, fut, exf = helpers.make_case(case, N=5)
        ra = fc.fit_aligned(spec, ds, y, floor=floor, cap=cap, extra=extra)
        T = len(ds)
        cut = [T, T - 37, T, T - 101, T]     # ragged: two series are truncated copies
        off = np.concatenate([[0], np.cumsum(cut)]).astype(np.int64)
        dsr = np.concatenate([ds[:c] for c in cut])
        yr = np.concatenate([y[i][:c] for i, c in enumerate(cut)])
        exr = None if extra is None else np.concatenate([extra[:, :c] for c in cut], axis=1)
        rr = fc.fit_ragged(spec, off, dsr, yr, floor=floor, cap=cap, extra=exr)
        for i in (0, 2, 4):                  # the full-length members
            assert np.array_equal(rr.theta[i], ra.theta[i]) and rr.n_eval[i] == ra.n_eval[i], (case, i)
            assert rr.status[i] == ra.status[i] and rr.fval[i] == ra.fval[i]
        csp = helpers.oracle_spec(spec)
        for i in (1, 3):                     # the truncated ones against the oracle
            o = cl.fit(csp, ds[:cut[i]], y[i][:cut[i]], floor[i], cap[i],
                       None if extra is None else extra[:, :cut[i]])
            S = o['info'].S
            assert rr.n_iter[i] == o['n_iter'] and rr.n_eval[i] == o['n_eval']
            assert n_bit_diff(rr.theta[i][:3 + S], o['theta'][:3 + S]) == 0


def test_ragged_call_is_cut_into_length_classes(env):
    """tsf_fit_ragged pads every series' device tables to the longest series of the call; since round 4 the host
    entry point cuts a call whose padding exceeds its rows into length classes (tsf_api.hip fit_host) -- workspace
    proportional to the rows that exist.  (a) a small mixed panel (lengths 90 ... 8 000, three models incl. explicit
    columns) split (TSF_RAGGED_SPLIT=1) and unsplit (=0): identical bits for every series, every output; (b) the
    round-2 advisor's case at full size -- ONE 100 000-row series among 10 000 series of ~700 rows, which as one
    padded layout would ask for ~224 GB -- fits, the long series and a sample of the short ones match the oracle
    bit for bit."""
    import os
    fc, cl = env
    from time_series_spark_amd import synth
    rng = np.random.default_rng(41)
    lens = np.concatenate([rng.integers(90, 130, 40), rng.integers(600, 760, 60), [8000, 3000, 2999, 1400, 65, 64, 2]])
    rng.shuffle(lens)
    N = len(lens)
    Tm = int(lens.max())
    for growth, mode, nh in (('linear', 'additive', 0), ('logistic', 'multiplicative', 0), ('linear', 'additive', 2)):
        dsm = synth.daily_grid(Tm)
        extra_full = None
        ex_spec = []
        if nh:
            extra_full, names = synth.holiday_matrix(dsm, nh)
            ex_spec = [{'name': n} for n in names]
        _, ym = synth.make_panel(N, Tm, growth, seed=5 + nh, holidays=extra_full)
        spec = fc.ModelSpec(growth=growth, seasonality_mode=mode, seasonalities=[helpers.WEEKLY], extra=ex_spec)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        dsr = np.concatenate([dsm[:c] for c in lens])
        yr = np.concatenate([ym[i][:c] for i, c in enumerate(lens)])
        exr = None if extra_full is None else np.concatenate([extra_full[:, :c] for c in lens], axis=1)
        cap = np.array([ym[i][:c].max() * 1.1 for i, c in enumerate(lens)])
        res = {}
        for mode_ in ('0', '1'):
            helpers.routes['TSF_RAGGED_SPLIT'] = mode_
            try:
                res[mode_] = fc.fit_ragged(spec, off, dsr, yr, floor=np.zeros(N), cap=cap, extra=exr)
            finally:
                helpers.routes.pop('TSF_RAGGED_SPLIT', None)
        for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status', 'y_scale'):
            assert np.array_equal(getattr(res['0'], name), getattr(res['1'], name), equal_nan=True), (growth, nh, name)
        assert res['0'].grid.tobytes() == res['1'].grid.tobytes()
    # (b)
    N, Tl = 10000, 100000
    lens = rng.integers(640, 731, N + 1)
    lens[4321] = Tl
    dsl = synth.daily_grid(Tl)
    _, ys = synth.make_panel(N + 1, 730, 'linear', seed=77)
    _, yl = synth.make_panel(1, Tl, 'linear', seed=78)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    dsr = np.concatenate([dsl[:c] for c in lens])
    yr = np.concatenate([(yl[0] if i == 4321 else ys[i][:c]) for i, c in enumerate(lens)])
    spec = fc.ModelSpec(growth='linear', seasonalities=[helpers.YEARLY, helpers.WEEKLY])
    r = fc.fit_ragged(spec, off, dsr, yr)
    assert (r.n_iter >= 1).all() and (r.status <= 0).sum() <= 2
    csp = helpers.oracle_spec(spec)
    for n in (0, 4320, 4321, 4322, N):
        o = cl.fit(csp, dsr[off[n]:off[n + 1]], yr[off[n]:off[n + 1]])
        assert (r.n_iter[n], r.n_eval[n], r.status[n]) == (o['n_iter'], o['n_eval'], o['status']), n
        assert n_bit_diff(r.theta[n], o['theta']) == 0 and n_bit_diff(r.fval[n], o['f']) == 0, n


def test_ragged_series_with_equal_timestamps_share_grid_tables(env):
    """Series of a ragged call whose timestamp vectors are byte-identical share one set of grid tables (t, changepoint
    counts, design matrix) and, on the quadratic form, one prebuilt Z^T Z (tsf_api.hip fit_host_one: hash + memcmp
    classes; gram_grids_kernel).  Sharing changes where a table lives, not a bit of any result: default against
    TSF_GRID_SHARE=0 (a grid per series) against TSF_GRAM_SHARE=0 (shared tables, every wave builds its own Z^T Z: three
    columns per pass with the Fourier columns expanded from the rows' base pairs, gram_columns_harm -- or, with
    TSF_HARM=0 on top, two columns per pass from the design tables, gram_columns2), three models and Newton; a series with the same LENGTH but timestamps one day later is its own class; a sample
    against the oracle."""
    import os
    fc, cl = env
    from time_series_spark_amd import _lib, synth
    rng = np.random.default_rng(43)
    N = 90
    kinds = rng.integers(0, 4, N)                       # four calendars ...
    kinds[[7, 50]] = 4                                  # ... and two loners: length of calendar 0, shifted by a day
    Tm = 760
    dsm = synth.daily_grid(Tm + 1)
    cal = {0: dsm[:730], 1: dsm[:365], 2: dsm[20:750], 3: dsm[:400][::2], 4: dsm[1:731]}
    lens = np.array([len(cal[k]) for k in kinds])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    dsr = np.concatenate([cal[k] for k in kinds])
    for growth, mode, algo in (('linear', 'additive', None), ('logistic', 'multiplicative', None),
                               ('linear', 'multiplicative', None), ('linear', 'additive', _lib.ALGO_NEWTON)):
        _, ym = synth.make_panel(N, Tm, growth, seed=9)
        yr = np.concatenate([ym[i][:c] for i, c in enumerate(lens)])
        cap = np.array([ym[i][:c].max() * 1.1 for i, c in enumerate(lens)])
        spec = fc.ModelSpec(growth=growth, seasonality_mode=mode, seasonalities=[helpers.YEARLY, helpers.WEEKLY])
        if algo is not None:
            spec = type(spec).from_dict(dict(spec.to_dict(), lbfgs=dict(spec.lbfgs, algorithm=algo)))
        res = {}
        for tag, envs in (('shared', {}), ('own_grids', {'TSF_GRID_SHARE': '0'}), ('own_gram', {'TSF_GRAM_SHARE': '0'}),
                          ('own_gram_from_tables', {'TSF_GRAM_SHARE': '0', 'TSF_HARM': '0'})):
            helpers.routes.update(envs)
            try:
                res[tag] = fc.fit_ragged(spec, off, dsr, yr, floor=np.zeros(N), cap=cap)
            finally:
                for k in envs:
                    helpers.routes.pop(k, None)
        for tag in ('own_grids', 'own_gram', 'own_gram_from_tables'):
            for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status', 'y_scale'):
                assert np.array_equal(getattr(res['shared'], name), getattr(res[tag], name), equal_nan=True), (growth, mode, algo, tag, name)
            assert res['shared'].grid.tobytes() == res[tag].grid.tobytes()
        r = res['shared']
        assert (r.status > 0).sum() >= N - 10
        if algo is None:
            csp = helpers.oracle_spec(spec)
            for n in (0, 7, 8, 50, N - 1):
                o = cl.fit(csp, dsr[off[n]:off[n + 1]], yr[off[n]:off[n + 1]], 0.0, cap[n])
                assert (r.n_iter[n], r.n_eval[n], r.status[n]) == (o['n_iter'], o['n_eval'], o['status']), (growth, mode, n)
                P = len(o['theta'])
                assert n_bit_diff(r.theta[n][:P], o['theta']) == 0 and n_bit_diff(r.fval[n], o['f']) == 0, (growth, mode, n)


def test_ragged_quadratic_form_with_a_calendar_per_series_reads_base_pairs(env):
    """Round 5: a ragged quadratic-form panel whose series have a calendar each builds Z^T Z per series inside the fit
    kernel; the rows are then read as their BASE sin / cos pairs (32 B instead of 224 B per row) and the Fourier columns
    expanded in registers, three columns of Z^T Z per pass (gram_columns_harm) and in the residual passes at the
    re-centrings (resid_eval_q HARM) -- 2.4 instead of 32 GB of reads per 10 000-series launch.  Same operands in the
    same order: with 0, 1 and 2 dense regressor columns behind the Fourier block (still read from the tables) the
    results equal, bit for bit, those of the table route (context option harm = 0) and the oracle's."""
    fc, cl = env
    from time_series_spark_amd import synth
    rng = np.random.default_rng(77)
    N, Tm = 24, 800
    dsm = synth.daily_grid(Tm)
    _, ym = synth.make_panel(N, Tm, 'linear', seed=31)
    keep = [np.sort(rng.choice(Tm, size=int(rng.integers(600, 731)), replace=False)) for _ in range(N)]
    lens = np.array([len(k) for k in keep])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    dsr = np.concatenate([dsm[k] for k in keep])
    yr = np.concatenate([ym[i][k] for i, k in enumerate(keep)])
    for n_extra in (0, 1, 2):
        exr = rng.normal(0, 1, (n_extra, len(dsr))) if n_extra else None
        spec = fc.ModelSpec(growth='linear', seasonalities=[helpers.YEARLY, helpers.WEEKLY],
                            extra=[{'name': 'x%d' % e} for e in range(n_extra)])
        assert helpers.uses_quadratic_form(spec)
        r = fc.fit_ragged(spec, off, dsr, yr, extra=exr)
        with fc.get_context().options(harm=0):
            r0 = fc.fit_ragged(spec, off, dsr, yr, extra=exr)
        for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status', 'y_scale'):
            assert np.array_equal(getattr(r, name), getattr(r0, name), equal_nan=True), (n_extra, name)
        assert (r.status > 0).all()
        csp = helpers.oracle_spec(spec)
        for n in (0, 11, N - 1):
            sl = slice(off[n], off[n + 1])
            o = cl.fit(csp, dsr[sl], yr[sl], 0.0, 0.0, None if exr is None else exr[:, sl])
            assert (r.n_iter[n], r.n_eval[n], r.status[n]) == (o['n_iter'], o['n_eval'], o['status']), (n_extra, n)
            P = len(o['theta'])
            assert n_bit_diff(r.theta[n][:P], o['theta']) == 0 and n_bit_diff(r.fval[n], o['f']) == 0, (n_extra, n)


def test_base_pair_kernel_with_row_prefetch_changes_no_bit(env):
    """Round 5: where a residual-form panel's rows come from HBM (a table per series: irregular timestamps) the launcher
    takes the variant of the base-pair kernel that requests a row one step ahead of its use, compiled for two waves per
    SIMD (fit_kernel<..., HARM, PF>; context option harm = 2 forces it, 1 forbids it).  Same operations on the same
    values: the reference's model on series with calendars of their own gives identical bits with and without the
    prefetch and from the tables (harm = 0), and matches the oracle."""
    fc, cl = env
    from time_series_spark_amd import synth
    rng = np.random.default_rng(78)
    N, Tm = 40, 800
    dsm = synth.daily_grid(Tm)
    _, ym = synth.make_panel(N, Tm, 'logistic', seed=32)
    keep = [np.sort(rng.choice(Tm, size=int(rng.integers(600, 731)), replace=False)) for _ in range(N)]
    lens = np.array([len(k) for k in keep])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    dsr = np.concatenate([dsm[k] + int(rng.integers(0, 3600)) * 1000000000 for k in keep])      # (off the daily lattice)
    yr = np.concatenate([ym[i][k] for i, k in enumerate(keep)])
    cap = np.array([ym[i][k].max() * 1.1 for i, k in enumerate(keep)])
    spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=[helpers.YEARLY, helpers.WEEKLY])
    res = {}
    for tag, h in (('prefetch', 2), ('plain', 1), ('tables', 0)):
        with fc.get_context().options(harm=h):
            res[tag] = fc.fit_ragged(spec, off, dsr, yr, floor=np.zeros(N), cap=cap)
    for tag in ('plain', 'tables'):
        for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status', 'y_scale'):
            assert np.array_equal(getattr(res['prefetch'], name), getattr(res[tag], name), equal_nan=True), (tag, name)
    r = res['prefetch']
    assert (r.status > 0).sum() >= N - 2
    csp = helpers.oracle_spec(spec)
    for n in (0, 13, N - 1):
        sl = slice(off[n], off[n + 1])
        o = cl.fit(csp, dsr[sl], yr[sl], 0.0, cap[n])
        assert (r.n_iter[n], r.n_eval[n], r.status[n]) == (o['n_iter'], o['n_eval'], o['status']), n
        assert n_bit_diff(r.theta[n][:len(o['theta'])], o['theta']) == 0 and n_bit_diff(r.fval[n], o['f']) == 0, n


def test_lattice_panels_read_the_base_pairs_of_their_lattice_points(env):
    """Round 6: series observed at their OWN subsets of the slots of one time lattice (the reference's fixture: Thu-Sun at
    11:15 and 21:45) keep rows of 22 bytes -- t, y, segment word, lattice point -- and every model with a compiled expansion
    takes the points' base pairs from ONE table all series share (fit_kernel<..., XIDX, ..., HARM, PF>, cooperative tail
    likewise) instead of a base-pair table per series (option lattice = 0) or gathered design rows (harm = 0).  Same
    operations on the same values: identical bits on all three routes, whichever kernel finishes a fit, and the oracle's."""
    fc, cl = env
    from time_series_spark_amd import _lib, synth
    rng = np.random.default_rng(81)
    day = synth.DAY_NS
    cases = [
        # (slots of the lattice, seasonalities, growth, mode)
        (synth.daily_grid(800), [helpers.YEARLY, helpers.WEEKLY], 'logistic', 'multiplicative'),
        (synth.daily_grid(800), [helpers.YEARLY, helpers.WEEKLY], 'linear', 'additive'),
        # two slots a day, as the fixture has them: weekly + daily, the model fbprophet picks for it
        (np.sort(np.concatenate([synth.daily_grid(400) + (11 * 3600 + 900) * 10 ** 9, synth.daily_grid(400) + (21 * 3600 + 2700) * 10 ** 9])),
         [helpers.WEEKLY, {'name': 'daily', 'period': 1, 'fourier_order': 4}], 'logistic', 'multiplicative'),
        (synth.daily_grid(300), [helpers.WEEKLY], 'linear', 'multiplicative'),
    ]
    for ci, (slots, seas, growth, mode) in enumerate(cases):
        N, Tm = 24, len(slots)
        _, ym = synth.make_panel(N, Tm, growth, seed=40 + ci)
        keep = [np.sort(rng.choice(Tm, size=int(rng.integers(Tm - 200, Tm - 40)), replace=False)) for _ in range(N)]
        off = np.concatenate([[0], np.cumsum([len(k) for k in keep])]).astype(np.int64)
        dsr = np.concatenate([slots[k] for k in keep])
        yr = np.concatenate([ym[i][k] for i, k in enumerate(keep)])
        cap = np.array([ym[i][k].max() * 1.1 for i, k in enumerate(keep)])
        kw = dict(floor=np.zeros(N), cap=cap) if growth == 'logistic' else {}
        base = dict(growth=growth, seasonality_mode=mode, seasonalities=seas, eval_form=_lib.EVAL_RESIDUAL)
        res = {}
        for tag, opts, extra in (('points', dict(), {}), ('tables', dict(lattice=0), {}), ('rows', dict(lattice=1, harm=0), {}),
                                 ('points, tail after 25', dict(), dict(coop_after=25)),
                                 ('points, workgroup kernel', dict(), dict(residual_kernel=_lib.RK_COOP))):
            with fc.get_context().options(**opts):
                res[tag] = fc.fit_ragged(fc.ModelSpec(**dict(base, **extra)), off, dsr, yr, **kw)
        for tag in res:
            for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status', 'y_scale'):
                assert np.array_equal(getattr(res['points'], name), getattr(res[tag], name), equal_nan=True), (ci, tag, name)
        r = res['points']
        assert (r.status > 0).sum() >= N - 2, ci
        if ci in (0, 2):
            # the MAP continuation reads the lattice points' pairs too (map_kernel<..., XIDX, HARM>)
            mp = {}
            for tag, opts in (('points', dict()), ('tables', dict(lattice=0))):
                with fc.get_context().options(**opts):
                    mp[tag] = fc.fit_ragged(fc.ModelSpec(**dict(base, converge=_lib.CONVERGE_MAP)), off, dsr, yr, **kw)
            for name in ('theta', 'fval', 'n_eval', 'status'):
                assert np.array_equal(getattr(mp['points'], name), getattr(mp['tables'], name), equal_nan=True), (ci, 'map', name)
            assert (mp['points'].fval <= r.fval + 1e-9).all()
        csp = helpers.oracle_spec(fc.ModelSpec(**base))
        for n in (0, 11, N - 1):
            sl = slice(off[n], off[n + 1])
            o = cl.fit(csp, dsr[sl], yr[sl], 0.0, cap[n]) if growth == 'logistic' else cl.fit(csp, dsr[sl], yr[sl])
            assert (r.n_iter[n], r.n_eval[n], r.status[n]) == (o['n_iter'], o['n_eval'], o['status']), (ci, n)
            assert n_bit_diff(r.theta[n][:len(o['theta'])], o['theta']) == 0 and n_bit_diff(r.fval[n], o['f']) == 0, (ci, n)


def test_randomised_lattice_panels_equal_a_table_per_series(env):
    """Randomised form of the test above (tools/dev/route_stress.py ran 16 113 such panels for the round's record): random time
    lattices (steps of 24 / 12 / 6 / 1.5 hours, 60-900 slots), every series at a random subset of the slots, five models (three
    with a compiled expansion, two without: those take the gathered design rows), both growths and column modes, three
    changepoint counts and three iteration limits -- the lattice routes against a table per series (option lattice = 0), every
    output bit for bit."""
    fc, cl = env
    from time_series_spark_amd import _lib, synth
    rng = np.random.default_rng(2026)
    Y10, W3 = helpers.YEARLY, helpers.WEEKLY
    D4 = {'name': 'daily', 'period': 1, 'fourier_order': 4}
    M5 = {'name': 'monthly', 'period': 30.5, 'fourier_order': 5}
    for trial in range(40):
        step_h = float(rng.choice([24, 12, 6, 1.5]))
        n_slots = int(rng.integers(60, 900))
        slots = synth.START_NS + (np.arange(n_slots) * step_h * 3600e9).astype(np.int64)
        N = int(rng.integers(1, 30))
        growth = str(rng.choice(['linear', 'logistic']))
        mode = str(rng.choice(['additive', 'multiplicative']))
        seas = [[Y10, W3], [W3, D4], [W3], [W3, M5], [Y10, W3, M5]][int(rng.integers(0, 5))]
        _, ym = synth.make_panel(N, n_slots, growth, seed=int(rng.integers(1, 1 << 30)))
        keep = [np.sort(rng.choice(n_slots, size=int(rng.integers(max(4, n_slots // 2), n_slots + 1)), replace=False)) for _ in range(N)]
        off = np.concatenate([[0], np.cumsum([len(k) for k in keep])]).astype(np.int64)
        dsr = np.concatenate([slots[k] for k in keep])
        yr = np.concatenate([ym[i][k] for i, k in enumerate(keep)])
        kw = dict(floor=np.zeros(N), cap=np.array([ym[i][k].max() * 1.1 for i, k in enumerate(keep)])) if growth == 'logistic' else {}
        spec = fc.ModelSpec(growth=growth, seasonality_mode=mode, seasonalities=seas, n_changepoints=int(rng.choice([0, 5, 25])),
                            eval_form=_lib.EVAL_RESIDUAL, max_iter=int(rng.choice([30, 200, 10000])))
        a = fc.fit_ragged(spec, off, dsr, yr, **kw)
        with fc.get_context().options(lattice=0):
            b = fc.fit_ragged(spec, off, dsr, yr, **kw)
        for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status', 'y_scale'):
            assert np.array_equal(getattr(a, name), getattr(b, name), equal_nan=True), \
                (trial, name, step_h, n_slots, N, growth, mode, [s_['name'] for s_ in seas])


def _used_sparse_columns(fc):
    import ctypes
    from time_series_spark_amd import _lib
    ctx = fc.get_context()
    v = ctypes.c_int32(-1)
    ctx.check(_lib.load().tsf_last_fit_route(ctx.handle, ctypes.byref(v)))
    return v.value == 1


def test_sparse_indicator_columns_equal_dense_columns(env):
    """Models with more than 28 design columns whose columns from the 29th on are 0 / 1 indicators (holidays) run the
    28-column kernel with those columns in sparse form (eval_fg<..., SPARSE>: the ones of a lane's rows as entry words,
    the per-column sums folded in the reduction network's order from a few LDS slots) -- adding a zero is exact, so
    not a bit may differ from the dense 64-column kernel (TSF_SPARSE_EXTRA=0) or from the oracle: BASELINE cfg4's
    model (logistic, multiplicative), its additive twin in residual form, a ragged call, stragglers included; and
    the cases the analysis kernel must REFUSE (a holiday value of 2, thirteen ones inside one lane's rows, an
    indicator column that is mostly ones) fall back to the dense kernel with the same bits as before."""
    import os
    fc, cl = env
    from time_series_spark_amd import synth

    def both(spec, call, want_sparse=True):
        out = {}
        for tag in ('sparse', 'dense'):
            if tag == 'dense':
                helpers.routes['TSF_SPARSE_EXTRA'] = '0'
            try:
                out[tag] = call()
                assert _used_sparse_columns(fc) == (tag == 'sparse' and want_sparse), (tag, want_sparse)
            finally:
                helpers.routes.pop('TSF_SPARSE_EXTRA', None)
        for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status', 'y_scale'):
            assert np.array_equal(getattr(out['sparse'], name), getattr(out['dense'], name), equal_nan=True), name
        return out['sparse']

    for case in ('cfg4_holidays', 'linear_additive_holidays@resid'):
        spec, ds, y, floor, cap, extra, fut, exf = helpers.make_case(case, N=48, seed=5)
        r = both(spec, lambda: fc.fit_aligned(spec, ds, y, floor=floor, cap=cap, extra=extra))
        assert (r.status > 0).sum() >= 40
        csp = helpers.oracle_spec(spec)
        order = np.argsort(r.n_eval)
        for n in list(order[-2:]) + [0, 17]:
            o = cl.fit(csp, ds, y[n], floor[n], cap[n], extra)
            assert (r.n_iter[n], r.n_eval[n], r.status[n]) == (o['n_iter'], o['n_eval'], o['status']), (case, n)
            P = len(o['theta'])
            assert n_bit_diff(r.theta[n][:P], o['theta']) == 0 and n_bit_diff(r.fval[n], o['f']) == 0, (case, n)
        # ragged: truncated histories, explicit columns per row
        T = len(ds)
        cut = np.array([T - 13 * (i % 7) for i in range(len(y))])
        off = np.concatenate([[0], np.cumsum(cut)]).astype(np.int64)
        dsr = np.concatenate([ds[:c] for c in cut])
        yr = np.concatenate([y[i][:c] for i, c in enumerate(cut)])
        exr = np.concatenate([extra[:, :c] for c in cut], axis=1)
        rr = both(spec, lambda: fc.fit_ragged(spec, off, dsr, yr, floor=floor, cap=cap, extra=exr))
        for n in (0, 7):                 # full-length members of the ragged call = the aligned fit
            assert np.array_equal(rr.theta[n], r.theta[n]) and rr.n_eval[n] == r.n_eval[n], (case, n)
    # the TAIL of the sparse kernel (round 5): fits handed over after 64 evaluations run the cooperative kernel on base-pair
    # rows and sparse entries (fit_coop_kernel<28, ..., HARM, SPARSE>) -- against the 64-column streaming tail behind the
    # same sparse fit kernel (option sparse_extra = 2), the dense route and the oracle
    spec, ds, y, floor, cap, extra, fut, exf = helpers.make_case('cfg4_holidays', N=48, seed=5)
    spec_t = type(spec).from_dict(dict(spec.to_dict(), lbfgs=dict(spec.lbfgs, coop_after=64)))
    res = {}
    for tag, opt in (('sparse_tail', -1), ('dense_tail', 2), ('dense', 0)):
        with fc.get_context().options(sparse_extra=opt):
            res[tag] = fc.fit_aligned(spec_t, ds, y, floor=floor, cap=cap, extra=extra)
            assert _used_sparse_columns(fc) == (tag != 'dense'), tag
    for tag in ('dense_tail', 'dense'):
        for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status', 'y_scale'):
            assert np.array_equal(getattr(res['sparse_tail'], name), getattr(res[tag], name), equal_nan=True), (tag, name)
    r = res['sparse_tail']
    assert (r.n_eval > 64).sum() >= 40           # (nearly every fit was finished by the tail)
    csp = helpers.oracle_spec(spec)
    for n in (0, 9, 33, int(np.argmax(r.n_eval))):
        o = cl.fit(csp, ds, y[n], floor[n], cap[n], extra)
        assert (r.n_iter[n], r.n_eval[n], r.status[n]) == (o['n_iter'], o['n_eval'], o['status']), n
        assert n_bit_diff(r.theta[n][:len(o['theta'])], o['theta']) == 0 and n_bit_diff(r.fval[n], o['f']) == 0, n
    # longer series: the dense route is the workgroup kernel from the first evaluation (<= 4 096 rows) or the ungrouped
    # one-wave kernel (longer); the sparse route the same one-wave kernel with the cooperative tail
    for T_long, Nl in ((1095, 20), (4200, 4)):
        dsl = synth.daily_grid(T_long)
        exl, names = synth.holiday_matrix(dsl, 10)
        exl[:, 2600:] = 0.0                  # (at most eight occurrences per column: what the sparse form takes)
        _, yl = synth.make_panel(Nl, T_long, 'logistic', seed=8, holidays=exl)
        specl = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=[helpers.YEARLY, helpers.WEEKLY],
                             extra=[{'name': n} for n in names], max_iter=400)
        capl = yl.max(axis=1) * 1.1
        r = both(specl, lambda: fc.fit_aligned(specl, dsl, yl, floor=np.zeros(Nl), cap=capl, extra=exl))
        cspl = helpers.oracle_spec(specl)
        o = cl.fit(cspl, dsl, yl[1], 0.0, capl[1], exl)
        assert (r.n_iter[1], r.n_eval[1], r.status[1]) == (o['n_iter'], o['n_eval'], o['status']), T_long
        assert n_bit_diff(r.theta[1][:len(o['theta'])], o['theta']) == 0 and n_bit_diff(r.fval[1], o['f']) == 0, T_long
    # refused by the analysis kernel: same results through the dense kernel
    spec, ds, y, floor, cap, extra, fut, exf = helpers.make_case('cfg4_holidays', N=6, seed=6)
    csp = helpers.oracle_spec(spec)
    for what in ('value 2', 'thirteen in one lane', 'dense column'):
        ex = extra.copy()
        if what == 'value 2':
            ex[3, np.flatnonzero(ex[3])[0]] = 2.0
        elif what == 'thirteen in one lane':
            ex[:, :] = 0.0
            for j in range(13):
                ex[2 + j, 24 + min(j, 11)] = 1.0     # rows 24..35 are lane 2's twelve rows (T = 730); two columns share the last
        else:
            ex[5, ::2] = 1.0
        r = both(spec, lambda: fc.fit_aligned(spec, ds, y, floor=floor, cap=cap, extra=ex), want_sparse=False)
        o = cl.fit(csp, ds, y[1], floor[1], cap[1], ex)
        assert (r.n_iter[1], r.n_eval[1], r.status[1]) == (o['n_iter'], o['n_eval'], o['status']), what
        assert n_bit_diff(r.theta[1][:len(o['theta'])], o['theta']) == 0, what


def test_randomised_sparse_indicator_layouts(env):
    """A seeded sweep over indicator layouts for the sparse-column kernel: 200 .. 2 000 rows (4 .. 32 rows per lane),
    seasonal blocks of 6 .. 26 columns (so that some indicator columns fall into the dense first 28), 4 .. 32 indicator
    columns with 1 .. 9 ones each at random rows -- up to eight lanes per column and several ones per lane and per row,
    sometimes more than the kernel takes (the analysis kernel must then refuse and the dense kernel run) --, both
    growths, both column modes: sparse route = dense route (TSF_SPARSE_EXTRA=0) bit for bit, one series per case
    against the oracle."""
    import os
    fc, cl = env
    from time_series_spark_amd import synth
    rng = np.random.default_rng(20260924)
    n_sparse = 0
    for trial in range(28):
        growth = 'logistic' if trial % 2 == 0 else 'linear'
        mode = 'multiplicative' if trial % 3 != 1 else 'additive'
        T = int(rng.choice([200, 365, 730, 768, 769, 1095, 2000]))
        seas = [{'name': 'weekly', 'period': 7, 'fourier_order': 3}]
        if rng.random() < 0.7:
            seas.append({'name': 'yearly', 'period': 365.25, 'fourier_order': int(rng.choice([4, 10]))})
        Kd = sum(2 * s_['fourier_order'] for s_ in seas)
        n_ind = int(rng.integers(max(4, 29 - Kd), 33))
        if Kd + n_ind <= 28:
            n_ind = 29 - Kd + 3
        ex = np.zeros((n_ind, T))
        for j in range(n_ind):
            k = int(rng.integers(1, 10 if trial % 7 == 3 else 6))
            ex[j, rng.choice(T, size=k, replace=False)] = 1.0
        if trial % 5 == 2:                                  # two indicators on the same rows, a window of consecutive days
            ex[1] = ex[0]
            ex[2, 1:] = ex[0, :-1]
        N = 3
        ds = synth.daily_grid(T)
        _, y = synth.make_panel(N, T, growth, seed=100 + trial, holidays=ex if mode == 'multiplicative' else None)
        spec = fc.ModelSpec(growth=growth, seasonality_mode=mode, seasonalities=seas, max_iter=120,
                            extra=[{'name': 'i%02d' % j} for j in range(n_ind)], eval_form=1)
        cap = y.max(axis=1) * 1.1
        out = {}
        for tag in ('sparse', 'dense'):
            if tag == 'dense':
                helpers.routes['TSF_SPARSE_EXTRA'] = '0'
            try:
                out[tag] = fc.fit_aligned(spec, ds, y, floor=np.zeros(N), cap=cap, extra=ex)
                n_sparse += int(_used_sparse_columns(fc))
            finally:
                helpers.routes.pop('TSF_SPARSE_EXTRA', None)
        for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status'):
            assert np.array_equal(getattr(out['sparse'], name), getattr(out['dense'], name), equal_nan=True), (trial, name)
        r = out['sparse']
        o = cl.fit(helpers.oracle_spec(spec), ds, y[0], 0.0, cap[0], ex)
        assert (r.n_iter[0], r.n_eval[0], r.status[0]) == (o['n_iter'], o['n_eval'], o['status']), trial
        assert n_bit_diff(r.theta[0][:len(o['theta'])], o['theta']) == 0 and n_bit_diff(r.fval[0], o['f']) == 0, trial
    assert 12 <= n_sparse <= 27, n_sparse        # most layouts qualify, some are refused (the dense runs never count)


def test_one_host_thread_per_device_gives_identical_bits(env, monkeypatch):
    """SURVEY 8e, in-process arrangement: a call cut into blocks of series, one tsf_ctx + host
    thread per device (here: three contexts on the one GPU the box has), equals the
    single-context call bit for bit -- aligned, ragged and predict."""
    fc, cl = env
    monkeypatch.setattr(fc, 'MIN_SERIES_PER_DEVICE', 4)
    for case in ('cfg2_linear_additive', 'ref_logistic_multiplicative'):
        spec, ds, y, floor, cap, extra, fut, exf = helpers.make_case(case, N=26)
        one = fc.fit_aligned(spec, ds, y, floor=floor, cap=cap, extra=extra)
        many = fc.fit_aligned(spec, ds, y, floor=floor, cap=cap, extra=extra, devices=[0, 0, 0])
        for k in ('theta', 'y_scale', 'fval', 'status', 'n_iter', 'n_eval'):
            assert np.array_equal(getattr(one, k), getattr(many, k)), (case, k)
        assert len(many.grid) == 1 and many.grid.tobytes() == one.grid.tobytes()
        T = len(ds)
        cut = np.array([T - 7 * (i % 5) for i in range(len(y))])
        off = np.concatenate([[0], np.cumsum(cut)]).astype(np.int64)
        dsr = np.concatenate([ds[:c] for c in cut])
        yr = np.concatenate([y[i][:c] for i, c in enumerate(cut)])
        r1 = fc.fit_ragged(spec, off, dsr, yr, floor=floor, cap=cap)
        r3 = fc.fit_ragged(spec, off, dsr, yr, floor=floor, cap=cap, devices='0,0,0')
        for k in ('theta', 'y_scale', 'fval', 'status', 'n_iter', 'n_eval'):
            assert np.array_equal(getattr(r1, k), getattr(r3, k)), (case, k)
        assert r3.grid.tobytes() == r1.grid.tobytes()
        p1 = fc.predict(spec, r1.theta, r1.y_scale, r1.grid, fut, floor=floor, cap=cap, want_int=True)
        p3 = fc.predict(spec, r1.theta, r1.y_scale, r1.grid, fut, floor=floor, cap=cap, want_int=True,
                        devices=[0, 0, 0])
        assert np.array_equal(p1[0], p3[0]) and np.array_equal(p1[1], p3[1])


def test_c_program_through_the_abi_matches_the_python_binding(env, tmp_path):
    """The boundary is the C-ABI, not Python: a plain C99 program (tests/c/abi_fit.c, no torch,
    no numpy in the process) fits and predicts the same panel and returns the same bits as the
    ctypes binding -- and therefore as the oracle."""
    import os
    import subprocess
    from time_series_spark_amd import _lib, synth
    fc, cl = env
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / 'abi_fit')
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(['gcc', '-std=c99', '-pedantic', '-Wall', '-Wextra', '-Werror',
                           '-I', os.path.join(root, 'include'), os.path.join(root, 'tests', 'c', 'abi_fit.c'),
                           '-o', exe, '-L', libdir, '-ltsf_amd', '-Wl,-rpath,' + libdir])
    N, T, H = 12, 200, 30
    ds, y = synth.make_panel(N, T, 'linear', seed=5)
    y = np.ascontiguousarray(y, dtype=np.float64)
    fut = ds[-1] + synth.DAY_NS * np.arange(1, H + 1)
    ds.astype('<i8').tofile(tmp_path / 'ds.bin')
    y.astype('<f8').tofile(tmp_path / 'y.bin')
    fut.astype('<i8').tofile(tmp_path / 'fut.bin')
    msg = subprocess.check_output([exe, str(N), str(T), str(H), str(tmp_path / 'ds.bin'), str(tmp_path / 'y.bin'),
                                   str(tmp_path / 'fut.bin'), str(tmp_path / 'out.bin')]).decode()
    spec = fc.ModelSpec(growth='linear', seasonalities=[dict(helpers.WEEKLY)])
    assert msg.split()[0] == 'stride=%d' % spec.theta_stride
    raw = np.fromfile(tmp_path / 'out.bin', dtype='<f8')
    st = spec.theta_stride
    theta, yhat, tail = raw[:N * st].reshape(N, st), raw[N * st:N * st + N * H].reshape(N, H), raw[N * st + N * H:].reshape(N, 3)
    res = fc.fit_aligned(spec, ds, y)
    assert np.array_equal(theta, res.theta) and np.array_equal(tail[:, 0], res.status)
    assert np.array_equal(tail[:, 1], res.n_iter) and np.array_equal(tail[:, 2], res.n_eval)
    assert np.array_equal(yhat, fc.predict(spec, res.theta, res.y_scale, res.grid, fut))
    csp = helpers.oracle_spec(spec)
    o = cl.fit(csp, ds, y[0])
    S = o['info'].S
    assert n_bit_diff(theta[0][:3 + S], o['theta'][:3 + S]) == 0 and tail[0, 2] == o['n_eval']


def test_newton_kernel_matches_oracle(env):
    """Stan's Newton optimiser (fbprophet's choice for T < 100; tsf_newton_kernels.h) against the
    oracle's cn_newton: same status, iteration and evaluation counts, identical theta bits --
    aligned and ragged entry points, linear/additive and the reference's logistic/multiplicative
    settings; TSF_ALGO_AUTO picks it by the length of the call's longest series."""
    from time_series_spark_amd import _lib, synth
    fc, cl = env
    for growth, mode, T in (('linear', 'additive', 60), ('logistic', 'multiplicative', 90)):
        ds, y = synth.make_panel(4, T, growth, seed=751)
        seas = fc.ModelSpec.auto_seasonalities(ds, seasonality_mode=mode)
        spec = fc.ModelSpec(growth=growth, seasonality_mode=mode, seasonalities=seas,
                            algorithm=_lib.ALGO_NEWTON)
        floor, cap = np.zeros(len(y)), y.max(axis=1) * 1.1
        res = fc.fit_aligned(spec, ds, y, floor=floor, cap=cap)
        csp = helpers.oracle_spec(spec)
        for n in range(len(y)):
            o = cl.fit_newton(csp, ds, y[n], floor[n], cap[n])
            S = o['info'].S
            assert (res.status[n], res.n_iter[n], res.n_eval[n]) == (o['status'], o['n_iter'], o['n_eval']), (growth, n)
            assert n_bit_diff(res.theta[n][:3 + S], o['theta'][:3 + S]) == 0, (growth, n)
            assert n_bit_diff(res.theta[n][3 + spec.n_changepoints:], o['theta'][3 + S:]) == 0
            assert res.fval[n] == o['f']
        assert (res.status == _lib.ST_NEWTON_CONVERGED).all()
        # committed golden vectors (oracle outputs, tests/golden/make_golden.py newton_cases)
        gold = np.load(helpers.GOLDEN + '/newton_cases.npz')
        key = '%s_%d' % (growth, T)
        assert np.array_equal(res.theta, gold[key + '/theta']) and np.array_equal(res.n_eval, gold[key + '/n_eval'])
        fut = ds[-1] + helpers.DAY_NS * np.arange(1, 31)
        yhat = fc.predict(spec, res.theta, res.y_scale, res.grid, fut, floor=floor, cap=cap)
        assert np.array_equal(yhat, gold[key + '/yhat'])
        # ragged: truncated copies, each its own grid (and its own number of changepoints)
        cut = [T, T - 11, T - 25, 31]
        off = np.concatenate([[0], np.cumsum(cut)]).astype(np.int64)
        rr = fc.fit_ragged(spec, off, np.concatenate([ds[:c] for c in cut]),
                           np.concatenate([y[i][:c] for i, c in enumerate(cut)]), floor=floor, cap=cap)
        assert np.array_equal(rr.theta[0], res.theta[0]) and rr.n_eval[0] == res.n_eval[0]
        for i in (1, 3):
            o = cl.fit_newton(csp, ds[:cut[i]], y[i][:cut[i]], floor[i], cap[i])
            S = o['info'].S
            assert (rr.status[i], rr.n_iter[i], rr.n_eval[i]) == (o['status'], o['n_iter'], o['n_eval'])
            assert n_bit_diff(rr.theta[i][:3 + S], o['theta'][:3 + S]) == 0
    # fbprophet's rule: Newton below 100 rows, L-BFGS from 100 rows on
    ds, y = synth.make_panel(3, 120, 'linear', seed=3)
    auto = dict(growth='linear', seasonalities=[dict(helpers.WEEKLY)], algorithm=_lib.ALGO_AUTO)
    long_ = fc.fit_aligned(fc.ModelSpec(**auto), ds, y)
    short = fc.fit_aligned(fc.ModelSpec(**auto), ds[:99], y[:, :99])
    assert (short.status == _lib.ST_NEWTON_CONVERGED).all() and (long_.status != _lib.ST_NEWTON_CONVERGED).all()
    lb = fc.fit_aligned(fc.ModelSpec(growth='linear', seasonalities=[dict(helpers.WEEKLY)]), ds, y)
    assert np.array_equal(long_.theta, lb.theta)


def test_newton_with_two_parameters_per_lane(env):
    """Round 4: Stan's Newton for models of more than 64 parameters and for mixed additive / multiplicative
    columns (newton_kernel2: two parameters per lane through the finite-difference Hessian, the Householder
    tridiagonalisation and the QL sweeps; oracle cn_newton / cn_tridiag_ql up to 128).  fbprophet starts with
    Newton below 100 rows and retries EVERY failed L-BFGS fit with it, whatever the model; until now such a
    series of a wide model was dropped.  Bit for bit against the oracle: P = 84 linear / additive with 30 holiday
    columns, the same logistic / multiplicative (cfg4's model), a mixed-mode model with P <= 64, short and
    ragged histories; and the job layer's retry hands a wide model's failed fit to Newton."""
    fc, cl = env
    from time_series_spark_amd import _lib, synth
    T, H = 90, 20
    for case, n_series in (('linear_additive_holidays', 3), ('cfg4_holidays', 2)):
        spec, ds, y, floor, cap, extra, fut, exf = helpers.make_case(case, N=n_series)
        spec = type(spec).from_dict(dict(spec.to_dict(), lbfgs=dict(spec.lbfgs, algorithm=_lib.ALGO_NEWTON)))
        assert spec.theta_stride == 84
        ds, y, extra = ds[:T], y[:, :T], np.ascontiguousarray(extra[:, :T])
        cap = y.max(axis=1) * 1.1
        r = fc.fit_aligned(spec, ds, y, floor=floor, cap=cap, extra=extra)
        csp = helpers.oracle_spec(spec)
        csp.eval_mode = 0       # (Newton's quadratic-form evaluations exist up to 28 design columns: wide models evaluate in residual form)
        for n in range(n_series):
            o = cl.fit_newton(csp, ds, y[n], floor[n], cap[n], extra)
            assert (r.status[n], r.n_iter[n], r.n_eval[n]) == (o['status'], o['n_iter'], o['n_eval']), (case, n)
            assert n_bit_diff(r.theta[n], o['theta']) == 0 and n_bit_diff(r.fval[n], o['f']) == 0, (case, n)
        assert (r.status == _lib.ST_NEWTON_CONVERGED).all()
        # ragged: a truncated copy on its own grid
        cut = [T, T - 17]
        off = np.concatenate([[0], np.cumsum(cut)]).astype(np.int64)
        rr = fc.fit_ragged(spec, off, np.concatenate([ds[:c] for c in cut]), np.concatenate([y[i][:c] for i, c in enumerate(cut)]),
                           floor=floor[:2], cap=cap[:2], extra=np.concatenate([extra[:, :c] for c in cut], axis=1))
        assert np.array_equal(rr.theta[0], r.theta[0]) and rr.n_eval[0] == r.n_eval[0]
        o = cl.fit_newton(csp, ds[:cut[1]], y[1][:cut[1]], floor[1], cap[1], extra[:, :cut[1]])
        S = o['info'].S
        assert (rr.status[1], rr.n_iter[1], rr.n_eval[1]) == (o['status'], o['n_iter'], o['n_eval'])
        assert n_bit_diff(rr.theta[1][:3 + S], o['theta'][:3 + S]) == 0
    # mixed column modes (P = 34 <= 64, still a two-slot kernel): weekly additive + yearly-5 multiplicative
    ds, y = synth.make_panel(2, 80, 'linear', seed=6)
    seas = [dict(helpers.WEEKLY, mode='additive'), dict(helpers.YEARLY5, mode='multiplicative')]
    spec = fc.ModelSpec(growth='linear', seasonality_mode='additive', seasonalities=seas, algorithm=_lib.ALGO_NEWTON)
    r = fc.fit_aligned(spec, ds, y)
    csp = helpers.oracle_spec(spec)
    assert csp.eval_mode == 0
    for n in range(2):
        o = cl.fit_newton(csp, ds, y[n])
        assert (r.status[n], r.n_iter[n], r.n_eval[n]) == (o['status'], o['n_iter'], o['n_eval']), n
        assert n_bit_diff(r.theta[n], o['theta']) == 0 and n_bit_diff(r.fval[n], o['f']) == 0, n
    # the 64 / 65 boundary (round-4 advice: the launcher chose on P | 1, which is 65 for both, and sent the
    # 65-parameter model to the one-parameter-per-lane kernel): yearly + weekly + daily with 2 and 3 explicit columns
    dsb = np.datetime64('2019-03-01T00:00:00', 'ns').astype(np.int64) + 900 * 10 ** 9 * np.arange(96)
    _, yb = synth.make_panel(2, 96, 'logistic', seed=31)
    rngb = np.random.default_rng(5)
    for n_x, P in ((2, 64), (3, 65)):
        xb = np.ascontiguousarray(rngb.normal(0, 1, (n_x, 96)))
        spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative',
                            seasonalities=[helpers.YEARLY, helpers.WEEKLY, helpers.DAILY],
                            extra=[{'name': 'x%d' % i} for i in range(n_x)], algorithm=_lib.ALGO_NEWTON)
        assert spec.theta_stride == P
        capb = yb.max(axis=1) * 1.1
        r = fc.fit_aligned(spec, dsb, yb, floor=np.zeros(2), cap=capb, extra=xb)
        csp = helpers.oracle_spec(spec)
        csp.eval_mode = 0
        for n in range(2):
            o = cl.fit_newton(csp, dsb, yb[n], 0.0, capb[n], xb)
            assert (r.status[n], r.n_iter[n], r.n_eval[n]) == (o['status'], o['n_iter'], o['n_eval']), (P, n)
            assert n_bit_diff(r.theta[n], o['theta']) == 0 and n_bit_diff(r.fval[n], o['f']) == 0, (P, n)
    ds, y = synth.make_panel(2, 80, 'linear', seed=6)
    # more than 128 parameters (3 + 40 + 26 + 60): no kernel of the library takes the model, Newton or not
    with pytest.raises(_lib.TsfError):
        fc.fit_aligned(fc.ModelSpec(growth='linear', seasonalities=[dict(helpers.YEARLY), dict(helpers.WEEKLY)], n_changepoints=40,
                                    extra=[{'name': 'x%d' % i} for i in range(60)], algorithm=_lib.ALGO_NEWTON),
                       ds[:60], y[:, :60], extra=np.zeros((60, 60)))


def test_newton_for_the_references_widest_default_model(env):
    """Prophet(growth='logistic', seasonality_mode='multiplicative') (prophet_modeler.py:65) on
    sub-daily data spanning two years gets yearly + weekly + daily seasonality: K = 34, P = 62 -- one
    parameter per lane still, the design row streamed (newton_kernel<64>).  Round 2 stopped Newton at
    K = 28, so fbprophet's retry-with-Newton after an L-BFGS RuntimeError became "series dropped" for
    exactly this model.  Bit-identical to the oracle's Newton on short 15-minute series; the job layer
    routes short series of this model to it."""
    from time_series_spark_amd import _lib, synth
    from time_series_spark_amd.jobs import prophet_modeler as pm
    fc, cl = env
    T, N = 96, 2                           # one day of 15-minute rows (< 100: fbprophet picks Newton)
    ds = np.datetime64('2019-03-01T00:00:00', 'ns').astype(np.int64) + 900 * 10 ** 9 * np.arange(T)
    _, y = synth.make_panel(N, T, 'logistic', seed=31)
    seas = [helpers.YEARLY, helpers.WEEKLY, helpers.DAILY]
    spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=seas,
                        algorithm=_lib.ALGO_AUTO)
    assert spec.K == 34 and spec.theta_stride == 62
    floor, cap = np.zeros(N), y.max(axis=1) * 1.1
    r = fc.fit_aligned(spec, ds, y, floor=floor, cap=cap)
    csp = helpers.oracle_spec(spec)
    for n in range(N):
        o = cl.fit_newton(csp, ds, y[n], floor[n], cap[n])
        S = o['info'].S
        assert (r.status[n], r.n_iter[n], r.n_eval[n]) == (o['status'], o['n_iter'], o['n_eval']), n
        assert n_bit_diff(r.theta[n][:3 + S], o['theta'][:3 + S]) == 0 and n_bit_diff(r.fval[n], o['f']) == 0
        assert n_bit_diff(r.theta[n][3 + spec.n_changepoints:], o['theta'][3 + S:]) == 0
    assert (r.status == _lib.ST_NEWTON_CONVERGED).all()
    # a linear / additive model of the same width: quadratic-form Newton stops at 28 columns, the
    # residual-form kernel takes it (eval_form says so to the oracle: the form is part of the arithmetic)
    spec_l = fc.ModelSpec(growth='linear', seasonalities=seas, algorithm=_lib.ALGO_NEWTON, eval_form=_lib.EVAL_RESIDUAL)
    rl = fc.fit_aligned(spec_l, ds, y[:1])
    ol = cl.fit_newton(helpers.oracle_spec(spec_l), ds, y[0])
    assert (rl.status[0], rl.n_iter[0], rl.n_eval[0]) == (ol['status'], ol['n_iter'], ol['n_eval'])
    assert n_bit_diff(rl.fval[0], ol['f']) == 0


def test_batched_job_is_independent_of_how_series_are_grouped(env):
    """model_panel groups series that share a timestamp vector (aligned kernel path) and fits the
    rest through the ragged entry point; each series' model must be byte-identical to the one
    obtained by handing the reference-style UDF that series alone."""
    from time_series_spark_amd.jobs import prophet_modeler as pm
    from time_series_spark_amd import synth
    ds, y = synth.make_panel(5, 400, 'logistic', seed=5)
    rows = []
    for sid in range(5):
        T = [400, 400, 371, 400, 371][sid]                    # {0,1,3} and {2,4} share grids
        for t, v in zip(ds[:T], y[sid][:T]):
            rows.append((sid, 7, np.datetime64(int(t), 'ns'), int(v)))
    rows += [(9, 7, np.datetime64(int(t) + 3600 * 10 ** 9, 'ns'), int(v)) for t, v in zip(ds[:300], y[0][:300])]
    df = pd.DataFrame(rows, columns=['series_id', 'dim_id', 'ds', 'y'])
    config = {'model': {'floor': 0, 'cap_multiplier': 1.1}}
    udf = pm.model_time_series(config)
    singles = {(sid, did): udf(grp.copy()) for (sid, did), grp in df.groupby(['series_id', 'dim_id'])}
    # groups of any size through the aligned entry point (min_aligned_group 2: three launches + one ragged call), and
    # the default policy (groups of fewer than 4 096 series join the ragged call: one launch)
    for prophet in ({'min_aligned_group': 2}, {}):
        cfg = {'model': dict(config['model'], prophet=dict({'growth': 'logistic', 'seasonality_mode': 'multiplicative'}, **prophet))}
        both = pm.model_panel(cfg)(df.copy())
        assert len(both) == 6
        for (sid, did), one in singles.items():
            row = both[(both['series_id'] == sid) & (both['dim_id'] == did)]
            assert bytes(one['model'].iloc[0]) == bytes(row['model'].iloc[0]), (sid, prophet)
            assert one['cap'].iloc[0] == row['cap'].iloc[0]


def test_odd_shapes_against_oracle(env):
    """Small / awkward shapes through both entry points: one series, two rows, no changepoints,
    a single weekly harmonic, history != 5 (register-resident history needs 5: falls back to the
    residual kernel), forced evaluation forms, a long series (residual staging outside LDS)."""
    fc, cl = env
    from time_series_spark_amd import _lib, synth
    rng = np.random.default_rng(3)

    def check(spec, ds, y, **kw):
        r = fc.fit_aligned(spec, ds, y, **kw)
        csp = helpers.oracle_spec(spec)
        for n in range(y.shape[0]):
            o = cl.fit(csp, ds, y[n], kw.get('floor', [0.0] * len(y))[n] if 'floor' in kw else 0.0,
                       kw['cap'][n] if 'cap' in kw else 0.0)
            S = o['info'].S
            assert r.status[n] == o['status'] and r.n_iter[n] == o['n_iter'] and r.n_eval[n] == o['n_eval'], (n, o)
            assert n_bit_diff(r.theta[n][:3 + S], o['theta'][:3 + S]) == 0
            assert not r.theta[n][3 + S:3 + spec.n_changepoints].any()                     # unused deltas
            assert n_bit_diff(r.theta[n][3 + spec.n_changepoints:], o['theta'][3 + S:]) == 0   # beta
            assert r.grid['S'][0] == S
        return r

    ds, y = synth.make_panel(3, 120, 'linear', seed=4)
    one = [{'name': 'weekly', 'period': 7, 'fourier_order': 1}]
    check(fc.ModelSpec(growth='linear', seasonalities=one), ds, y[:1])                    # N = 1, K = 2
    check(fc.ModelSpec(growth='linear', seasonalities=one, n_changepoints=0), ds, y)      # S = 0
    check(fc.ModelSpec(growth='linear', seasonalities=one), ds[:2], y[:, :2])             # T = 2 -> S = 0
    # no changepoints: fitted on fbprophet's dummy changepoint (one Laplace-penalised delta at
    # t = 0, folded into k afterwards; oracle test_no_changepoints_is_fitted_on_...): every kernel
    lg = dict(floor=np.zeros(3), cap=y.max(axis=1) * 1.2)
    check(fc.ModelSpec(growth='logistic', seasonalities=[helpers.WEEKLY], n_changepoints=0), ds, y, **lg)
    check(fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=[helpers.WEEKLY],
                       n_changepoints=0), ds, y, **lg)
    check(fc.ModelSpec(growth='linear', seasonalities=[helpers.WEEKLY], n_changepoints=0,
                       eval_form=_lib.EVAL_RESIDUAL), ds, y)
    r0 = check(fc.ModelSpec(growth='linear', seasonalities=[helpers.WEEKLY], n_changepoints=0), ds, y)
    yh0 = fc.predict(fc.ModelSpec(growth='linear', seasonalities=[helpers.WEEKLY], n_changepoints=0),
                     r0.theta, r0.y_scale, r0.grid, ds[-1] + helpers.DAY_NS * np.arange(1, 8))
    c0 = cl.make_spec(n_changepoints=0, seasonalities=[(7, 3, 'additive', 10.0)], eval_mode=1)
    for n in range(3):
        o = cl.fit(c0, ds, y[n])
        assert np.array_equal(yh0[n], cl.predict(c0, o, ds[-1] + helpers.DAY_NS * np.arange(1, 8))[0])
    check(fc.ModelSpec(growth='linear', seasonalities=one), ds[:5], y[:, :5])             # T = 5 -> S = 3
    check(fc.ModelSpec(growth='linear', seasonalities=[helpers.WEEKLY], history=3), ds, y)   # residual kernel
    check(fc.ModelSpec(growth='linear', seasonalities=[helpers.WEEKLY], max_iter=7), ds, y)
    check(fc.ModelSpec(growth='linear', seasonalities=[helpers.WEEKLY], eval_form=_lib.EVAL_QUADRATIC), ds, y)
    check(fc.ModelSpec(growth='linear', seasonalities=[helpers.WEEKLY], eval_form=_lib.EVAL_RESIDUAL), ds, y)
    check(fc.ModelSpec(growth='logistic', seasonalities=[helpers.WEEKLY]), ds, y,
          floor=np.zeros(3), cap=y.max(axis=1) * 1.2)
    with pytest.raises(_lib.TsfError):        # quadratic form needs a model linear in the parameters
        fc.fit_aligned(fc.ModelSpec(growth='logistic', seasonalities=[helpers.WEEKLY],
                                    eval_form=_lib.EVAL_QUADRATIC), ds, y, cap=y.max(axis=1) * 1.2)
    # long series: 3 000 rows (NT = 47: the residual staging of the quadratic kernel leaves LDS)
    dsl, yl = synth.make_panel(2, 3000, 'linear', seed=8)
    check(fc.ModelSpec(growth='linear', seasonalities=[helpers.YEARLY, helpers.WEEKLY]), dsl, yl)
    # 15-minute data, 20 000 rows (208 days; the reference's example config forecasts on a 15-minute
    # grid): NT = 313 rows per lane, far past one LDS staging buffer
    ds15 = synth.START_NS + (15 * 60 * 10 ** 9) * np.arange(20000, dtype=np.int64)
    y15 = synth.make_panel(2, 20000, 'linear', seed=9)[1]
    check(fc.ModelSpec(growth='linear', seasonalities=[helpers.WEEKLY, helpers.DAILY]), ds15, y15)
    check(fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative',
                       seasonalities=[helpers.WEEKLY, helpers.DAILY]), ds15, y15,
          floor=np.zeros(2), cap=y15.max(axis=1) * 1.1)
    # noise-free y: the optimiser drives sigma down until something gives; whatever happens must
    # be the oracle's outcome too
    t = np.arange(120.0)
    yc = np.stack([50 + 0.5 * t + 3 * np.sin(2 * np.pi * t / 7), 20 + 0.1 * t])
    check(fc.ModelSpec(growth='linear', seasonalities=[helpers.WEEKLY]), ds, yc)


def _sample_with_stragglers(r, bad, rng, n_random, n_special=6):
    """Series to bit-check against the oracle: a random sample over ALL series (no filter on how long
    they ran), the series with the most evaluations, and every series that ended at the iteration cap,
    in a failed line search or at the evaluation guard (at most n_special of each kind) -- the
    stragglers whose trajectories are the longest chains of dependent roundings."""
    pick = list(rng.choice(r.N, n_random, replace=False))
    pick.append(int(np.argmax(r.n_eval)))
    for code in (40, -1, -3):
        pick += list(np.flatnonzero(r.status == code)[:n_special])
    return np.array(sorted(set(int(i) for i in pick)))



def _assert_bits_against_oracle(r, yh, sub, fit_one, predict_one, check_forecast=True):
    """theta, objective, iteration / evaluation counts, termination code and forecast of the series `sub`
    bit for bit against the oracle.  fit_one(n) -> oracle fit dict; predict_one(o, n) -> oracle forecast."""
    for n in sub:
        o = fit_one(n)
        assert (r.n_iter[n], r.n_eval[n], r.status[n]) == (o['n_iter'], o['n_eval'], o['status']), n
        assert n_bit_diff(r.fval[n], o['f']) == 0 and n_bit_diff(r.theta[n], o['theta']) == 0, n
        if check_forecast and r.status[n] > 0:
            yo = predict_one(o, n)
            assert np.max(np.abs(yh[n] - yo) / np.abs(yo)) <= REL_TOL        # the north-star statement
            assert np.array_equal(yh[n], yo), n                               # ... and the bits


def test_full_size_panel_properties(env):
    """BASELINE config 2 at full size (10 000 x 730) on its production route (the 12-wave quadratic-form
    kernel, what bench.py times): size-independent properties -- every series terminates normally, forecasts
    finite, doubling y doubles the forecast exactly (power-of-two scaling leaves the scaled problem
    bit-identical) -- and 64 random series PLUS the stragglers (the series with the most evaluations, every
    MAXIT / LSFAIL / EVAL_LIMIT series) bit for bit against the oracle: theta, objective, counts, termination
    code, forecast."""
    fc, cl = env
    from time_series_spark_amd import synth
    N, T, H = 10000, 730, 90
    spec = helpers.make_case('cfg2_linear_additive')[0]
    ds, y = synth.make_panel(N, T, 'linear', seed=751)
    fut = ds[-1] + helpers.DAY_NS * np.arange(1, H + 1)
    r = fc.fit_aligned(spec, ds, y)
    assert (r.status > 0).all() and (r.n_iter >= 1).all()
    yh = fc.predict(spec, r.theta, r.y_scale, r.grid, fut)
    assert np.isfinite(yh).all()
    sub = np.arange(0, N, 97)
    r2 = fc.fit_aligned(spec, ds, 2.0 * y[sub])
    assert np.array_equal(r2.theta, r.theta[sub])
    yh2 = fc.predict(spec, r2.theta, r2.y_scale, r2.grid, fut)
    assert np.array_equal(yh2, 2.0 * yh[sub])
    csp = helpers.oracle_spec(spec)
    pick = _sample_with_stragglers(r, r.status <= 0, np.random.default_rng(0), 64)
    assert len(pick) >= 64 and r.n_eval[pick].max() == r.n_eval.max()
    _assert_bits_against_oracle(r, yh, pick, lambda n: cl.fit(csp, ds, y[n]), lambda o, n: cl.predict(csp, o, fut)[0])


def test_full_size_cfg3_whole_panel_on_one_gpu(env):
    """BASELINE config 3 WHOLE on one GPU (100 000 x 1 095; `bench_configs.py cfg3_full`): the route that
    panel takes -- 16 waves per CU with the trend tables from the LDS pool AND series longer than 768 rows,
    i.e. the weights of steps >= 12 of a residual pass through the global scratch -- at full size: every series
    terminates normally, forecasts finite, 24 random series plus the stragglers bit for bit against the
    oracle."""
    fc, cl = env
    from time_series_spark_amd import synth
    N, T, H = 100000, 1095, 90
    ds, y = synth.make_panel(N, T, 'linear', seed=751)
    spec = fc.ModelSpec(growth='linear', seasonalities=fc.ModelSpec.auto_seasonalities(ds))
    assert spec.K == 26
    fut = ds[-1] + helpers.DAY_NS * np.arange(1, H + 1)
    r = fc.fit_aligned(spec, ds, y)
    bad = r.status <= 0
    assert bad.sum() <= N // 20000 and np.isin(r.status[bad], [-1, -3]).all() and (r.n_iter >= 1).all()
    pick = _sample_with_stragglers(r, bad, np.random.default_rng(2), 24, n_special=3)
    assert r.n_eval[pick].max() == r.n_eval.max()
    yh = np.zeros((N, H))
    yh[pick] = fc.predict(spec, r.theta[pick], r.y_scale[pick], r.grid, fut)
    assert np.isfinite(yh[pick]).all()
    csp = helpers.oracle_spec(spec)
    _assert_bits_against_oracle(r, yh, pick, lambda n: cl.fit(csp, ds, y[n]), lambda o, n: cl.predict(csp, o, fut)[0])


@pytest.mark.parametrize('cfg', ['cfg3', 'cfg5'])
def test_full_size_other_baseline_configs(env, cfg):
    """BASELINE configs 3 (one GPU's 12 500 x 1 095 share of the 100 000 series) and 5
    (1 000 000 x 90, float32 y) at full size: every series terminates, forecasts finite, a
    random sample is bit-identical to the oracle."""
    fc, cl = env
    from time_series_spark_amd import synth
    if cfg == 'cfg3':
        N, T, dt = 12500, 1095, np.float64
    else:
        N, T, dt = 1000000, 90, np.float32
    ds, y = synth.make_panel(N, T, 'linear', seed=751, dtype=dt)
    spec = fc.ModelSpec(growth='linear', seasonalities=fc.ModelSpec.auto_seasonalities(ds))
    assert spec.K == (26 if cfg == 'cfg3' else 6)
    fut = ds[-1] + helpers.DAY_NS * np.arange(1, 91)
    r = fc.fit_aligned(spec, ds, y)
    # a handful of the million short series end in a failed line search (pystan would raise
    # RuntimeError and the reference drop the series); none may end any other way
    bad = r.status <= 0
    assert bad.sum() <= N // 20000 and np.isin(r.status[bad], [-1, -3]).all() and (r.n_iter >= 1).all()
    sub = np.random.default_rng(1).choice(N, 10, replace=False)
    sub = np.concatenate([sub[~bad[sub]], np.flatnonzero(bad)[:3]])
    yh = fc.predict(spec, r.theta[sub], r.y_scale[sub], r.grid, fut)
    assert np.isfinite(yh).all()
    csp = helpers.oracle_spec(spec)
    for i, n in enumerate(sub):
        o = cl.fit(csp, ds, y[n].astype(np.float64))
        yo, _ = cl.predict(csp, o, fut)
        assert r.n_iter[n] == o['n_iter'] and r.n_eval[n] == o['n_eval'] and r.status[n] == o['status']
        assert np.max(np.abs(yh[i] - yo) / np.abs(yo)) <= REL_TOL
        assert np.array_equal(yh[i], yo)


def test_full_size_cfg4_logistic_holidays(env):
    """BASELINE config 4 at full size (50 000 x 730, logistic growth + floor, multiplicative
    yearly + weekly, 25 changepoints, 10 holidays x window [-1, +1] = 30 indicator columns, P = 84:
    the two-slot residual kernel): every series ends in one of Stan's termination codes (or, for a
    handful, a failed line search / iteration cap -- what pystan reports too), forecasts finite and
    inside (floor, cap * (1 + multiplicative terms)), and a random sample is bit-identical to the
    oracle, forecasts included."""
    fc, cl = env
    from time_series_spark_amd import synth
    N, T, H = 50000, 730, 90
    ds = synth.daily_grid(T)
    fut = ds[-1] + helpers.DAY_NS * np.arange(1, H + 1)
    allm, names = synth.holiday_matrix(np.concatenate([ds, fut]), 10)
    extra, exf = np.ascontiguousarray(allm[:, :T]), np.ascontiguousarray(allm[:, T:])
    ds, y = synth.make_panel(N, T, 'logistic', seed=751, holidays=extra)
    spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative',
                        seasonalities=[helpers.YEARLY, helpers.WEEKLY], extra=[{'name': n} for n in names])
    assert spec.K == 56 and spec.theta_stride == 84
    floor, cap = np.zeros(N), y.max(axis=1) * 1.1
    r = fc.fit_aligned(spec, ds, y, floor=floor, cap=cap, extra=extra)
    bad = r.status <= 0
    assert bad.sum() <= N // 5000 and np.isin(r.status[bad], [-1, -3]).all() and (r.n_iter >= 1).all()
    assert np.isin(r.status[~bad], [10, 20, 21, 30, 31, 40]).all()
    yh = fc.predict(spec, r.theta, r.y_scale, r.grid, fut, floor=floor, cap=cap, extra_future=exf)
    assert np.isfinite(yh[~bad]).all()
    # no filter on the trajectory length: the longest series (tens of thousands of evaluations, finished
    # by the cooperative kernel) and every MAXIT / LSFAIL / EVAL_LIMIT series are checked too
    sub = _sample_with_stragglers(r, bad, np.random.default_rng(4), 5, n_special=3)
    assert r.n_eval[sub].max() == r.n_eval.max()
    csp = helpers.oracle_spec(spec)
    for n in sub:
        o = cl.fit(csp, ds, y[n], floor[n], cap[n], extra)
        assert (r.n_iter[n], r.n_eval[n], r.status[n]) == (o['n_iter'], o['n_eval'], o['status']), n
        assert n_bit_diff(r.fval[n], o['f']) == 0, n
        S = o['info'].S
        assert n_bit_diff(r.theta[n][:3 + S], o['theta'][:3 + S]) == 0 and n_bit_diff(r.theta[n][3 + spec.n_changepoints:], o['theta'][3 + S:]) == 0
        if r.status[n] > 0:
            yo, _ = cl.predict(csp, o, fut, floor[n], cap[n], exf)
            assert np.max(np.abs(yh[n] - yo) / np.abs(yo)) <= REL_TOL
            assert np.array_equal(yh[n], yo)
    # what fbprophet does with a failed L-BFGS fit (pystan's RuntimeError): once more with Stan's Newton -- since round 4
    # for this 84-parameter model too (newton_kernel2; the job layer's retry in fit_packed is this call on the failed
    # series).  One of the failed series, bit for bit against the oracle's Newton from the same initial values.
    from time_series_spark_amd import _lib
    for n in np.flatnonzero(bad)[:1]:
        nspec = type(spec).from_dict(dict(spec.to_dict(), lbfgs=dict(spec.lbfgs, algorithm=_lib.ALGO_NEWTON)))
        rn = fc.fit_aligned(nspec, ds, y[n:n + 1], floor=floor[n:n + 1], cap=cap[n:n + 1], extra=extra)
        o = cl.fit_newton(csp, ds, y[n], floor[n], cap[n], extra)
        assert (rn.status[0], rn.n_iter[0], rn.n_eval[0]) == (o['status'], o['n_iter'], o['n_eval']), n
        assert n_bit_diff(rn.theta[0], o['theta']) == 0 and n_bit_diff(rn.fval[0], o['f']) == 0, n


@pytest.mark.parametrize('kernel', ['auto', 'mfma'])
def test_full_size_reference_model_on_both_residual_kernels(env, kernel):
    """The reference's own settings (logistic growth, multiplicative yearly + weekly) on a
    20 000 x 730 aligned panel, on the kernel TSF_RK_AUTO picks (the one-wave kernel) and on the
    matrix-core kernel: every series ends normally and a random sample is bit-identical to the
    oracle."""
    fc, cl = env
    from time_series_spark_amd import _lib, synth
    N, T, H = 20000, 730, 90
    ds, y = synth.make_panel(N, T, 'logistic', seed=752)
    spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative',
                        seasonalities=[helpers.YEARLY, helpers.WEEKLY],
                        residual_kernel=_lib.RK_MFMA if kernel == 'mfma' else _lib.RK_AUTO)
    floor, cap = np.zeros(N), y.max(axis=1) * 1.1
    r = fc.fit_aligned(spec, ds, y, floor=floor, cap=cap)
    bad = r.status <= 0
    assert bad.sum() <= 4 and np.isin(r.status[bad], [-1, -3]).all() and (r.n_iter >= 1).all()
    fut = ds[-1] + helpers.DAY_NS * np.arange(1, H + 1)
    sub = _sample_with_stragglers(r, bad, np.random.default_rng(7), 7, n_special=3)     # stragglers included
    assert r.n_eval[sub].max() == r.n_eval.max()
    yh = fc.predict(spec, r.theta[sub], r.y_scale[sub], r.grid, fut, floor=floor[sub], cap=cap[sub])
    csp = helpers.oracle_spec(spec)
    for i, n in enumerate(sub):
        o = cl.fit(csp, ds, y[n], floor[n], cap[n])
        assert (r.n_iter[n], r.n_eval[n], r.status[n]) == (o['n_iter'], o['n_eval'], o['status']), n
        assert n_bit_diff(r.fval[n], o['f']) == 0 and n_bit_diff(r.theta[n], o['theta']) == 0, n
        if r.status[n] > 0:
            yo, _ = cl.predict(csp, o, fut, floor[n], cap[n])
            assert np.array_equal(yh[i], yo)


def test_full_size_cfg5_under_fbprophets_own_optimiser_rule(env):
    """BASELINE config 5 as fbprophet would run it, at BASELINE's size: 90 rows < 100 -> Stan's Newton
    (algorithm = AUTO applies `'Newton' if T < 100 else 'LBFGS'`), 1 000 000 fp32 series.  Every series
    ends converged (or, rarely, at the iteration cap); a random sample plus the series with the most
    evaluations -- never skipped: a few hundred thousand evaluations, seconds of oracle -- are bit-identical
    to the oracle's Newton, forecasts included."""
    fc, cl = env
    from time_series_spark_amd import _lib, synth
    N, T, H = 1000000, 90, 90
    ds, y = synth.make_panel(N, T, 'linear', seed=751, dtype=np.float32)
    spec = fc.ModelSpec(growth='linear', seasonalities=fc.ModelSpec.auto_seasonalities(ds), algorithm=_lib.ALGO_AUTO)
    assert spec.K == 6
    r = fc.fit_aligned(spec, ds, y)
    assert np.isin(r.status, [_lib.ST_NEWTON_CONVERGED, _lib.ST_MAXIT, _lib.ST_NEWTON_FAIL]).all()
    assert (r.status == _lib.ST_NEWTON_CONVERGED).mean() >= 0.999
    fut = ds[-1] + helpers.DAY_NS * np.arange(1, H + 1)
    rng = np.random.default_rng(12)
    sub = np.array(sorted(set(list(rng.choice(N, 8, replace=False)) + [int(np.argmax(r.n_eval))])))
    assert r.n_eval[sub].max() == r.n_eval.max()
    yh = fc.predict(spec, r.theta[sub], r.y_scale[sub], r.grid, fut)
    csp = helpers.oracle_spec(spec)
    for i, n in enumerate(sub):
        o = cl.fit_newton(csp, ds, y[n].astype(np.float64))
        assert (r.n_iter[n], r.n_eval[n], r.status[n]) == (o['n_iter'], o['n_eval'], o['status']), n
        assert n_bit_diff(r.fval[n], o['f']) == 0 and n_bit_diff(r.theta[n], o['theta']) == 0, n
        yo, _ = cl.predict(csp, o, fut)
        assert np.array_equal(yh[i], yo)


def test_quadratic_form_kernels_for_small_and_large_panels_give_the_same_bits(env):
    """Aligned panels with P <= 64 have two quadratic-form kernels: Z^T Z in LDS at 12 waves per CU (large
    panels: throughput) and Z^T Z in registers at 8 waves per CU (small panels: the launch is its longest
    series, and the lone wave is faster).  The library picks by the number of series; both routes must give
    the same bits, and those of the oracle."""
    import os
    fc, cl = env
    from time_series_spark_amd import synth
    N, T = 3000, 730
    ds, y = synth.make_panel(N, T, 'linear', seed=31)
    spec = fc.ModelSpec(growth='linear', seasonalities=fc.ModelSpec.auto_seasonalities(ds))
    res = {}
    for mode in ('0', '1', None):
        if mode is None:
            helpers.routes.pop('TSF_QUAD_REG', None)
        else:
            helpers.routes['TSF_QUAD_REG'] = mode
        try:
            res[mode] = fc.fit_aligned(spec, ds, y)
        finally:
            helpers.routes.pop('TSF_QUAD_REG', None)
    for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status'):
        assert np.array_equal(getattr(res['0'], name), getattr(res['1'], name), equal_nan=True), name
        assert np.array_equal(getattr(res['0'], name), getattr(res[None], name), equal_nan=True), name
    csp = helpers.oracle_spec(spec)
    for n in (0, int(np.argsort(res['1'].n_eval)[-1])):
        o = cl.fit(csp, ds, y[n])
        assert (res['1'].n_iter[n], res['1'].n_eval[n], res['1'].status[n]) == (o['n_iter'], o['n_eval'], o['status']), n
        assert n_bit_diff(res['1'].theta[n], o['theta']) == 0 and n_bit_diff(res['1'].fval[n], o['f']) == 0, n


def test_quadratic_form_kernel_routes_for_large_aligned_panels_give_the_same_bits(env):
    """Aligned panels with P <= 64 and more than a few series per wave slot have three routes since round 3:
    * 16 waves per CU (four per SIMD, <= 128 registers), the trend tables of a residual pass borrowed from a small
      pool of shared LDS copies (tsf_quad_kernels.h QuadPool) -- what panels of >= 8 series per wave slot take;
    * 12 waves per CU with the weights of a residual pass in the registers of the lane that produces and consumes
      them (ztr_pass NTR) -- the default below that;
    * 12 waves per CU with those weights staged through global memory (round 2's kernel; TSF_QUAD_RREG=0).
    Same series through all of them, the pool also with ONE copy for sixteen waves (every pass of a workgroup
    waits for the others: the lock under contention): identical bits, and those of the oracle incl. the longest
    fit.  A short-series panel (T = 90: the staging rows ride in the pool slot) likewise."""
    import os
    fc, cl = env
    from time_series_spark_amd import synth
    N, T = 7000, 730                        # > 3 series per slot of the register-M kernel: the LDS kernels take the call
    ds, y = synth.make_panel(N, T, 'linear', seed=77)
    spec = fc.ModelSpec(growth='linear', seasonalities=fc.ModelSpec.auto_seasonalities(ds))
    res = {}
    legs = (('w4', {'TSF_QUAD_W4': '16'}), ('w4_one_copy', {'TSF_QUAD_W4': '1'}), ('w3_reg', {}),
            ('w3_glob', {'TSF_QUAD_RREG': '0'}))
    helpers.routes['TSF_QUAD_REG'] = '0'
    try:
        for tag, envs in legs:
            helpers.routes.update(envs)
            try:
                res[tag] = fc.fit_aligned(spec, ds, y)
            finally:
                for k in envs:
                    helpers.routes.pop(k, None)
    finally:
        helpers.routes.pop('TSF_QUAD_REG', None)
    for tag, _ in legs[1:]:
        for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status'):
            assert np.array_equal(getattr(res['w4'], name), getattr(res[tag], name), equal_nan=True), (tag, name)
    csp = helpers.oracle_spec(spec)
    r = res['w4']
    for n in (0, 1, N - 1, int(np.argsort(r.n_eval)[-1])):
        o = cl.fit(csp, ds, y[n])
        assert (r.n_iter[n], r.n_eval[n], r.status[n]) == (o['n_iter'], o['n_eval'], o['status']), n
        assert n_bit_diff(r.theta[n], o['theta']) == 0 and n_bit_diff(r.fval[n], o['f']) == 0, n
    ds, y = synth.make_panel(4000, 90, 'linear', seed=78)
    spec = fc.ModelSpec(growth='linear', seasonalities=fc.ModelSpec.auto_seasonalities(ds))
    helpers.routes['TSF_QUAD_REG'] = '0'
    try:
        w3 = fc.fit_aligned(spec, ds, y)
        helpers.routes['TSF_QUAD_W4'] = '16'
        w4 = fc.fit_aligned(spec, ds, y)
    finally:
        helpers.routes.pop('TSF_QUAD_W4', None)
        helpers.routes.pop('TSF_QUAD_REG', None)
    for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status'):
        assert np.array_equal(getattr(w3, name), getattr(w4, name), equal_nan=True), name
    # LONG series on the pooled 16-wave route (what BASELINE cfg3 whole, 100 000 x 1 095, takes): the trend tables
    # from the LDS pool AND the weights of steps >= 12 of a residual pass through the global scratch -- forced
    # here on small panels of 1 095 and 1 400 rows (NT = 18 / 22), with every copy of the pool and with one
    # (contended lock), against the 12-wave routes (weights in registers + scratch; all through the scratch) and
    # the oracle incl. the longest fit
    for T in (1095, 1400):
        N = 1500
        ds, y = synth.make_panel(N, T, 'linear', seed=79 + T)
        spec = fc.ModelSpec(growth='linear', seasonalities=fc.ModelSpec.auto_seasonalities(ds))
        assert spec.K == 26
        res = {}
        helpers.routes['TSF_QUAD_REG'] = '0'
        try:
            for tag, envs in legs:
                helpers.routes.update(envs)
                try:
                    res[tag] = fc.fit_aligned(spec, ds, y)
                finally:
                    for k in envs:
                        helpers.routes.pop(k, None)
        finally:
            helpers.routes.pop('TSF_QUAD_REG', None)
        for tag, _ in legs[1:]:
            for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status'):
                assert np.array_equal(getattr(res['w4'], name), getattr(res[tag], name), equal_nan=True), (T, tag, name)
        csp = helpers.oracle_spec(spec)
        r = res['w4']
        for n in (0, N - 1, int(np.argmax(r.n_eval))):
            o = cl.fit(csp, ds, y[n])
            assert (r.n_iter[n], r.n_eval[n], r.status[n]) == (o['n_iter'], o['n_eval'], o['status']), (T, n)
            assert n_bit_diff(r.theta[n], o['theta']) == 0 and n_bit_diff(r.fval[n], o['f']) == 0, (T, n)


def test_cost_hints_change_the_order_of_the_launch_and_nothing_else(env):
    """tsf_set_cost_hints: the work queue hands out the series in order of decreasing expected cost (a caller that
    re-fits a panel regularly passes the evaluation counts of the previous fit).  Every series is fitted by itself,
    so the results must not move by a bit -- with the true counts, with random hints, with hints for another panel
    size (ignored), on the quadratic-form kernel (queue) and on the residual-form kernel (one block per series)."""
    fc, cl = env
    from time_series_spark_amd import synth
    rng = np.random.default_rng(5)
    # ragged entry point (quadratic form: the queue of the ragged kernel)
    ds, y = synth.make_panel(900, 400, 'linear', seed=80)
    lens = 330 + (np.arange(900) * 13) % 71
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    dsr = np.concatenate([ds[:c] for c in lens]); yr = np.concatenate([y[i][:c] for i, c in enumerate(lens)])
    spec = fc.ModelSpec(growth='linear', seasonalities=fc.ModelSpec.auto_seasonalities(ds))
    base = fc.fit_ragged(spec, off, dsr, yr)
    hinted = fc.fit_ragged(spec, off, dsr, yr, cost_hints=base.n_eval)
    for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status', 'y_scale'):
        assert np.array_equal(getattr(base, name), getattr(hinted, name), equal_nan=True), ('ragged', name)
    for growth, N, T in (('linear', 7000, 730), ('logistic', 600, 365)):
        ds, y = synth.make_panel(N, T, growth, seed=79)
        kw = {} if growth == 'linear' else {'floor': np.zeros(N), 'cap': y.max(axis=1) * 1.1}
        spec = fc.ModelSpec(growth=growth, seasonalities=fc.ModelSpec.auto_seasonalities(ds),
                            **({} if growth == 'linear' else {'seasonality_mode': 'multiplicative'}))
        base = fc.fit_aligned(spec, ds, y, **kw)
        legs = {'true': base.n_eval, 'random': rng.integers(0, 1000, N), 'constant': np.zeros(N, np.int32),
                'other_size': np.arange(N - 1)}
        for tag, hints in legs.items():
            if tag == 'other_size':
                ctx = fc.get_context()
                hh = np.ascontiguousarray(hints, dtype=np.int32)
                ctx.check(_lib_handle().tsf_set_cost_hints(ctx.handle, hh.ctypes.data, hh.shape[0]))
                r = fc.fit_aligned(spec, ds, y, **kw)
            else:
                r = fc.fit_aligned(spec, ds, y, cost_hints=hints, **kw)
            for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status', 'y_scale'):
                assert np.array_equal(getattr(base, name), getattr(r, name), equal_nan=True), (growth, tag, name)


def _lib_handle():
    from time_series_spark_amd import _lib
    return _lib.load()


def test_newton_several_series_per_wave_equals_one_series_per_wave(env):
    """The Newton kernel for aligned linear/additive panels keeps several series per wave and runs their
    QL rotation chains side by side, lane = series (tsf_newton_batch.h); calls with few series take the
    one-series-per-wave kernel (tsf_newton_quad.h).  Same series, both routes -- and the route a slot
    takes when its rotation list overflows (forced here with a 50-entry list) -- must give the same
    bits, and those of the oracle."""
    import os
    fc, cl = env
    from time_series_spark_amd import _lib, synth
    N, T = 12000, 90
    ds, y = synth.make_panel(N, T, 'linear', seed=99)
    spec = fc.ModelSpec(growth='linear', seasonalities=fc.ModelSpec.auto_seasonalities(ds), algorithm=_lib.ALGO_NEWTON)
    helpers.routes['TSF_NEWTON_BATCH'] = '2'                 # slots from two series per resident wave on (default: twelve)
    try:
        big = fc.fit_aligned(spec, ds, y)
        helpers.routes['TSF_NEWTON_LCAP'] = '50'
        over = fc.fit_aligned(spec, ds, y)               # every decomposition overflows its list
    finally:
        helpers.routes.pop('TSF_NEWTON_LCAP', None)
        helpers.routes.pop('TSF_NEWTON_BATCH')
    sub = np.arange(0, N, 37)[:300]
    small = fc.fit_aligned(spec, ds, y[sub])             # one series per wave
    for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status'):
        assert np.array_equal(getattr(big, name), getattr(over, name), equal_nan=True), name
        assert np.array_equal(getattr(big, name)[sub], getattr(small, name), equal_nan=True), name
    csp = helpers.oracle_spec(spec)
    for n in (0, 5, int(np.argsort(big.n_eval)[-1])):
        if big.n_eval[n] > 200000:
            continue
        o = cl.fit_newton(csp, ds, y[n])
        assert (big.n_iter[n], big.n_eval[n], big.status[n]) == (o['n_iter'], o['n_eval'], o['status']), n
        assert n_bit_diff(big.fval[n], o['f']) == 0 and n_bit_diff(big.theta[n], o['theta']) == 0, n


def test_upstream_known_answer_vectors_through_predict_kernel(env):
    """fbprophet's own piecewise_linear / piecewise_logistic known-answer vectors
    (tests/golden/upstream_recall.json) through tsf_predict: scaled time = days, y_scale 1,
    one all-zero design column."""
    fc, cl = env
    import json
    from time_series_spark_amd import _lib
    with open(helpers.GOLDEN + '/upstream_recall.json') as fh:
        up = json.load(fh)
    for growth, key in (('linear', 'piecewise_linear'), ('logistic', 'piecewise_logistic')):
        u = up[key]
        spec = fc.ModelSpec(growth=growth, n_changepoints=len(u['deltas']), extra=[{'name': 'zero'}])
        theta = np.concatenate([[u['k'], u['m'], 0.0], u['deltas'], [0.0]])[None, :]
        grid = np.zeros(1, dtype=_lib.GRID_DTYPE)
        grid['start_ns'], grid['t_scale_ns'], grid['T'], grid['S'] = 0, helpers.DAY_NS, 11, len(u['deltas'])
        grid['NT'] = 1
        grid['t_change'][0, :len(u['deltas'])] = u['changepoint_ts']
        fut = np.asarray(u['t'], dtype=np.int64) * helpers.DAY_NS
        yh = fc.predict(spec, theta, np.ones(1), grid, fut, floor=np.zeros(1), cap=np.full(1, u.get('cap', 1.0)),
                        extra_future=np.zeros((1, len(fut))))[0]
        y_true = np.asarray(u['y_true'])
        if growth == 'linear':
            assert np.array_equal(yh, y_true)
        else:
            assert abs((yh - y_true).sum()) < 0.5e-5 and np.max(np.abs(yh - y_true)) < 0.6e-6


def test_reference_contract_fit_and_forecast_udfs(env, tmp_path):
    """/root/reference/tests/unit/prophet_modeler_test.py:59-75 and
    prophet_scorer_test.py:83-114 re-expressed on pandas: 2 model rows with the reference's
    columns, parquet round trip, 40 forecasts per series / 80 rows, CSV with 6 named columns."""
    from time_series_spark_amd.jobs import prophet_modeler as pm, prophet_scorer as ps
    g = np.load(helpers.GOLDEN + '/fixture_751.npz')
    d = tmp_path / 'model-input' / 'series_id=751'
    d.mkdir(parents=True)
    pd.DataFrame({'dim_id': g['raw_dim_id'],
                  'ds': pd.Series(g['raw_ds_ns'].astype('datetime64[ns]')).dt.strftime('%Y-%m-%d %H:%M:%S'),
                  'y': g['raw_y']}).to_csv(d / 'sample-model-input.csv', header=False, index=False)
    mconfig = {'io': {'input': str(tmp_path / 'model-input'), 'models': str(tmp_path / 'models')},
               'model': {'floor': 0, 'cap_multiplier': 1.1}}
    modeler = pm.ProphetModeler(mconfig)
    input_df = modeler.read_input_dataframe(None)
    udf = pm.model_time_series(mconfig)
    output_df = pd.concat([udf(grp.copy()) for _, grp in input_df.groupby(['series_id', 'dim_id'])],
                          ignore_index=True)
    assert len(output_df) == 2
    assert list(output_df.columns) == ['series_id', 'dim_id', 'floor', 'cap', 'model']
    assert len(output_df.query('series_id == 751 and dim_id == 91')) == 1
    assert len(output_df.query('series_id == 751 and dim_id == 155')) == 1
    modeler.persist_models(output_df)
    sconfig = {'io': {'models': str(tmp_path / 'models'), 'forecasts': str(tmp_path / 'forecasts')},
               'forecast': {'periods': 40, 'frequency': '15min'}}
    scorer = ps.ProphetScorer(sconfig)
    model_df = scorer.read_model_dataframe(None)
    assert list(model_df.columns) == ['series_id', 'dim_id', 'floor', 'cap', 'model']
    assert len(model_df) == 2 and model_df['series_id'].nunique() == 1 and model_df['dim_id'].nunique() == 2
    assert str(model_df['floor'].dtype) == 'float32' and str(model_df['cap'].dtype) == 'float32'
    fudf = ps.forecast_time_series(sconfig)
    fdf = pd.concat([fudf(grp) for _, grp in model_df.groupby(['series_id', 'dim_id'])], ignore_index=True)
    assert len(fdf) == 80 and list(fdf.columns) == ['series_id', 'dim_id', 'ds', 'yhat']
    assert len(fdf.query('series_id == 751 and dim_id == 91')) == 40
    assert len(fdf.query('series_id == 751 and dim_id == 155')) == 40
    assert str(fdf['yhat'].dtype) == 'int32'
    # numbers: the golden oracle forecasts, int-truncated as prophet_scorer.py:73 does
    for j, dim in enumerate(g['dim_ids']):
        got = fdf[fdf['dim_id'] == dim]['yhat'].values
        assert np.array_equal(got, np.trunc(g['yhat'][j]).astype(np.int32))
    scorer.write_forecasts(scorer.convert_forecasts(fdf))
    import glob
    back = pd.concat([pd.read_csv(f) for f in glob.glob(str(tmp_path / 'forecasts' / '*.csv'))])
    assert list(back.columns) == ['created_timestamp', 'series_id', 'dim_id', 'forecast_date',
                                  'forecast_timestamp', 'forecast_quantity']
    assert len(back) == 80
    # the batched job entry points give the same rows as the per-group UDF calls
    both = pm.ProphetModeler.model(None, mconfig)
    assert len(both) == 2
    # the two command-line drivers (the reference's *_spark_driver.py without Spark)
    import yaml
    from time_series_spark_amd import modeler_driver, scorer_driver
    (tmp_path / 'm.yaml').write_text(yaml.safe_dump(mconfig))
    (tmp_path / 's.yaml').write_text(yaml.safe_dump(sconfig))
    assert modeler_driver.main(['x']) == 1                      # "arg1 must be the config YAML"
    assert modeler_driver.main(['x', str(tmp_path / 'm.yaml')]) == 0
    assert scorer_driver.main(['x', str(tmp_path / 's.yaml')]) == 0
    assert ps.ProphetScorer.score(None, sconfig) is None        # writes; returns nothing (prophet_scorer.py:152-165)
    conv = pd.concat([pd.read_csv(f) for f in glob.glob(str(tmp_path / 'forecasts' / '*.csv'))])
    assert len(conv) == 80 and np.array_equal(np.sort(conv['forecast_quantity'].values), np.sort(fdf['yhat'].values))


def test_jobs_as_pipelines_over_chunks_change_no_bit(env, tmp_path, capsys):
    """Round 6: ProphetModeler.model over ranges of partition directories (read | fit | parquet part, io.chunks) and
    ProphetScorer.score over the row groups of the model parts (read | predict | CSV parts) on the GPU: every model blob
    and every forecast row equal to the whole-input run's, for the reference's own model (ragged histories: two
    calendars, one series shorter) and for BASELINE cfg2's -- and the models' forecasts equal to the oracle's."""
    fc, cl = env
    from time_series_spark_amd import synth
    from time_series_spark_amd.jobs import prophet_modeler as pm, prophet_scorer as ps
    import glob
    N, T, H = 24, 400, 30
    for kind, prophet in (('reference', None), ('cfg2', {'growth': 'linear', 'seasonality_mode': 'additive', 'yearly_seasonality': True})):
        ds, y = synth.make_panel(N, T, 'logistic' if prophet is None else 'linear', seed=21)
        stamps = pd.DatetimeIndex(ds.astype('datetime64[ns]')).strftime('%Y-%m-%d %H:%M:%S').values
        root = tmp_path / kind / 'model-input'
        for n in range(N):
            d = root / ('series_id=%d' % (100 + n))
            d.mkdir(parents=True)
            t0 = 30 if n % 5 == 3 else 0                      # some series start a month later (their own calendar)
            (d / 'part-00000.csv').write_text('\n'.join('%d,%s,%d' % (1 + n % 2, s, v) for s, v in zip(stamps[t0:], y[n][t0:])) + '\n')
        out = {}
        for tag, chunks in (('whole', 1), ('chunked', 5)):
            mcfg = {'io': {'input': str(root), 'models': str(tmp_path / kind / ('models_' + tag)), 'chunks': chunks},
                    'model': {'floor': 0, 'cap_multiplier': 1.1, 'schedule_from_previous_models': False}}
            if prophet:
                mcfg['model']['prophet'] = prophet
            scfg = {'io': {'models': mcfg['io']['models'], 'forecasts': str(tmp_path / kind / ('fc_' + tag))},
                    'forecast': {'periods': H, 'frequency': 'D'}}
            assert pm.ProphetModeler.model(None, mcfg, return_frame=False) is None
            assert ps.ProphetScorer.score(None, scfg) is None
            models = pd.read_parquet(mcfg['io']['models']).sort_values(['series_id', 'dim_id']).reset_index(drop=True)
            fcs = pd.concat([pd.read_csv(f) for f in sorted(glob.glob(scfg['io']['forecasts'] + '/*.csv'))], ignore_index=True)
            fcs = fcs.sort_values(['series_id', 'dim_id', 'forecast_timestamp']).reset_index(drop=True)
            out[tag] = (models, fcs.drop(columns=['created_timestamp']))
            assert len(glob.glob(mcfg['io']['models'] + '/*.parquet')) == (5 if tag == 'chunked' else 1)
        assert len(out['whole'][0]) == N and len(out['whole'][1]) == N * H
        assert [bytes(b) for b in out['whole'][0]['model']] == [bytes(b) for b in out['chunked'][0]['model']]
        assert out['whole'][1].equals(out['chunked'][1])
        # ... and against the oracle: three series of the chunked run, one of them on the late calendar
        seas = fc.ModelSpec.auto_seasonalities(ds, seasonality_mode='multiplicative') if prophet is None else \
            [{'name': 'yearly', 'period': 365.25, 'fourier_order': 10}, {'name': 'weekly', 'period': 7, 'fourier_order': 3}]
        mode = 'multiplicative' if prophet is None else 'additive'
        csp = cl.make_spec(growth='logistic' if prophet is None else 'linear',
                           seasonalities=[(s['period'], s['fourier_order'], mode, 10.0) for s in seas],
                           eval_mode=int(prophet is not None))
        fut = ds[-1] + synth.DAY_NS * np.arange(1, H + 1)
        for n in (0, 3, 17):
            t0 = 30 if n % 5 == 3 else 0
            cap = float(np.float32(y[n][t0:].max() * 1.1))            # (the cap crosses the model frame as float32, :35-36)
            o = cl.fit(csp, ds[t0:], y[n][t0:].astype(np.float64), 0.0, y[n][t0:].max() * 1.1)
            yo, _ = cl.predict(csp, o, fut, 0.0, cap)
            got = out['chunked'][1].query('series_id == %d' % (100 + n))['forecast_quantity'].to_numpy()
            assert np.array_equal(got, np.maximum(np.trunc(yo), 0).astype(np.int64)), (kind, n)
    capsys.readouterr()


def test_permissive_input_mode_end_to_end(env, tmp_path):
    """io.input_mode: PERMISSIVE through the whole training job (round-3 advice: one malformed line used to come
    out as the key (series_id, dim_id = 0) and abort the run with 'less than 2 non-NaN rows', or merge into a real
    series 0): the reference's fixture with a bad date, a bad quantity, a short line and a bad dim_id spliced in
    trains the SAME two models as the clean file (the malformed records are dropped and counted, no other key
    appears); under FAILFAST the same input raises with file and line."""
    from time_series_spark_amd.jobs import prophet_modeler as pm
    g = np.load(helpers.GOLDEN + '/fixture_751.npz')
    clean = pd.DataFrame({'dim_id': g['raw_dim_id'],
                          'ds': pd.Series(g['raw_ds_ns'].astype('datetime64[ns]')).dt.strftime('%Y-%m-%d %H:%M:%S'),
                          'y': g['raw_y']}).to_csv(header=False, index=False).splitlines()
    bad = ['91,not-a-date,5', '155,2019-03-01 00:00:00,five', '91,2019-03-01 00:15:00', 'x,2019-03-01 00:30:00,7']
    dirty = clean[:100] + bad[:2] + clean[100:500] + bad[2:] + clean[500:]
    out = {}
    for tag, lines in (('clean', clean), ('dirty', dirty)):
        d = tmp_path / tag / 'model-input' / 'series_id=751'
        d.mkdir(parents=True)
        (d / 'part-0.csv').write_text('\n'.join(lines) + '\n')
        cfg = {'io': {'input': str(tmp_path / tag / 'model-input'), 'models': str(tmp_path / tag / 'models'),
                      'input_mode': 'PERMISSIVE'},
               'model': {'floor': 0, 'cap_multiplier': 1.1}}
        out[tag] = pm.ProphetModeler.model(None, cfg).sort_values(['series_id', 'dim_id']).reset_index(drop=True)
        if tag == 'dirty':
            with pytest.raises(ValueError, match=r'part-0.csv line 101 '):
                pm.ProphetModeler.model(None, dict(cfg, io=dict(cfg['io'], input_mode='FAILFAST')))
    a, b = out['clean'], out['dirty']
    assert len(a) == 2 and list(b['dim_id']) == list(a['dim_id']) == [91, 155] and (b['series_id'] == 751).all()
    assert [bytes(x) for x in a['model']] == [bytes(x) for x in b['model']]
    assert np.array_equal(a['cap'].values, b['cap'].values)


def test_holidays_through_the_job_layer(env):
    """SURVEY 8a U5 / BASELINE cfg4 shape through the reference-shaped functions: Prophet(holidays=...)
    via config['model']['prophet'], model_panel -> model blobs (which carry the holidays) ->
    forecast_panel, against the oracle fed with the LITERAL make_holiday_features columns."""
    fc, cl = env
    from time_series_spark_amd import synth
    from time_series_spark_amd.jobs import prophet_modeler as pm, prophet_scorer as ps
    from oracle.fbprophet_restated import ProphetOracle
    T, H = 400, 30
    ds, y = synth.make_panel(3, T, 'logistic', seed=12)
    days = pd.to_datetime(ds)
    hol = pd.DataFrame({'holiday': ['a'] * 3 + ['b'] * 2,
                        'ds': [days[40], days[200], days[-1] + pd.Timedelta(days=10), days[120], days[390]],
                        'lower_window': [-1] * 3 + [0] * 2, 'upper_window': [1] * 3 + [2] * 2})
    frames = [pd.DataFrame({'series_id': 7, 'dim_id': n, 'ds': days, 'y': y[n]}) for n in range(2)]
    frames.append(pd.DataFrame({'series_id': 7, 'dim_id': 2, 'ds': days[:350], 'y': y[2][:350]}))   # other grid
    df = pd.concat(frames, ignore_index=True)
    config = {'model': {'floor': 0, 'cap_multiplier': 1.1, 'prophet': {'holidays': hol, 'yearly_seasonality': False}},
              'forecast': {'periods': H, 'frequency': 'D'}}
    models = pm.model_panel(config)(df)
    assert len(models) == 3
    fcst = ps.forecast_panel(config)(models)
    assert len(fcst) == 3 * H
    m = ProphetOracle(holidays=hol)
    for n, Tn in ((0, T), (1, T), (2, 350)):
        dsn = ds[:Tn]
        fut = dsn[-1] + helpers.DAY_NS * np.arange(1, H + 1)
        cols, scales, _ = m.make_holiday_features(pd.Series(pd.to_datetime(dsn)), hol)
        colsf, _, _ = m.make_holiday_features(pd.Series(pd.to_datetime(fut)), hol)
        assert list(cols.columns) == sorted(cols.columns) and len(cols.columns) == 3 + 3
        csp = cl.make_spec(growth='logistic', seasonalities=[(7, 3, 'multiplicative', 10.0)],
                           extra=[('multiplicative', s) for s in scales])
        cap = float(np.float32(y[n][:Tn].max() * 1.1))          # the scorer reads cap back as float32
        o = cl.fit(csp, dsn, y[n][:Tn], 0.0, y[n][:Tn].max() * 1.1, cols.values.T)
        yo, _ = cl.predict(csp, o, fut, 0.0, cap, colsf.values.T)
        got = fcst[fcst['dim_id'] == n]
        assert np.array_equal(got['ds'].values.astype('datetime64[ns]').astype(np.int64), fut)
        assert np.array_equal(got['yhat'].values, np.maximum(np.trunc(yo), 0).astype(np.int32))
        assert colsf.values.any() or n == 2                     # the future window holds a holiday
    # opt-in interval columns (the reference drops them): same point forecast, lower < yhat < upper
    cfg_iv = dict(config, forecast=dict(config['forecast'], intervals=True, uncertainty_samples=300, seed=1))
    fi = ps.forecast_panel(cfg_iv)(models)
    assert list(fi.columns) == ['series_id', 'dim_id', 'ds', 'yhat', 'yhat_lower', 'yhat_upper']
    assert np.array_equal(fi['yhat'].values, fcst['yhat'].values)
    assert (fi['yhat_lower'] < fi['yhat'] + 1).all() and (fi['yhat_upper'] > fi['yhat']).all()
    fi2 = ps.forecast_panel(cfg_iv)(models.iloc[::-1].reset_index(drop=True))   # other batch order, same streams
    a = fi.sort_values(['dim_id', 'ds']).reset_index(drop=True)
    b = fi2.sort_values(['dim_id', 'ds']).reset_index(drop=True)
    assert np.array_equal(a['yhat_lower'].values, b['yhat_lower'].values)


@pytest.mark.parametrize('case', ['cfg2_linear_additive', 'ref_logistic_multiplicative', 'cfg4_holidays'])
def test_uncertainty_intervals_match_the_seeded_oracle(env, case):
    """yhat_lower / yhat_upper (Prophet.predict_uncertainty; computed and dropped by the
    reference, prophet_scorer.py:70, :86).  Parity is defined with a seeded counter-based generator
    (oracle cn_predict_intervals): the GPU kernels draw the same streams in the same order, so the
    percentiles are identical; the streams are keyed by series_key, so a series gets the same
    interval in any batch; and the interval behaves like one (contains the point forecast, ~80 %
    wide as 2 x 1.28 sigma plus the trend's uncertainty)."""
    fc, cl = env
    spec, ds, y, floor, cap, extra, fut, exf = helpers.make_case(case)
    csp = helpers.oracle_spec(spec)
    r = fc.fit_aligned(spec, ds, y, floor=floor, cap=cap, extra=extra)
    keys = np.array([1000 + 7 * n for n in range(len(y))], dtype=np.int64)
    yhat, lo, hi = fc.predict_intervals(spec, r.theta, r.y_scale, r.grid, fut, floor=floor, cap=cap,
                                        extra_future=exf, series_key=keys, uncertainty_samples=1000,
                                        interval_width=0.8, seed=42)
    assert np.array_equal(yhat, fc.predict(spec, r.theta, r.y_scale, r.grid, fut, floor=floor, cap=cap,
                                           extra_future=exf))
    for n in range(len(y)):
        o = cl.fit(csp, ds, y[n], floor[n], cap[n], extra)
        lo_o, hi_o = cl.predict_intervals(csp, o, fut, floor[n], cap[n], exf, n_samples=1000,
                                          interval_width=0.8, seed=42, series_key=int(keys[n]))
        assert np.array_equal(lo[n], lo_o) and np.array_equal(hi[n], hi_o), n
        sig = np.exp(o['theta'][2]) * o['info'].y_scale
        assert ((lo[n] < yhat[n]) & (yhat[n] < hi[n])).all()
        assert 2 * 1.2 * sig < (hi[n] - lo[n]).mean() < 2 * 1.2816 * sig * 1.6
    # batching does not matter: a sub-batch in another order gives the same intervals
    sub = np.array([4, 1])
    _, lo2, hi2 = fc.predict_intervals(spec, r.theta[sub], r.y_scale[sub], r.grid, fut, floor=floor[sub],
                                       cap=cap[sub], extra_future=exf, series_key=keys[sub], seed=42)
    assert np.array_equal(lo2, lo[sub]) and np.array_equal(hi2, hi[sub])
    # another seed gives other draws; 200 samples and a 95 % interval work too
    _, lo3, hi3 = fc.predict_intervals(spec, r.theta, r.y_scale, r.grid, fut, floor=floor, cap=cap,
                                       extra_future=exf, series_key=keys, uncertainty_samples=200,
                                       interval_width=0.95, seed=43)
    assert not np.array_equal(lo3, lo) and ((hi3 - lo3).mean(axis=1) > (hi - lo).mean(axis=1)).all()


def test_randomised_model_shapes_against_oracle(env):
    """A seeded sweep over model shapes the fixed cases do not enumerate: history length 8 .. 1 500
    rows (1 .. 24 rows per lane), 0 .. 30 changepoints, any subset of daily / weekly / yearly terms of
    random order (8-, 16-, 28- and 64-column kernels, one and two parameters per lane), both growths,
    both column modes, forced and automatic evaluation forms, both residual kernels, truncated and
    full runs: status, iteration and evaluation counts, objective and parameters bit for bit."""
    fc, cl = env
    from time_series_spark_amd import _lib, synth
    rng = np.random.default_rng(20260923)
    n_cases = 0
    for trial in range(36):
        growth = 'logistic' if rng.random() < 0.5 else 'linear'
        mode = 'multiplicative' if rng.random() < 0.5 else 'additive'
        T = int(rng.choice([8, 30, 64, 65, 99, 128, 200, 365, 513, 730, 1100, 1500]))
        seas = []
        if rng.random() < 0.8:
            seas.append({'name': 'weekly', 'period': 7, 'fourier_order': int(rng.integers(1, 4))})
        if T >= 200 and rng.random() < 0.6:
            seas.append({'name': 'yearly', 'period': 365.25, 'fourier_order': int(rng.integers(1, 13))})
        if rng.random() < 0.25:
            seas.append({'name': 'monthly', 'period': 30.5, 'fourier_order': int(rng.integers(1, 6))})
        if not seas:
            seas.append({'name': 'weekly', 'period': 7, 'fourier_order': 2})
        n_cp = int(rng.choice([0, 1, 3, 10, 25, 30]))
        if trial % 9 == 4:      # 40 design columns, 68 parameters: the two-parameters-per-lane kernels
            T = max(T, 365)
            seas = [{'name': 'weekly', 'period': 7, 'fourier_order': 3},
                    {'name': 'yearly', 'period': 365.25, 'fourier_order': 12},
                    {'name': 'monthly', 'period': 30.5, 'fourier_order': 5}]
            n_cp = 25
        kw = dict(growth=growth, seasonality_mode=mode, seasonalities=seas, n_changepoints=n_cp,
                  max_iter=int(rng.choice([5, 40, 10000])))
        K = 2 * sum(s['fourier_order'] for s in seas)
        linear_additive = growth == 'linear' and mode == 'additive'
        if linear_additive and rng.random() < 0.3:
            kw['eval_form'] = _lib.EVAL_RESIDUAL
        residual = not linear_additive or kw.get('eval_form') == _lib.EVAL_RESIDUAL
        if residual and K <= 28 and 3 + n_cp + K <= 64 and n_cp <= 28 and rng.random() < 0.4:
            kw['residual_kernel'] = _lib.RK_MFMA
        N = int(rng.integers(1, 5))
        ds, y = synth.make_panel(N, T, growth, seed=1000 + trial)
        fit_kw = {}
        if growth == 'logistic':
            fit_kw = dict(floor=np.zeros(N), cap=y.max(axis=1) * (1.05 + 0.5 * rng.random()))
        spec = fc.ModelSpec(**kw)
        try:
            r = fc.fit_aligned(spec, ds, y, **fit_kw)
        except _lib.TsfError as e:          # a shape the matrix-core kernel declines: the one-wave kernel takes it
            assert 'MFMA' in str(e), e
            kw.pop('residual_kernel')
            spec = fc.ModelSpec(**kw)
            r = fc.fit_aligned(spec, ds, y, **fit_kw)
        csp = helpers.oracle_spec(spec)
        for n in range(N):
            o = cl.fit(csp, ds, y[n], 0.0, fit_kw['cap'][n] if fit_kw else 0.0)
            S = o['info'].S
            ctx = (trial, n, kw, T)
            assert (r.status[n], r.n_iter[n], r.n_eval[n]) == (o['status'], o['n_iter'], o['n_eval']), ctx
            assert n_bit_diff(r.fval[n], o['f']) == 0, ctx
            assert n_bit_diff(r.theta[n][:3 + S], o['theta'][:3 + S]) == 0, ctx
            assert n_bit_diff(r.theta[n][3 + spec.n_changepoints:], o['theta'][3 + S:]) == 0, ctx
            n_cases += 1
        # the same series through the ragged entry point (own timestamps per series: drop a few rows)
        if trial % 3 == 0 and T >= 30:
            keep = [np.sort(rng.choice(T, size=T - int(rng.integers(0, 4)), replace=False)) for _ in range(N)]
            offs = np.concatenate([[0], np.cumsum([len(k) for k in keep])]).astype(np.int64)
            dsr = np.concatenate([ds[k] for k in keep])
            yr = np.concatenate([y[n][k] for n, k in enumerate(keep)])
            kwr = {k: v for k, v in kw.items() if k != 'residual_kernel'}
            specr = fc.ModelSpec(**kwr)
            rr = fc.fit_ragged(specr, offs, dsr, yr, **fit_kw)
            cspr = helpers.oracle_spec(specr)
            for n in range(N):
                o = cl.fit(cspr, ds[keep[n]], y[n][keep[n]], 0.0, fit_kw['cap'][n] if fit_kw else 0.0)
                S = o['info'].S
                assert (rr.status[n], rr.n_iter[n], rr.n_eval[n]) == (o['status'], o['n_iter'], o['n_eval']), (trial, n, 'ragged')
                assert n_bit_diff(rr.theta[n][:3 + S], o['theta'][:3 + S]) == 0, (trial, n, 'ragged')
                n_cases += 1
    assert n_cases >= 80


def test_randomised_newton_shapes_against_oracle(env):
    """The same for Stan's Newton (fbprophet's optimiser below 100 rows): 12 .. 99 rows, 0 .. 25
    changepoints (3 .. 46 parameters: eigen-problems of every size), weekly terms of any order, both
    growths and modes (quadratic-form and residual-form kernels), aligned and ragged."""
    fc, cl = env
    from time_series_spark_amd import _lib, synth
    rng = np.random.default_rng(90)
    n_cases = 0
    for trial in range(14):
        growth = 'logistic' if trial % 3 == 1 else 'linear'
        mode = 'multiplicative' if trial % 4 == 2 else 'additive'
        T = int(rng.choice([12, 25, 40, 60, 77, 99]))
        seas = [{'name': 'weekly', 'period': 7, 'fourier_order': int(rng.integers(1, 4))}]
        if rng.random() < 0.3:
            seas.append({'name': 'monthly', 'period': 30.5, 'fourier_order': int(rng.integers(1, 4))})
        n_cp = int(rng.choice([0, 2, 8, 25]))
        spec = fc.ModelSpec(growth=growth, seasonality_mode=mode, seasonalities=seas, n_changepoints=n_cp,
                            algorithm=_lib.ALGO_NEWTON, max_iter=int(rng.choice([3, 40, 10000])))
        N = int(rng.integers(1, 4))
        ds, y = synth.make_panel(N, T, growth, seed=2000 + trial)
        floor, cap = np.zeros(N), y.max(axis=1) * 1.15
        res = fc.fit_aligned(spec, ds, y, floor=floor, cap=cap)
        csp = helpers.oracle_spec(spec)
        for n in range(N):
            o = cl.fit_newton(csp, ds, y[n], floor[n], cap[n])
            S = o['info'].S
            ctx = (trial, n, growth, mode, T, n_cp)
            assert (res.status[n], res.n_iter[n], res.n_eval[n]) == (o['status'], o['n_iter'], o['n_eval']), ctx
            assert n_bit_diff(res.theta[n][:3 + S], o['theta'][:3 + S]) == 0, ctx
            assert n_bit_diff(res.theta[n][3 + spec.n_changepoints:], o['theta'][3 + S:]) == 0, ctx
            assert res.fval[n] == o['f'], ctx
            n_cases += 1
        if trial % 2 == 0 and T >= 25:      # ragged: truncated copies, each its own grid
            cut = [T - int(rng.integers(0, 9)) for _ in range(N)]
            off = np.concatenate([[0], np.cumsum(cut)]).astype(np.int64)
            rr = fc.fit_ragged(spec, off, np.concatenate([ds[:c] for c in cut]),
                               np.concatenate([y[i][:c] for i, c in enumerate(cut)]), floor=floor, cap=cap)
            for n in range(N):
                o = cl.fit_newton(csp, ds[:cut[n]], y[n][:cut[n]], floor[n], cap[n])
                S = o['info'].S
                assert (rr.status[n], rr.n_iter[n], rr.n_eval[n]) == (o['status'], o['n_iter'], o['n_eval']), (trial, n, 'ragged')
                assert n_bit_diff(rr.theta[n][:3 + S], o['theta'][:3 + S]) == 0, (trial, n, 'ragged')
                n_cases += 1
    assert n_cases >= 30


def test_stream_ordered_allocator_probe(env):
    """tools/probes/mallocasync_probe.hip (no library code; profiles/r05_mallocasync/README.md): blocks of 0.05 .. 6 GB, a
    kernel that writes and re-reads them in rounds, a verify kernel.  On plain hipMalloc / hipFree blocks (mode 16, the
    control: what the library does) the program must be clean -- the kernels and the checks themselves are sound.  On
    hipMallocAsync / hipFreeAsync blocks ROCm 7.2.0 corrupts the running kernel's multi-gigabyte blocks (intermittently;
    rarer with the pool's release threshold raised), which is why the library allocates no stream-ordered scratch: those
    runs are committed under profiles/r05_mallocasync/, this test keeps the control honest."""
    import shutil
    import subprocess
    src = os.path.join(helpers.ROOT, 'tools', 'probes', 'mallocasync_probe.hip')
    exe = os.path.join(helpers.ROOT, 'tools', 'probes', 'bin', 'mallocasync_probe')
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-w', src, '-o', exe])
    control = subprocess.run([exe, '16'], capture_output=True, text=True, timeout=600)
    assert control.returncode == 0 and 'MALLOCASYNC_PROBE_CLEAN' in control.stdout, control.stdout[-2000:]
    # (the stream-ordered modes are NOT run here: a program known to corrupt its own memory has no place in a suite
    # other jobs share a GPU with; tools/dev/async_scratch_hunt.sh and tools/dev/r05_f.sh run them, results under
    # profiles/r05_mallocasync/)


def test_baseline_config_1_every_series_against_the_oracle(env):
    """BASELINE config 1 -- 100 synthetic daily series x 365 points with the reference's own settings (logistic growth,
    floor 0, cap = 1.1 max y, multiplicative seasonality, fbprophet's auto rule: 364 days -> weekly only) -- is timed
    by bench.py (`other_baseline_configs.cfg1`) and had no test of its own (round-4 review): all 100 series, fit and
    90-step forecast, bit for bit against the oracle; the route is the base-pair kernel (HARM_W3, 8 columns) with the
    cooperative tail, and the table kernel must give the same bits."""
    fc, cl = env
    from time_series_spark_amd import synth
    N, T, H = 100, 365, 90
    ds, y = synth.make_panel(N, T, 'logistic', seed=751)
    seas = fc.ModelSpec.auto_seasonalities(ds, seasonality_mode='multiplicative')
    assert [s['name'] for s in seas] == ['weekly']
    spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=seas)
    floor, cap = np.zeros(N), y.max(axis=1) * 1.1
    fut = ds[-1] + helpers.DAY_NS * np.arange(1, H + 1)
    r = fc.fit_aligned(spec, ds, y, floor=floor, cap=cap)
    yhat = fc.predict(spec, r.theta, r.y_scale, r.grid, fut, floor=floor, cap=cap)
    csp = helpers.oracle_spec(spec)
    for n in range(N):
        o = cl.fit(csp, ds, y[n], 0.0, cap[n])
        assert (r.status[n], r.n_iter[n], r.n_eval[n]) == (o['status'], o['n_iter'], o['n_eval']), n
        assert n_bit_diff(r.theta[n], o['theta']) == 0 and n_bit_diff(r.fval[n], o['f']) == 0, n
        yo, _ = cl.predict(csp, o, fut, 0.0, cap[n])
        assert n_bit_diff(yhat[n], yo) == 0, n
    with fc.get_context().options(harm=0):
        r2 = fc.fit_aligned(spec, ds, y, floor=floor, cap=cap)
    for name in ('theta', 'fval', 'n_iter', 'n_eval', 'status'):
        assert np.array_equal(getattr(r, name), getattr(r2, name)), name
