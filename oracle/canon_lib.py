"""ctypes loader for oracle/prophet_canon.c -- TEST INFRASTRUCTURE ONLY.

The canonical-arithmetic CPU oracle (see the header of prophet_canon.c).  Built with
``gcc -O2 -ffp-contract=off -mfma`` so that every ``fma()`` is one hardware FMA and nothing
else is fused or reassociated.  PARITY UNPINNED w.r.t. real fbprophet output, see
oracle/fbprophet_restated.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, 'prophet_canon.c')
_HDR = os.path.join(_HERE, 'det_math.h')
_OUT_DIR = os.path.join(_HERE, '_build')
_SO = os.path.join(_OUT_DIR, 'liboracle_canon.so')

MAX_SEAS = 8
MAX_EXTRA = 64

STATUS_NAMES = {0: 'SUCCESS', 10: 'ABSX', 20: 'ABSF', 21: 'RELF', 30: 'ABSGRAD', 31: 'RELGRAD',
                40: 'MAXIT', -1: 'LSFAIL', -2: 'INIT_NONFINITE', -3: 'EVAL_LIMIT', 50: 'CONSTANT',
                -10: 'ERR_TOO_FEW', -11: 'ERR_CAP', -12: 'ERR_SIZE',
                60: 'NEWTON_CONVERGED', -4: 'NEWTON_FAIL', -13: 'NEWTON_TOO_WIDE'}


class CnSpec(ctypes.Structure):
    _fields_ = [('growth', ctypes.c_int32), ('n_changepoints', ctypes.c_int32),
                ('changepoint_range', ctypes.c_double), ('tau', ctypes.c_double),
                ('n_seas', ctypes.c_int32), ('n_extra', ctypes.c_int32),
                ('seas_period', ctypes.c_double * MAX_SEAS),
                ('seas_prior', ctypes.c_double * MAX_SEAS),
                ('seas_order', ctypes.c_int32 * MAX_SEAS),
                ('seas_mode', ctypes.c_int32 * MAX_SEAS),
                ('extra_prior', ctypes.c_double * MAX_EXTRA),
                ('extra_mode', ctypes.c_int32 * MAX_EXTRA),
                ('max_iter', ctypes.c_int32), ('history', ctypes.c_int32),
                ('init_alpha', ctypes.c_double), ('tol_obj', ctypes.c_double),
                ('tol_rel_obj', ctypes.c_double), ('tol_grad', ctypes.c_double),
                ('tol_rel_grad', ctypes.c_double), ('tol_param', ctypes.c_double),
                ('eval_mode', ctypes.c_int32), ('recenter_every', ctypes.c_int32),
                ('recenter_ratio', ctypes.c_double)]


class CnFitInfo(ctypes.Structure):
    _fields_ = [('status', ctypes.c_int32), ('n_iter', ctypes.c_int32),
                ('n_eval', ctypes.c_int32), ('S', ctypes.c_int32), ('K', ctypes.c_int32),
                ('pad_', ctypes.c_int32), ('f', ctypes.c_double), ('y_scale', ctypes.c_double),
                ('floor_', ctypes.c_double), ('cap_scaled', ctypes.c_double),
                ('start_ns', ctypes.c_int64), ('t_scale_ns', ctypes.c_int64)]


def build(force=False):
    """Compile the canonical C oracle (the checker).  Called by __graft_entry__.build()."""
    newest = max(os.path.getmtime(_SRC), os.path.getmtime(_HDR))
    if not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= newest:
        return _SO
    os.makedirs(_OUT_DIR, exist_ok=True)
    cmd = ['gcc', '-O2', '-ffp-contract=off', '-mfma', '-fPIC', '-shared', '-std=gnu11',
           '-o', _SO, _SRC, '-lm']
    subprocess.check_call(cmd)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        assert L.cn_spec_size() == ctypes.sizeof(CnSpec), 'cn_spec layout mismatch'
        vp, i32, f64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_double
        L.cn_default_spec.argtypes = [ctypes.POINTER(CnSpec)]
        L.cn_default_spec.restype = None
        L.cn_design.argtypes = [ctypes.POINTER(CnSpec), i32, vp, vp, f64, f64, vp, vp, vp, vp, vp,
                                vp, ctypes.POINTER(CnFitInfo)]
        L.cn_eval_at.argtypes = [ctypes.POINTER(CnSpec), i32, vp, vp, f64, f64, vp, vp,
                                 ctypes.POINTER(f64), vp]
        L.cn_fit.argtypes = [ctypes.POINTER(CnSpec), i32, vp, vp, f64, f64, vp, vp, vp,
                             ctypes.POINTER(CnFitInfo)]
        L.cn_fit_newton.argtypes = L.cn_fit.argtypes
        L.cn_jacobi_eigh.argtypes = [i32, vp, vp, vp]
        L.cn_ql_eigh.argtypes = [i32, vp, vp, vp]
        L.cn_fit_checked.argtypes = [ctypes.POINTER(CnSpec), i32, vp, vp, f64, f64, vp, vp,
                                     ctypes.POINTER(CnFitInfo), vp]
        L.cn_predict.argtypes = [ctypes.POINTER(CnSpec), ctypes.POINTER(CnFitInfo), vp, vp, i32,
                                 vp, f64, f64, vp, vp, vp]
        L.cn_predict_intervals.argtypes = [ctypes.POINTER(CnSpec), ctypes.POINTER(CnFitInfo), vp, vp, i32,
                                           vp, f64, f64, vp, i32, f64, ctypes.c_uint64, ctypes.c_uint64, vp, vp]
        L.cn_det_exp.argtypes = [f64]
        L.cn_det_exp.restype = f64
        L.cn_det_log.argtypes = [f64]
        L.cn_det_log.restype = f64
        L.cn_det_sincos.argtypes = [f64, ctypes.POINTER(f64), ctypes.POINTER(f64)]
        L.cn_det_sincos.restype = None
        _lib = L
    return _lib


def make_spec(growth='linear', n_changepoints=25, changepoint_range=0.8,
              changepoint_prior_scale=0.05, seasonalities=(), extra=(), **opt):
    """seasonalities: iterable of (period, fourier_order, mode, prior_scale);
    extra: iterable of (mode, prior_scale) for explicit columns; mode in
    {'additive','multiplicative'}.  opt: L-BFGS options (max_iter, history, init_alpha,
    tol_obj, tol_rel_obj, tol_grad, tol_rel_grad, tol_param)."""
    sp = CnSpec()
    lib().cn_default_spec(ctypes.byref(sp))
    sp.growth = {'linear': 0, 'logistic': 1}[growth]
    sp.n_changepoints = int(n_changepoints)
    sp.changepoint_range = float(changepoint_range)
    sp.tau = float(changepoint_prior_scale)
    seasonalities = list(seasonalities)
    extra = list(extra)
    if len(seasonalities) > MAX_SEAS or len(extra) > MAX_EXTRA:
        raise ValueError('too many seasonalities / extra columns for the oracle')
    sp.n_seas = len(seasonalities)
    for i, (period, order, mode, prior) in enumerate(seasonalities):
        sp.seas_period[i] = float(period)
        sp.seas_order[i] = int(order)
        sp.seas_mode[i] = int(mode == 'multiplicative')
        sp.seas_prior[i] = float(prior)
    sp.n_extra = len(extra)
    for i, (mode, prior) in enumerate(extra):
        sp.extra_mode[i] = int(mode == 'multiplicative')
        sp.extra_prior[i] = float(prior)
    for k, v in opt.items():
        if k not in ('max_iter', 'history', 'init_alpha', 'tol_obj', 'tol_rel_obj', 'tol_grad',
                     'tol_rel_grad', 'tol_param', 'eval_mode', 'recenter_every',
                     'recenter_ratio'):
            raise TypeError('unknown option %r' % k)
        setattr(sp, k, v)
    return sp


def spec_K(sp):
    return sp.n_extra + 2 * sum(sp.seas_order[i] for i in range(sp.n_seas))


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _extra_ptr(sp, extra, n):
    if sp.n_extra == 0:
        return None, None
    e = _f64(extra)
    if e.shape != (sp.n_extra, n):
        raise ValueError('extra must be [n_extra][%d]' % n)
    return e, e.ctypes.data


def design(sp, ds_ns, y, floor=0.0, cap=0.0, extra=None):
    ds_ns, y = _i64(ds_ns), _f64(y)
    T = len(ds_ns)
    K = spec_K(sp)
    X = np.zeros((T, max(K, 1)))
    t = np.zeros(T)
    ysc = np.zeros(T)
    tch = np.zeros(64)
    init = np.zeros(2)
    info = CnFitInfo()
    ekeep, eptr = _extra_ptr(sp, extra, T)
    rc = lib().cn_design(ctypes.byref(sp), T, ds_ns.ctypes.data, y.ctypes.data, float(floor),
                         float(cap), eptr, X.ctypes.data, t.ctypes.data, ysc.ctypes.data,
                         tch.ctypes.data, init.ctypes.data, ctypes.byref(info))
    if rc != 0:
        raise ValueError('cn_design: %s' % STATUS_NAMES.get(rc, rc))
    return {'X': X[:, :K], 't': t, 'y_scaled': ysc, 't_change': tch[:info.S].copy(),
            'k0': init[0], 'm0': init[1], 'info': info}


def eval_at(sp, ds_ns, y, theta, floor=0.0, cap=0.0, extra=None):
    ds_ns, y, theta = _i64(ds_ns), _f64(y), _f64(theta)
    T = len(ds_ns)
    g = np.zeros(128)
    f = ctypes.c_double(0.0)
    ekeep, eptr = _extra_ptr(sp, extra, T)
    th = np.zeros(128)
    th[:len(theta)] = theta
    rc = lib().cn_eval_at(ctypes.byref(sp), T, ds_ns.ctypes.data, y.ctypes.data, float(floor),
                          float(cap), eptr, th.ctypes.data, ctypes.byref(f), g.ctypes.data)
    return f.value, g[:len(theta)].copy(), rc


def eval_quadratic_at(sp, ds_ns, y, theta_ref, theta, extra=None):
    """Quadratic-form evaluation (cn_eval_gram) at theta around the reference point theta_ref
    (cn_eval_quadratic_at): returns (f, gradient, rc)."""
    ds_ns, y = _i64(ds_ns), _f64(y)
    T = len(ds_ns)
    g = np.zeros(128)
    f = ctypes.c_double(0.0)
    ekeep, eptr = _extra_ptr(sp, extra, T)
    th, thr = np.zeros(128), np.zeros(128)
    th[:len(theta)] = theta
    thr[:len(theta_ref)] = theta_ref
    L = lib()
    L.cn_eval_quadratic_at.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]
    rc = L.cn_eval_quadratic_at(ctypes.byref(sp), T, ds_ns.ctypes.data, y.ctypes.data, eptr, thr.ctypes.data,
                                th.ctypes.data, ctypes.byref(f), g.ctypes.data)
    return f.value, g[:len(theta)].copy(), rc


def fit(sp, ds_ns, y, floor=0.0, cap=0.0, extra=None):
    """Returns dict(theta, t_change, info, status, n_iter, n_eval, f)."""
    ds_ns, y = _i64(ds_ns), _f64(y)
    T = len(ds_ns)
    theta = np.zeros(128)
    tch = np.zeros(64)
    info = CnFitInfo()
    ekeep, eptr = _extra_ptr(sp, extra, T)
    lib().cn_fit(ctypes.byref(sp), T, ds_ns.ctypes.data, y.ctypes.data, float(floor), float(cap),
                 eptr, theta.ctypes.data, tch.ctypes.data, ctypes.byref(info))
    P = 3 + info.S + info.K
    return {'theta': theta[:P].copy(), 't_change': tch[:info.S].copy(), 'info': info,
            'status': info.status, 'status_name': STATUS_NAMES.get(info.status, '?'),
            'n_iter': info.n_iter, 'n_eval': info.n_eval, 'n_resid': info.pad_, 'f': info.f}


def fit_newton(sp, ds_ns, y, floor=0.0, cap=0.0, extra=None):
    """cn_fit with Stan's Newton optimiser (what fbprophet uses for T < 100); same dict."""
    ds_ns, y = _i64(ds_ns), _f64(y)
    T = len(ds_ns)
    theta = np.zeros(128)
    tch = np.zeros(64)
    info = CnFitInfo()
    ekeep, eptr = _extra_ptr(sp, extra, T)
    lib().cn_fit_newton(ctypes.byref(sp), T, ds_ns.ctypes.data, y.ctypes.data, float(floor), float(cap),
                        eptr, theta.ctypes.data, tch.ctypes.data, ctypes.byref(info))
    P = 3 + info.S + info.K
    return {'theta': theta[:P].copy(), 't_change': tch[:info.S].copy(), 'info': info,
            'status': info.status, 'status_name': STATUS_NAMES.get(info.status, '?'),
            'n_iter': info.n_iter, 'n_eval': info.n_eval, 'f': info.f}


def jacobi_eigh(A):
    """The oracle's canonical symmetric eigen-solver: (eigenvalues, eigenvectors in columns,
    sweeps)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    V = np.zeros((n, n))
    lam = np.zeros(n)
    sweeps = lib().cn_jacobi_eigh(n, A.ctypes.data, V.ctypes.data, lam.ctypes.data)
    if sweeps < 0:
        raise ValueError('cn_jacobi_eigh: n out of range')
    return lam, V, sweeps


def ql_eigh(A):
    """The eigen-solver of the oracle's Newton restatement (Householder tridiagonalisation +
    implicit QL): (eigenvalues, eigenvectors in columns, QL iterations)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    V = np.zeros((n, n))
    lam = np.zeros(n)
    its = lib().cn_ql_eigh(n, A.ctypes.data, V.ctypes.data, lam.ctypes.data)
    if its < 0:
        raise ValueError('cn_ql_eigh: n out of range')
    return lam, V, its


def fit_checked(sp, ds_ns, y):
    """Quadratic-form fit (linear growth, additive columns) in which every evaluation is also
    done in residual form; returns (fit dict, max rel |df|, max rel |dg|_2)."""
    ds_ns, y = _i64(ds_ns), _f64(y)
    theta = np.zeros(128)
    info = CnFitInfo()
    chk = np.zeros(2)
    lib().cn_fit_checked(ctypes.byref(sp), len(ds_ns), ds_ns.ctypes.data, y.ctypes.data, 0.0, 0.0,
                         None, theta.ctypes.data, ctypes.byref(info), chk.ctypes.data)
    P = 3 + info.S + info.K
    return ({'theta': theta[:P].copy(), 'status': info.status, 'n_iter': info.n_iter,
             'n_eval': info.n_eval, 'n_resid': info.pad_, 'f': info.f}, chk[0], chk[1])


def predict(sp, fitres, ds_future_ns, floor=0.0, cap=0.0, extra_future=None):
    ds = _i64(ds_future_ns)
    H = len(ds)
    yhat = np.zeros(H)
    trend = np.zeros(H)
    theta = _f64(fitres['theta'])
    tch = _f64(np.concatenate([fitres['t_change'], [0.0]]))
    ekeep, eptr = _extra_ptr(sp, extra_future, H)
    lib().cn_predict(ctypes.byref(sp), ctypes.byref(fitres['info']), theta.ctypes.data,
                     tch.ctypes.data, H, ds.ctypes.data, float(floor), float(cap), eptr,
                     yhat.ctypes.data, trend.ctypes.data)
    return yhat, trend


def predict_intervals(sp, fitres, ds_future_ns, floor=0.0, cap=0.0, extra_future=None, n_samples=1000,
                      interval_width=0.8, seed=0, series_key=0):
    """Seeded restatement of Prophet.predict_uncertainty (yhat_lower, yhat_upper); see
    cn_predict_intervals."""
    ds = _i64(ds_future_ns)
    H = len(ds)
    lo, hi = np.zeros(H), np.zeros(H)
    theta = _f64(fitres['theta'])
    tch = _f64(np.concatenate([fitres['t_change'], [0.0]]))
    ekeep, eptr = _extra_ptr(sp, extra_future, H)
    lib().cn_predict_intervals(ctypes.byref(sp), ctypes.byref(fitres['info']), theta.ctypes.data,
                               tch.ctypes.data, H, ds.ctypes.data, float(floor), float(cap), eptr,
                               int(n_samples), float(interval_width), int(seed), int(series_key),
                               lo.ctypes.data, hi.ctypes.data)
    return lo, hi
