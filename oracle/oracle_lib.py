"""ctypes loader for oracle/stan_lbfgs.c -- TEST INFRASTRUCTURE ONLY.

Builds ``oracle/_build/liboracle.so`` with gcc on first use (``-O2 -ffp-contract=off`` so
the arithmetic is plain IEEE double, one rounding per operation).  PARITY UNPINNED, see
oracle/fbprophet_restated.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, 'stan_lbfgs.c')
_OUT_DIR = os.path.join(_HERE, '_build')
_SO = os.path.join(_OUT_DIR, 'liboracle.so')

TERM_NAMES = {0: 'SUCCESS', 10: 'ABSX', 20: 'ABSF', 21: 'RELF', 30: 'ABSGRAD', 31: 'RELGRAD',
              40: 'MAXIT', -1: 'LSFAIL', -2: 'INIT_NONFINITE'}


class OracleData(ctypes.Structure):
    _fields_ = [('T', ctypes.c_int32), ('K', ctypes.c_int32), ('S', ctypes.c_int32),
                ('growth', ctypes.c_int32),
                ('t', ctypes.c_void_p), ('y', ctypes.c_void_p), ('cap', ctypes.c_void_p),
                ('X', ctypes.c_void_p), ('s_a', ctypes.c_void_p), ('s_m', ctypes.c_void_p),
                ('sigmas', ctypes.c_void_p), ('t_change', ctypes.c_void_p),
                ('tau', ctypes.c_double)]


class OracleOpts(ctypes.Structure):
    _fields_ = [('max_iter', ctypes.c_int32), ('history', ctypes.c_int32),
                ('init_alpha', ctypes.c_double), ('tol_obj', ctypes.c_double),
                ('tol_rel_obj', ctypes.c_double), ('tol_grad', ctypes.c_double),
                ('tol_rel_grad', ctypes.c_double), ('tol_param', ctypes.c_double)]


class OracleResult(ctypes.Structure):
    _fields_ = [('status', ctypes.c_int32), ('n_iter', ctypes.c_int32),
                ('n_eval', ctypes.c_int32), ('f', ctypes.c_double)]


def build(force=False):
    """Compile the C restatement (the checker).  Called by __graft_entry__.build()."""
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= os.path.getmtime(_SRC)):
        return _SO
    os.makedirs(_OUT_DIR, exist_ok=True)
    cmd = ['gcc', '-O2', '-ffp-contract=off', '-fPIC', '-shared', '-std=c99', '-o', _SO, _SRC,
           '-lm']
    subprocess.check_call(cmd)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = build()
        _lib = ctypes.CDLL(so)
        _lib.oracle_fg.restype = ctypes.c_int
        _lib.oracle_fg.argtypes = [ctypes.POINTER(OracleData), ctypes.c_void_p,
                                   ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]
        _lib.oracle_lbfgs.restype = ctypes.c_int
        _lib.oracle_lbfgs.argtypes = [ctypes.POINTER(OracleData), ctypes.POINTER(OracleOpts),
                                      ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.POINTER(OracleResult)]
        _lib.oracle_default_opts.restype = None
        _lib.oracle_default_opts.argtypes = [ctypes.POINTER(OracleOpts)]
    return _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def pack(dat):
    """dict from ProphetOracle.stan_data -> (OracleData, keepalive list)."""
    keep = {k: _f64(dat[k]) for k in ('t', 'y', 'cap', 'X', 's_a', 's_m', 'sigmas', 't_change')}
    d = OracleData()
    d.T, d.K, d.S = int(dat['T']), int(dat['K']), int(dat['S'])
    d.growth = int(dat['trend_indicator'])
    for k, v in keep.items():
        setattr(d, k, v.ctypes.data)
    d.tau = float(dat['tau'])
    return d, keep


def default_opts(**over):
    o = OracleOpts()
    lib().oracle_default_opts(ctypes.byref(o))
    for k, v in over.items():
        if not hasattr(o, k):
            raise TypeError('unknown L-BFGS option %r' % k)
        setattr(o, k, v)
    return o


def neg_log_prob_grad(dat, theta):
    d, keep = pack(dat)
    theta = _f64(theta)
    g = np.zeros_like(theta)
    f = ctypes.c_double(0.0)
    rc = lib().oracle_fg(ctypes.byref(d), theta.ctypes.data, ctypes.byref(f), g.ctypes.data)
    return f.value, g, rc


def neg_log_prob_grad_packed(packed, theta):
    """neg_log_prob_grad on a `pack(dat)` result (callers that evaluate one series thousands of times)."""
    d, keep = packed
    theta = _f64(theta)
    g = np.zeros_like(theta)
    f = ctypes.c_double(0.0)
    rc = lib().oracle_fg(ctypes.byref(d), theta.ctypes.data, ctypes.byref(f), g.ctypes.data)
    return f.value, g, rc


def stan_lbfgs(dat, theta0, **opts):
    """Run the restated Stan L-BFGS from theta0; returns (theta, info)."""
    d, keep = pack(dat)
    o = default_opts(**opts)
    theta0 = _f64(theta0)
    out = np.zeros_like(theta0)
    res = OracleResult()
    rc = lib().oracle_lbfgs(ctypes.byref(d), ctypes.byref(o), theta0.ctypes.data,
                            out.ctypes.data, ctypes.byref(res))
    if rc != 0:
        raise MemoryError('oracle_lbfgs rc=%d' % rc)
    info = {'status': res.status, 'status_name': TERM_NAMES.get(res.status, '?'),
            'n_iter': res.n_iter, 'n_eval': res.n_eval, 'f': res.f}
    if res.status == -2:
        info['error'] = 'Rejecting initial value: non-finite log probability or gradient'
    return out, info
