"""ctypes binding of libtsf_amd.so (C-ABI declared in include/tsf.h).

This is the only place Python touches the native library.  There is NO CPU fallback: if the
shared library is missing or no GPU is visible, the functions raise.
"""
import contextlib
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TSF_LIB_PATH') or os.path.join(_HERE, 'libtsf_amd.so')   # override: A/B runs of two builds

MAX_SEAS = 8
MAX_EXTRA = 64
MAX_S = 60
MAX_K = 64
MAX_P = 128

GROWTH_LINEAR, GROWTH_LOGISTIC = 0, 1
MODE_ADDITIVE, MODE_MULTIPLICATIVE = 0, 1
Y_F64, Y_F32, Y_I32 = 0, 1, 2
EVAL_AUTO, EVAL_RESIDUAL, EVAL_QUADRATIC = 0, 1, 2
ALGO_LBFGS, ALGO_NEWTON, ALGO_AUTO = 0, 1, 2
RK_AUTO, RK_WAVE, RK_MFMA, RK_COOP = 0, 1, 2, 3
CONVERGE_STAN, CONVERGE_MAP = 0, 1

ST_ABSX, ST_ABSF, ST_RELF, ST_ABSGRAD, ST_RELGRAD, ST_MAXIT = 10, 20, 21, 30, 31, 40
ST_CONSTANT, ST_LSFAIL, ST_INIT_NONFINITE, ST_TOO_FEW, ST_CAP = 50, -1, -2, -10, -11
ST_EVAL_LIMIT = -3
ST_NEWTON_CONVERGED, ST_NEWTON_FAIL = 60, -4
ST_MAP_KKT, ST_MAP_FTOL, ST_MAP_MAXIT, ST_MAP_LS = 70, 71, 72, 73
STATUS_NAMES = {10: 'ABSX', 20: 'ABSF', 21: 'RELF', 30: 'ABSGRAD', 31: 'RELGRAD', 40: 'MAXIT',
                50: 'CONSTANT', -1: 'LSFAIL', -2: 'INIT_NONFINITE', -3: 'EVAL_LIMIT', -10: 'TOO_FEW',
                -11: 'CAP', 60: 'NEWTON_CONVERGED', -4: 'NEWTON_FAIL', 70: 'MAP_KKT', 71: 'MAP_FTOL', 72: 'MAP_MAXIT',
                73: 'MAP_LS'}


class TsfSpec(ctypes.Structure):
    """tsf_spec (include/tsf.h)."""
    _fields_ = [('growth', ctypes.c_int32), ('n_changepoints', ctypes.c_int32),
                ('changepoint_range', ctypes.c_double),
                ('changepoint_prior_scale', ctypes.c_double),
                ('n_seas', ctypes.c_int32), ('n_extra', ctypes.c_int32),
                ('seas_period', ctypes.c_double * MAX_SEAS),
                ('seas_prior_scale', ctypes.c_double * MAX_SEAS),
                ('seas_order', ctypes.c_int32 * MAX_SEAS),
                ('seas_mode', ctypes.c_int32 * MAX_SEAS),
                ('extra_prior_scale', ctypes.c_double * MAX_EXTRA),
                ('extra_mode', ctypes.c_int32 * MAX_EXTRA),
                ('max_iter', ctypes.c_int32), ('history', ctypes.c_int32),
                ('init_alpha', ctypes.c_double), ('tol_obj', ctypes.c_double),
                ('tol_rel_obj', ctypes.c_double), ('tol_grad', ctypes.c_double),
                ('tol_rel_grad', ctypes.c_double), ('tol_param', ctypes.c_double),
                ('eval_form', ctypes.c_int32), ('recenter_every', ctypes.c_int32),
                ('recenter_ratio', ctypes.c_double),
                ('algorithm', ctypes.c_int32), ('residual_kernel', ctypes.c_int32),
                ('coop_after', ctypes.c_int32), ('converge', ctypes.c_int32), ('map_max_iter', ctypes.c_int32),
                ('map_tol', ctypes.c_double)]


class TsfGridInfo(ctypes.Structure):
    """tsf_grid_info (include/tsf.h)."""
    _fields_ = [('start_ns', ctypes.c_int64), ('t_scale_ns', ctypes.c_int64),
                ('T', ctypes.c_int32), ('S', ctypes.c_int32), ('i1', ctypes.c_int32),
                ('NT', ctypes.c_int32), ('t_change', ctypes.c_double * (MAX_S + 4))]


GRID_DTYPE = np.dtype([('start_ns', '<i8'), ('t_scale_ns', '<i8'), ('T', '<i4'), ('S', '<i4'),
                       ('i1', '<i4'), ('NT', '<i4'), ('t_change', '<f8', (MAX_S + 4,))])


class TsfFitOut(ctypes.Structure):
    """tsf_fit_out (include/tsf.h)."""
    _fields_ = [('theta', ctypes.c_void_p), ('y_scale', ctypes.c_void_p),
                ('fval', ctypes.c_void_p), ('status', ctypes.c_void_p),
                ('n_iter', ctypes.c_void_p), ('n_eval', ctypes.c_void_p),
                ('grid', ctypes.c_void_p)]


# every symbol include/tsf.h declares; tests check the library exports all of them
EXPORTS = ['tsf_create', 'tsf_destroy', 'tsf_last_error', 'tsf_device_count', 'tsf_spec_default',
           'tsf_spec_size', 'tsf_grid_info_size', 'tsf_spec_K', 'tsf_theta_stride',
           'tsf_fit_aligned', 'tsf_fit_aligned_dev', 'tsf_fit_ragged', 'tsf_fit_ragged_dev',
           'tsf_predict', 'tsf_predict_dev', 'tsf_predict_intervals', 'tsf_predict_intervals_dev', 'tsf_eval', 'tsf_eval_quadratic', 'tsf_design', 'tsf_selftest_math',
           'tsf_set_option', 'tsf_get_option', 'tsf_set_cost_hints', 'tsf_set_profiling', 'tsf_profile_read', 'tsf_last_fit_kernel_ms', 'tsf_last_fit_route',
           'tsf_pack_rows', 'tsf_pack_rows_typed', 'tsf_pack_fetch', 'tsf_pack_flags', 'tsf_pack_free', 'tsf_model_blobs',
           'tsf_csv_read', 'tsf_csv_fetch', 'tsf_csv_columns', 'tsf_csv_malformed', 'tsf_csv_free', 'tsf_csv_write_forecasts', 'tsf_csv_write_forecasts_i32',
           'tsf_csv_discover', 'tsf_csv_discover_load', 'tsf_csv_root_open', 'tsf_csv_root_load', 'tsf_csv_root_free', 'tsf_csv_read_loaded', 'tsf_csv_dir_paths', 'tsf_csv_dir_series_id', 'tsf_csv_dir_error_path', 'tsf_csv_dir_free']

CSV_E_OPEN, CSV_E_PARSE, CSV_E_CODEC = -10, -11, -12          # TSF_CSV_E_* (include/tsf.h)

# TSF_OPT_* (include/tsf.h): route switches of one context
OPTIONS = ['harm', 'lattice', 'sparse_extra', 'fit_grouped', 'gram_share', 'grid_order', 'grid_share', 'ragged_split',
           'quad_reg', 'quad_m2_lds', 'quad_w4', 'quad_rreg', 'newton_batch', 'newton_flags', 'newton_ns', 'newton_lcap',
           'newton_fill', 'quad_raw_y', 'debug_async_scratch', 'coop_tail', 'map_direct']

_lib = None


class TsfError(RuntimeError):
    pass


def load():
    """Load libtsf_amd.so (once).  Raises if it has not been built -- no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TsfError('libtsf_amd.so not built: run `python -m time_series_spark_amd.build` '
                       '(or __graft_entry__.build()).  There is no CPU fallback.')
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    psp = ctypes.POINTER(TsfSpec)
    L.tsf_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.tsf_destroy.argtypes = [vp]
    L.tsf_destroy.restype = None
    L.tsf_last_error.argtypes = [vp]
    L.tsf_last_error.restype = ctypes.c_char_p
    L.tsf_device_count.argtypes = []
    L.tsf_spec_default.argtypes = [psp]
    L.tsf_spec_default.restype = None
    L.tsf_spec_K.argtypes = [psp]
    L.tsf_theta_stride.argtypes = [psp]
    L.tsf_fit_aligned.argtypes = [vp, psp, i64, i32, vp, vp, i32, vp, vp, vp,
                                  ctypes.POINTER(TsfFitOut)]
    L.tsf_fit_aligned_dev.argtypes = [vp, psp, i64, i32, vp, vp, i32, vp, vp, vp,
                                      ctypes.POINTER(TsfFitOut), vp]
    L.tsf_fit_ragged.argtypes = [vp, psp, i64, vp, vp, vp, i32, vp, vp, vp,
                                 ctypes.POINTER(TsfFitOut)]
    L.tsf_fit_ragged_dev.argtypes = [vp, psp, i64, vp, i64, i32, vp, vp, i32, vp, vp, vp,
                                     ctypes.POINTER(TsfFitOut), vp]
    L.tsf_predict.argtypes = [vp, psp, i64, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp]
    L.tsf_predict_dev.argtypes = [vp, psp, i64, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp,
                                  vp]
    f64, u64 = ctypes.c_double, ctypes.c_uint64
    L.tsf_predict_intervals.argtypes = [vp, psp, i64, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, i32, f64,
                                        u64, vp, vp, vp]
    L.tsf_predict_intervals_dev.argtypes = L.tsf_predict_intervals.argtypes + [vp]
    L.tsf_eval.argtypes = [vp, psp, i64, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.tsf_eval_quadratic.argtypes = [vp, psp, i64, i32, vp, vp, i32, vp, vp, vp, vp, vp]
    L.tsf_design.argtypes = [vp, psp, i32, vp, vp, vp, vp, vp]
    L.tsf_selftest_math.argtypes = [vp, i32, i64, vp, vp, vp]
    L.tsf_set_cost_hints.argtypes = [vp, vp, i64]
    L.tsf_set_option.argtypes = [vp, i32, i32]
    L.tsf_get_option.argtypes = [vp, i32]
    L.tsf_set_profiling.argtypes = [vp, i32]
    L.tsf_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_float), i32, ctypes.POINTER(i32)]
    L.tsf_last_fit_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.tsf_last_fit_route.argtypes = [vp, ctypes.POINTER(ctypes.c_int32)]
    L.tsf_pack_rows.argtypes = [i64, vp, vp, vp, vp, i32, ctypes.POINTER(vp), ctypes.POINTER(i64),
                                ctypes.POINTER(i64), ctypes.POINTER(i32)]
    L.tsf_pack_fetch.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.tsf_pack_rows_typed.argtypes = [i64, vp, vp, i32, vp, vp, i32, i32, ctypes.POINTER(vp), ctypes.POINTER(i64),
                                      ctypes.POINTER(i64), ctypes.POINTER(i32)]
    L.tsf_pack_flags.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.tsf_pack_free.argtypes = [vp]
    L.tsf_pack_free.restype = None
    L.tsf_model_blobs.argtypes = [i64, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, i32]
    L.tsf_csv_read.argtypes = [i32, vp, vp, ctypes.c_char_p, i32,
                               ctypes.POINTER(vp), ctypes.POINTER(i64), ctypes.POINTER(i32),
                               ctypes.POINTER(i64)]
    L.tsf_csv_fetch.argtypes = [vp, vp, vp, vp, vp]
    L.tsf_csv_columns.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.tsf_csv_discover.argtypes = [ctypes.c_char_p, i32, ctypes.POINTER(vp), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.tsf_csv_discover_load.argtypes = L.tsf_csv_discover.argtypes
    L.tsf_csv_read_loaded.argtypes = [vp, i32, i32, ctypes.c_char_p, i32, ctypes.POINTER(vp), ctypes.POINTER(i64),
                                      ctypes.POINTER(i32), ctypes.POINTER(i64)]
    L.tsf_csv_root_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.tsf_csv_root_load.argtypes = [vp, i32, i32, i32, ctypes.POINTER(vp), ctypes.POINTER(i32), ctypes.POINTER(i32),
                                    ctypes.POINTER(i32)]
    L.tsf_csv_root_free.argtypes = [vp]
    L.tsf_csv_root_free.restype = None
    L.tsf_csv_dir_paths.argtypes = [vp]
    L.tsf_csv_dir_paths.restype = vp
    L.tsf_csv_dir_series_id.argtypes = [vp]
    L.tsf_csv_dir_series_id.restype = vp
    L.tsf_csv_dir_error_path.argtypes = [vp]
    L.tsf_csv_dir_error_path.restype = ctypes.c_char_p
    L.tsf_csv_dir_free.argtypes = [vp]
    L.tsf_csv_dir_free.restype = None
    L.tsf_csv_malformed.argtypes = [vp]
    L.tsf_csv_malformed.restype = ctypes.c_int64
    L.tsf_csv_free.argtypes = [vp]
    L.tsf_csv_free.restype = None
    L.tsf_csv_write_forecasts.argtypes = [ctypes.c_char_p, ctypes.c_char_p, i64, vp, vp, vp, vp, i32]
    L.tsf_csv_write_forecasts_i32.argtypes = [ctypes.c_char_p, ctypes.c_char_p, i64, vp, vp, vp, vp, i32]
    if L.tsf_spec_size() != ctypes.sizeof(TsfSpec):
        raise TsfError('tsf_spec layout mismatch between _lib.py and libtsf_amd.so')
    if L.tsf_grid_info_size() != ctypes.sizeof(TsfGridInfo) or GRID_DTYPE.itemsize != ctypes.sizeof(TsfGridInfo):
        raise TsfError('tsf_grid_info layout mismatch between _lib.py and libtsf_amd.so')
    _lib = L
    return L


class Context(object):
    """One tsf_ctx: bound to one GPU, not re-entrant."""

    def __init__(self, device=0):
        L = load()
        h = ctypes.c_void_p()
        rc = L.tsf_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise TsfError('tsf_create(device=%d) failed (rc=%d): no usable MI355X visible. '
                           'This library has no CPU fallback.' % (device, rc))
        self._h = h
        self.device = int(device)
        # measurement tools: TSF_OPTIONS="harm=0,sparse_extra=0" sets route switches on every context this PROCESS
        # creates (read here, in the Python host layer; the library itself reads no environment variable for routes)
        for item in filter(None, os.environ.get('TSF_OPTIONS', '').split(',')):
            k, v = item.split('=')
            self.set_option(k.strip(), int(v))

    def close(self):
        if getattr(self, '_h', None):
            load().tsf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            msg = load().tsf_last_error(self._h)
            raise TsfError('libtsf_amd: rc=%d: %s' % (rc, msg.decode() if msg else '?'))

    @property
    def handle(self):
        return self._h

    def set_option(self, name, value=-1):
        """tsf_set_option: a route switch of THIS context (name: one of OPTIONS; -1 / None = the library's default).
        Results never depend on a route; tests and measurements use this."""
        self.check(load().tsf_set_option(self._h, OPTIONS.index(name), -1 if value is None else int(value)))

    def get_option(self, name):
        return int(load().tsf_get_option(self._h, OPTIONS.index(name)))

    @contextlib.contextmanager
    def options(self, **kw):
        """with ctx.options(harm=0, sparse_extra=0): ... -- set, run, restore."""
        old = {k: self.get_option(k) for k in kw}
        try:
            for k, v in kw.items():
                self.set_option(k, v)
            yield self
        finally:
            for k, v in old.items():
                self.set_option(k, v)


def default_spec():
    s = TsfSpec()
    load().tsf_spec_default(ctypes.byref(s))
    return s


def _ptr(a):
    return None if a is None else a.ctypes.data


def y_dtype_code(y):
    if y.dtype == np.float64:
        return Y_F64
    if y.dtype == np.float32:
        return Y_F32
    if y.dtype == np.int32:
        return Y_I32
    raise TypeError('y must be float64, float32 or int32')
