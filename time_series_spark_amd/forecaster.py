"""Batched panel API over the C-ABI: fit_aligned / fit_ragged / predict on numpy arrays.

Stands where the reference calls fbprophet one series at a time
(/root/reference/src/jobs/prophet_modeler.py:65-66, /root/reference/src/jobs/prophet_scorer.py:70)
but takes the whole (series x timestamp x y) panel in one call.  The model options mirror
fbprophet 0.5's ``Prophet.__init__`` / ``set_auto_seasonalities`` (spec in SURVEY.md 8a
U1-U5); the arithmetic runs in libtsf_amd.so on the GPU -- this module only packs arrays.
"""
import ctypes
import os

import numpy as np

from . import _lib, parallel

DAY_NS = 86400 * 10 ** 9


class ModelSpec(object):
    """Model + optimiser settings shared by all series of a call.

    seasonalities: list of dicts {name, period (days), fourier_order, prior_scale, mode}.
    extra: list of dicts {name, prior_scale, mode} for explicit design columns (holiday
    indicators, regressors) whose values the caller supplies.
    """

    def __init__(self, growth='linear', seasonality_mode='additive', n_changepoints=25,
                 changepoint_range=0.8, changepoint_prior_scale=0.05,
                 seasonality_prior_scale=10.0, holidays_prior_scale=10.0,
                 seasonalities=None, extra=None, holidays=None, **lbfgs):
        if growth not in ('linear', 'logistic'):
            raise ValueError("Parameter 'growth' should be 'linear' or 'logistic'.")
        if seasonality_mode not in ('additive', 'multiplicative'):
            raise ValueError("seasonality_mode must be 'additive' or 'multiplicative'")
        if changepoint_range < 0 or changepoint_range > 1:
            raise ValueError("Parameter 'changepoint_range' must be in [0, 1]")
        self.growth = growth
        self.seasonality_mode = seasonality_mode
        self.n_changepoints = int(n_changepoints)
        self.changepoint_range = float(changepoint_range)
        self.changepoint_prior_scale = float(changepoint_prior_scale)
        self.seasonality_prior_scale = float(seasonality_prior_scale)
        self.holidays_prior_scale = float(holidays_prior_scale)
        self.seasonalities = [dict(s) for s in (seasonalities or [])]
        self.extra = [dict(e) for e in (extra or [])]
        # normalised holidays (features.normalize_holidays) whose indicator columns are the FIRST
        # len(features.holiday_columns(holidays)[0]) entries of `extra`: carried so that the scorer
        # can rebuild the columns for future dates (fbprophet keeps the holidays frame in the model)
        self.holidays = list(holidays) if holidays else None
        self.lbfgs = dict(lbfgs)
        for k in self.lbfgs:
            if k not in ('max_iter', 'history', 'init_alpha', 'tol_obj', 'tol_rel_obj',
                         'tol_grad', 'tol_rel_grad', 'tol_param', 'eval_form', 'recenter_every',
                         'recenter_ratio', 'algorithm', 'residual_kernel', 'coop_after', 'converge', 'map_max_iter', 'map_tol'):
                raise TypeError('unknown optimiser option %r' % k)

    # -- fbprophet set_auto_seasonalities on a timestamp vector --------------------------------
    @staticmethod
    def auto_seasonalities(ds_ns, yearly='auto', weekly='auto', daily='auto',
                           seasonality_mode='additive', seasonality_prior_scale=10.0,
                           user_seasonalities=()):
        """Returns the seasonality list fbprophet 0.5 would build for history timestamps
        ``ds_ns`` (int64 ns, sorted): yearly (365.25 d, order 10) off when span < 730 d;
        weekly (7 d, order 3) off when span < 14 d or the smallest non-zero spacing >= 7 d;
        daily (1 d, order 4) off when span < 2 d or the smallest spacing >= 1 d."""
        ds_ns = np.asarray(ds_ns, dtype=np.int64)
        first, last = int(ds_ns.min()), int(ds_ns.max())
        diffs = np.diff(np.sort(ds_ns))
        nz = diffs[diffs != 0]
        min_dt = int(nz.min()) if nz.size else -1
        return ModelSpec.auto_from_stats(last - first, min_dt, yearly, weekly, daily,
                                         seasonality_mode, seasonality_prior_scale,
                                         user_seasonalities)

    @staticmethod
    def auto_from_stats(span, min_dt, yearly='auto', weekly='auto', daily='auto',
                        seasonality_mode='additive', seasonality_prior_scale=10.0,
                        user_seasonalities=()):
        """Same rules from (span_ns, smallest non-zero spacing in ns or -1 if none)."""
        names = {s['name'] for s in user_seasonalities}

        def order(name, arg, auto_disable, default):
            if isinstance(arg, str) and arg == 'auto':
                if name in names or auto_disable:
                    return 0
                return default
            if arg is True:
                return default
            if arg is False:
                return 0
            return int(arg)

        out = [dict(s) for s in user_seasonalities]
        has_dt = min_dt is not None and min_dt >= 0
        fo = order('yearly', yearly, span < 730 * DAY_NS, 10)
        if fo > 0:
            out.append({'name': 'yearly', 'period': 365.25, 'fourier_order': fo,
                        'prior_scale': seasonality_prior_scale, 'mode': seasonality_mode})
        fo = order('weekly', weekly, (span < 14 * DAY_NS) or (has_dt and min_dt >= 7 * DAY_NS), 3)
        if fo > 0:
            out.append({'name': 'weekly', 'period': 7, 'fourier_order': fo,
                        'prior_scale': seasonality_prior_scale, 'mode': seasonality_mode})
        fo = order('daily', daily, (span < 2 * DAY_NS) or (has_dt and min_dt >= DAY_NS), 4)
        if fo > 0:
            out.append({'name': 'daily', 'period': 1, 'fourier_order': fo,
                        'prior_scale': seasonality_prior_scale, 'mode': seasonality_mode})
        return out

    @property
    def K(self):
        return sum(2 * int(s['fourier_order']) for s in self.seasonalities) + len(self.extra)

    @property
    def theta_stride(self):
        return 3 + self.n_changepoints + self.K

    def to_c(self):
        s = _lib.default_spec()
        s.growth = _lib.GROWTH_LOGISTIC if self.growth == 'logistic' else _lib.GROWTH_LINEAR
        s.n_changepoints = self.n_changepoints
        s.changepoint_range = self.changepoint_range
        s.changepoint_prior_scale = self.changepoint_prior_scale
        if len(self.seasonalities) > _lib.MAX_SEAS or len(self.extra) > _lib.MAX_EXTRA:
            raise ValueError('too many seasonalities (max %d) or extra columns (max %d)'
                             % (_lib.MAX_SEAS, _lib.MAX_EXTRA))
        s.n_seas = len(self.seasonalities)
        for i, se in enumerate(self.seasonalities):
            s.seas_period[i] = float(se['period'])
            s.seas_order[i] = int(se['fourier_order'])
            s.seas_prior_scale[i] = float(se.get('prior_scale', self.seasonality_prior_scale))
            s.seas_mode[i] = int(se.get('mode', self.seasonality_mode) == 'multiplicative')
        s.n_extra = len(self.extra)
        for i, e in enumerate(self.extra):
            s.extra_prior_scale[i] = float(e.get('prior_scale', self.holidays_prior_scale))
            s.extra_mode[i] = int(e.get('mode', self.seasonality_mode) == 'multiplicative')
        for k, v in self.lbfgs.items():
            setattr(s, k, v)
        return s

    def to_dict(self):
        return {'growth': self.growth, 'seasonality_mode': self.seasonality_mode,
                'n_changepoints': self.n_changepoints, 'changepoint_range': self.changepoint_range,
                'changepoint_prior_scale': self.changepoint_prior_scale,
                'seasonality_prior_scale': self.seasonality_prior_scale,
                'holidays_prior_scale': self.holidays_prior_scale,
                'seasonalities': self.seasonalities, 'extra': self.extra, 'lbfgs': self.lbfgs,
                **({'holidays': self.holidays} if self.holidays else {})}

    @classmethod
    def from_dict(cls, d):
        d = dict(d)
        lb = d.pop('lbfgs', {})
        return cls(**d, **lb)


class FitResult(object):
    """Arrays returned by a fit: theta [N][stride], y_scale [N], fval [N], status [N],
    n_iter [N], n_eval [N], grid (structured array, 1 or N entries)."""

    def __init__(self, spec, theta, y_scale, fval, status, n_iter, n_eval, grid):
        self.spec = spec
        self.theta = theta
        self.y_scale = y_scale
        self.fval = fval
        self.status = status
        self.n_iter = n_iter
        self.n_eval = n_eval
        self.grid = grid

    @property
    def N(self):
        return self.theta.shape[0]

    def grid_of(self, n):
        return self.grid[0 if len(self.grid) == 1 else n]


_ctx_cache = {}


def get_context(device=0, slot=0):
    """Cached tsf_ctx for a GPU.  `slot` distinguishes several contexts on one device (each
    context is single-threaded; the multi-device path gives every worker thread its own)."""
    c = _ctx_cache.get((device, slot))
    if c is None:
        c = _lib.Context(device)
        _ctx_cache[(device, slot)] = c
    return c


# ---- several GPUs from one process --------------------------------------------------------------
# series are independent, so a call is cut into contiguous blocks of series, one per
# device, each driven by its own host thread through its own tsf_ctx (ctypes releases the GIL for
# the duration of the C call); no data crosses between devices and the pieces are concatenated
# on the host.  Results are bit-identical to the single-device call.  bench.py measures the
# other arrangement (one PROCESS per GPU under torch.distributed.run).
MIN_SERIES_PER_DEVICE = 512


def resolve_devices(devices=None):
    """None -> the TSF_DEVICES environment variable ("all", or "0,1,2"; unset = one device,
    the default context); 'all' -> every visible GPU; otherwise a list of device ids (an id
    may repeat: that many contexts on that GPU)."""
    if devices is None:
        devices = os.environ.get('TSF_DEVICES') or None
        if devices is None:
            return None
    if isinstance(devices, str):
        if devices.strip().lower() == 'all':
            devices = list(range(_lib.load().tsf_device_count()))
        else:
            devices = [int(x) for x in devices.split(',') if x.strip() != '']
    devices = [int(d) for d in devices]
    return devices if len(devices) > 1 else None


def _contexts(devices):
    seen = {}
    out = []
    for d in devices:
        k = seen.get(d, 0)
        seen[d] = k + 1
        out.append(get_context(d, slot=k + 1))
    return out


def _cuts(weights, parts):
    """Cut points [parts+1] over len(weights) series so that every block carries about the
    same total weight (rows)."""
    cum = np.concatenate([[0], np.cumsum(np.asarray(weights, dtype=np.int64))])
    want = cum[-1] * np.arange(1, parts) / float(parts)
    inner = np.searchsorted(cum, want, side='left')
    return np.concatenate([[0], inner, [len(weights)]]).astype(np.int64)


def _run_blocks(fn, blocks):
    import concurrent.futures
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(blocks)) as ex:
        futs = [ex.submit(fn, *b) for b in blocks]
        return [f.result() for f in futs]


def _merge_fits(spec, parts, shared_grid):
    cat = lambda k: np.concatenate([getattr(p, k) for p in parts])   # noqa: E731
    grid = parts[0].grid if shared_grid else np.concatenate([p.grid for p in parts])
    return FitResult(spec, cat('theta'), cat('y_scale'), cat('fval'), cat('status'), cat('n_iter'),
                     cat('n_eval'), grid)


def _merge_interleaved(spec, parts_res, N, parts):
    """Inverse of the i mod parts split of an aligned panel."""
    def put(k):
        first = getattr(parts_res[0], k)
        out = np.zeros((N,) + first.shape[1:], dtype=first.dtype)
        for d, p in enumerate(parts_res):
            out[parallel.shard_indices(N, d, parts)] = getattr(p, k)
        return out
    return FitResult(spec, put('theta'), put('y_scale'), put('fval'), put('status'), put('n_iter'),
                     put('n_eval'), parts_res[0].grid)


def _opt_f64(a, N, name):
    if a is None:
        return None
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (N,)))
    return a


def _alloc_out(N, stride, n_grids):
    theta = np.zeros((N, stride))
    y_scale = np.zeros(N)
    fval = np.zeros(N)
    status = np.zeros(N, dtype=np.int32)
    n_iter = np.zeros(N, dtype=np.int32)
    n_eval = np.zeros(N, dtype=np.int32)
    grid = np.zeros(n_grids, dtype=_lib.GRID_DTYPE)
    out = _lib.TsfFitOut(theta.ctypes.data, y_scale.ctypes.data, fval.ctypes.data,
                         status.ctypes.data, n_iter.ctypes.data, n_eval.ctypes.data,
                         grid.ctypes.data)
    return out, (theta, y_scale, fval, status, n_iter, n_eval, grid)


def fit_aligned(spec, ds_ns, y, floor=None, cap=None, extra=None, ctx=None, devices=None, cost_hints=None):
    """Fit N series observed on the same T timestamps.  y: [N][T] float64/float32/int32.
    devices: see resolve_devices (several GPUs, one host thread each).
    cost_hints: optional [N] expected relative cost per series (the `n_eval` of an earlier fit of the same series):
    the launch starts its longest fits first (tsf_set_cost_hints); results do not depend on it."""
    devs = None if ctx is not None else resolve_devices(devices)
    ch = None if cost_hints is None else np.ascontiguousarray(cost_hints, dtype=np.int32)
    if ch is not None and ch.shape != (len(y),):
        raise ValueError('cost_hints must be [N]')
    if devs and len(y) >= 2 * MIN_SERIES_PER_DEVICE:
        parts = min(len(devs), len(y) // MIN_SERIES_PER_DEVICE)
        fl = _opt_f64(floor, len(y), 'floor')
        cp = _opt_f64(cap, len(y), 'cap')
        # series i goes to device i mod parts (evaluation counts vary 3-40x per series and
        # neighbours in a panel tend to be alike; interleaving evens the devices out where contiguous
        # blocks would not); parallel.shard_indices is the one statement of that layout (bench.py --gpus N
        # deals its ranks the same way)
        y = np.asarray(y)
        blocks = [(c, parallel.shard_indices(len(y), d, parts)) for d, c in enumerate(_contexts(devs[:parts]))]
        res = _run_blocks(lambda c, idx: fit_aligned(
            spec, ds_ns, np.ascontiguousarray(y[idx]), None if fl is None else fl[idx],
            None if cp is None else cp[idx], extra, ctx=c,
            cost_hints=None if ch is None else ch[idx]), blocks)
        return _merge_interleaved(spec, res, len(y), parts)
    ctx = ctx or get_context()
    L = _lib.load()
    ds_ns = np.ascontiguousarray(ds_ns, dtype=np.int64)
    y = np.ascontiguousarray(y)
    if y.ndim != 2 or y.shape[1] != ds_ns.shape[0]:
        raise ValueError('y must be [N][T] with T == len(ds)')
    if ch is not None:
        ctx.check(L.tsf_set_cost_hints(ctx.handle, ch.ctypes.data, ch.shape[0]))
    N, T = y.shape
    cs = spec.to_c()
    floor = _opt_f64(floor, N, 'floor')
    cap = _opt_f64(cap, N, 'cap')
    ex = None
    if spec.extra:
        ex = np.ascontiguousarray(extra, dtype=np.float64)
        if ex.shape != (len(spec.extra), T):
            raise ValueError('extra must be [n_extra][T]')
    out, arrs = _alloc_out(N, spec.theta_stride, 1)
    rc = L.tsf_fit_aligned(ctx.handle, ctypes.byref(cs), N, T, ds_ns.ctypes.data, y.ctypes.data,
                           _lib.y_dtype_code(y), _lib._ptr(floor), _lib._ptr(cap), _lib._ptr(ex),
                           ctypes.byref(out))
    ctx.check(rc)
    return FitResult(spec, *arrs)


def fit_ragged(spec, offsets, ds_ns, y, floor=None, cap=None, extra=None, ctx=None, devices=None, cost_hints=None):
    """Fit N series of different lengths / timestamps; series n owns rows
    offsets[n]:offsets[n+1] of ds_ns / y (each slice sorted by ds, NaN rows removed).
    cost_hints: as in fit_aligned."""
    devs = None if ctx is not None else resolve_devices(devices)
    ch = None if cost_hints is None else np.ascontiguousarray(cost_hints, dtype=np.int32)
    if ch is not None and ch.shape != (len(offsets) - 1,):
        raise ValueError('cost_hints must be [N]')
    if devs and len(offsets) - 1 >= 2 * MIN_SERIES_PER_DEVICE:
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        N = len(offsets) - 1
        parts = min(len(devs), N // MIN_SERIES_PER_DEVICE)
        cuts = _cuts(np.diff(offsets), parts)
        fl = _opt_f64(floor, N, 'floor')
        cp = _opt_f64(cap, N, 'cap')
        ex = None if extra is None else np.asarray(extra)
        blocks = [(c, int(a), int(b)) for c, a, b in zip(_contexts(devs[:parts]), cuts[:-1], cuts[1:])
                  if b > a]

        def one(c, a, b):
            r0, r1 = int(offsets[a]), int(offsets[b])
            return fit_ragged(spec, offsets[a:b + 1] - r0, ds_ns[r0:r1], y[r0:r1],
                              None if fl is None else fl[a:b], None if cp is None else cp[a:b],
                              None if ex is None else ex[:, r0:r1], ctx=c,
                              cost_hints=None if ch is None else ch[a:b])
        return _merge_fits(spec, _run_blocks(one, blocks), shared_grid=False)
    ctx = ctx or get_context()
    L = _lib.load()
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    ds_ns = np.ascontiguousarray(ds_ns, dtype=np.int64)
    y = np.ascontiguousarray(y)
    N = len(offsets) - 1
    if y.ndim != 1 or y.shape[0] != ds_ns.shape[0] or offsets[-1] != y.shape[0]:
        raise ValueError('ds / y must be 1-D with offsets[-1] rows')
    cs = spec.to_c()
    floor = _opt_f64(floor, N, 'floor')
    cap = _opt_f64(cap, N, 'cap')
    ex = None
    if spec.extra:
        ex = np.ascontiguousarray(extra, dtype=np.float64)
        if ex.shape != (len(spec.extra), y.shape[0]):
            raise ValueError('extra must be [n_extra][total_rows]')
    out, arrs = _alloc_out(N, spec.theta_stride, N)
    if ch is not None:
        ctx.check(L.tsf_set_cost_hints(ctx.handle, ch.ctypes.data, ch.shape[0]))
    rc = L.tsf_fit_ragged(ctx.handle, ctypes.byref(cs), N, offsets.ctypes.data, ds_ns.ctypes.data,
                          y.ctypes.data, _lib.y_dtype_code(y), _lib._ptr(floor), _lib._ptr(cap),
                          _lib._ptr(ex), ctypes.byref(out))
    ctx.check(rc)
    return FitResult(spec, *arrs)


def predict(spec, theta, y_scale, grid, ds_future_ns, floor=None, cap=None, extra_future=None,
            want_int=False, ctx=None, devices=None):
    """yhat [N][H] (float64) and, if want_int, the reference's int-truncated + floor-clamped
    column (prophet_scorer.py:73-84).  ds_future_ns: [H] (shared) or [N][H]."""
    devs = None if ctx is not None else resolve_devices(devices)
    if devs and len(theta) >= 2 * MIN_SERIES_PER_DEVICE:
        N = len(theta)
        parts = min(len(devs), N // MIN_SERIES_PER_DEVICE)
        cuts = _cuts(np.ones(N, np.int64), parts)
        fl = _opt_f64(floor, N, 'floor')
        cp = _opt_f64(cap, N, 'cap')
        fut = np.asarray(ds_future_ns)
        exf = None if extra_future is None else np.asarray(extra_future)
        blocks = [(c, int(a), int(b)) for c, a, b in zip(_contexts(devs[:parts]), cuts[:-1], cuts[1:])]

        def one(c, a, b):
            return predict(spec, theta[a:b], np.asarray(y_scale)[a:b],
                           grid if len(grid) == 1 else grid[a:b],
                           fut if fut.ndim == 1 else fut[a:b],
                           None if fl is None else fl[a:b], None if cp is None else cp[a:b],
                           exf if (exf is None or fut.ndim == 1) else exf[a:b], want_int=want_int, ctx=c)
        res = _run_blocks(one, blocks)
        if want_int:
            return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])
        return np.concatenate(res)
    ctx = ctx or get_context()
    L = _lib.load()
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    N = theta.shape[0]
    y_scale = np.ascontiguousarray(y_scale, dtype=np.float64)
    grid = np.ascontiguousarray(grid, dtype=_lib.GRID_DTYPE)
    ds_future_ns = np.ascontiguousarray(ds_future_ns, dtype=np.int64)
    shared = ds_future_ns.ndim == 1
    H = ds_future_ns.shape[-1]
    if not shared and ds_future_ns.shape != (N, H):
        raise ValueError('ds_future must be [H] or [N][H]')
    cs = spec.to_c()
    floor = _opt_f64(floor, N, 'floor')
    cap = _opt_f64(cap, N, 'cap')
    ex = None
    if spec.extra:
        ex = np.ascontiguousarray(extra_future, dtype=np.float64)
        want = (len(spec.extra), H) if shared else (N, len(spec.extra), H)
        if ex.shape != want:
            raise ValueError('extra_future must be %r' % (want,))
    yhat = np.zeros((N, H))
    yint = np.zeros((N, H), dtype=np.int32) if want_int else None
    rc = L.tsf_predict(ctx.handle, ctypes.byref(cs), N, H, theta.ctypes.data, y_scale.ctypes.data,
                       grid.ctypes.data, len(grid), ds_future_ns.ctypes.data, int(shared),
                       _lib._ptr(floor), _lib._ptr(cap), _lib._ptr(ex), yhat.ctypes.data,
                       _lib._ptr(yint))
    ctx.check(rc)
    return (yhat, yint) if want_int else yhat


def predict_intervals(spec, theta, y_scale, grid, ds_future_ns, floor=None, cap=None, extra_future=None,
                      series_key=None, uncertainty_samples=1000, interval_width=0.8, seed=0, ctx=None):
    """(yhat, yhat_lower, yhat_upper), each [N][H]: fbprophet's predict_uncertainty -- which the
    reference computes inside model.predict (prophet_scorer.py:70) and drops (:86) -- with a seeded
    counter-based generator (include/tsf.h tsf_predict_intervals).  series_key [N] int64: what the
    random streams are keyed by (e.g. a hash of (series_id, dim_id)), so that a series gets the same
    interval whatever batch it is in; default: its index in this call."""
    ctx = ctx or get_context()
    L = _lib.load()
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    N = theta.shape[0]
    y_scale = np.ascontiguousarray(y_scale, dtype=np.float64)
    grid = np.ascontiguousarray(grid, dtype=_lib.GRID_DTYPE)
    ds_future_ns = np.ascontiguousarray(ds_future_ns, dtype=np.int64)
    shared = ds_future_ns.ndim == 1
    H = ds_future_ns.shape[-1]
    if not shared and ds_future_ns.shape != (N, H):
        raise ValueError('ds_future must be [H] or [N][H]')
    cs = spec.to_c()
    floor = _opt_f64(floor, N, 'floor')
    cap = _opt_f64(cap, N, 'cap')
    ex = None
    if spec.extra:
        ex = np.ascontiguousarray(extra_future, dtype=np.float64)
        want = (len(spec.extra), H) if shared else (N, len(spec.extra), H)
        if ex.shape != want:
            raise ValueError('extra_future must be %r' % (want,))
    key = None if series_key is None else np.ascontiguousarray(series_key, dtype=np.int64)
    if key is not None and key.shape != (N,):
        raise ValueError('series_key must be [N]')
    yhat, lo, hi = np.zeros((N, H)), np.zeros((N, H)), np.zeros((N, H))
    rc = L.tsf_predict_intervals(ctx.handle, ctypes.byref(cs), N, H, theta.ctypes.data, y_scale.ctypes.data,
                                 grid.ctypes.data, len(grid), ds_future_ns.ctypes.data, int(shared),
                                 _lib._ptr(floor), _lib._ptr(cap), _lib._ptr(ex), _lib._ptr(key),
                                 int(uncertainty_samples), float(interval_width), int(seed),
                                 yhat.ctypes.data, lo.ctypes.data, hi.ctypes.data)
    ctx.check(rc)
    return yhat, lo, hi


# ---- diagnostics used by the parity tests -----------------------------------------------------

def eval_aligned(spec, ds_ns, y, theta, floor=None, cap=None, extra=None, ctx=None):
    """-log posterior and gradient at theta [N][stride] for an aligned panel."""
    ctx = ctx or get_context()
    L = _lib.load()
    ds_ns = np.ascontiguousarray(ds_ns, dtype=np.int64)
    y = np.ascontiguousarray(y)
    N, T = y.shape
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    cs = spec.to_c()
    floor = _opt_f64(floor, N, 'floor')
    cap = _opt_f64(cap, N, 'cap')
    ex = np.ascontiguousarray(extra, dtype=np.float64) if spec.extra else None
    f = np.zeros(N)
    g = np.zeros_like(theta)
    rc = L.tsf_eval(ctx.handle, ctypes.byref(cs), N, T, ds_ns.ctypes.data, y.ctypes.data,
                    _lib.y_dtype_code(y), _lib._ptr(floor), _lib._ptr(cap), _lib._ptr(ex),
                    theta.ctypes.data, f.ctypes.data, g.ctypes.data)
    ctx.check(rc)
    return f, g


def eval_quadratic(spec, ds_ns, y, theta_ref, theta, extra=None, ctx=None):
    """-log posterior and gradient at theta [N][stride] in the QUADRATIC evaluation form built around the
    reference point theta_ref [N][stride] (include/tsf.h tsf_eval_quadratic): what fit_quad_kernel
    evaluates at the trial points of its line searches."""
    ctx = ctx or get_context()
    L = _lib.load()
    ds_ns = np.ascontiguousarray(ds_ns, dtype=np.int64)
    y = np.ascontiguousarray(y)
    N, T = y.shape
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    theta_ref = np.ascontiguousarray(theta_ref, dtype=np.float64)
    if theta.shape != (N, spec.theta_stride) or theta_ref.shape != theta.shape:
        raise ValueError('theta and theta_ref must be [N][theta_stride]')
    cs = spec.to_c()
    ex = np.ascontiguousarray(extra, dtype=np.float64) if spec.extra else None
    f = np.zeros(N)
    g = np.zeros_like(theta)
    rc = L.tsf_eval_quadratic(ctx.handle, ctypes.byref(cs), N, T, ds_ns.ctypes.data, y.ctypes.data,
                              _lib.y_dtype_code(y), _lib._ptr(ex), theta_ref.ctypes.data, theta.ctypes.data,
                              f.ctypes.data, g.ctypes.data)
    ctx.check(rc)
    return f, g


def design(spec, ds_ns, extra=None, ctx=None):
    """Design matrix X [T][K], scaled time t [T] and the grid info, as the device builds them."""
    ctx = ctx or get_context()
    L = _lib.load()
    ds_ns = np.ascontiguousarray(ds_ns, dtype=np.int64)
    T = len(ds_ns)
    cs = spec.to_c()
    ex = np.ascontiguousarray(extra, dtype=np.float64) if spec.extra else None
    X = np.zeros((T, spec.K))
    t = np.zeros(T)
    grid = np.zeros(1, dtype=_lib.GRID_DTYPE)
    rc = L.tsf_design(ctx.handle, ctypes.byref(cs), T, ds_ns.ctypes.data, _lib._ptr(ex),
                      X.ctypes.data, t.ctypes.data, grid.ctypes.data)
    ctx.check(rc)
    return X, t, grid


def selftest_math(op, a, b=None, ctx=None):
    ctx = ctx or get_context()
    L = _lib.load()
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
    out = np.zeros_like(a)
    rc = L.tsf_selftest_math(ctx.handle, int(op), a.size, a.ctypes.data, _lib._ptr(b),
                             out.ctypes.data)
    ctx.check(rc)
    return out
