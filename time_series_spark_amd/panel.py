"""Host-side panel packing: long (series_id, dim_id, ds, y) frames -> the SoA arrays the C-ABI
takes, and back.  This is the work Spark's shuffle + Arrow grouped-map runner does for the
reference (/root/reference/src/jobs/prophet_modeler.py:139-141): bring each series' rows
together; here they end up contiguous in one array instead of one pandas frame per group."""
import ctypes
import io
import json
import struct

import numpy as np
import pandas as pd

from . import _lib, forecaster as fc

MAGIC = b'TSFM'
VERSION = 2
KEYS = ['series_id', 'dim_id']


def ds_to_ns(ds):
    """datetime64 column / index -> int64 nanoseconds since the epoch (timezone-naive)."""
    v = pd.to_datetime(ds)
    if isinstance(v, pd.Series):
        v = v.dt.tz_localize(None) if v.dt.tz is not None else v
        return v.values.astype('datetime64[ns]').astype(np.int64)
    v = pd.DatetimeIndex(v)
    if v.tz is not None:
        v = v.tz_localize(None)
    return v.values.astype('datetime64[ns]').astype(np.int64)


class PackedPanel(object):
    """keys: DataFrame[series_id, dim_id] one row per series (in packed order);
    offsets [N+1]; ds_ns [rows]; y [rows] float64; aligned: True when every series has the same
    timestamp vector (then ds_grid [T] and y2d [N][T] are set)."""

    def __init__(self, keys, offsets, ds_ns, y, aligned=None):
        self.keys = keys
        self.offsets = offsets
        self.ds_ns = ds_ns
        self.y = y
        self.N = len(offsets) - 1
        lens = np.diff(offsets)
        self.lengths = lens
        self.aligned = False
        self.stats = None
        # fbprophet's history_dates = ALL ds of the group, null-y rows included (Prophet.fit sets
        # it before dropping them), and make_future_dataframe starts from its max: last_ds_all is
        # that max per series (pack_rows fills it in when rows were dropped); dropped_keys are
        # the (series_id, dim_id) groups whose every y is null (Prophet.fit raises for them)
        self.last_ds_all = ds_ns[offsets[1:] - 1] if self.N > 0 else np.zeros(0, np.int64)
        self.dropped_keys = []
        self.ds_grid = None
        self.y2d = None
        # aligned: the native packer saw it on its pass over the rows (tsf_pack_flags) -- None = look here
        if self.N > 0 and lens.min() == lens.max() and lens[0] > 0 and aligned is not False:
            T = int(lens[0])
            grid = ds_ns.reshape(self.N, T)
            if aligned or np.all(grid == grid[0]):
                self.aligned = True
                self.ds_grid = np.ascontiguousarray(grid[0])
                self.y2d = np.ascontiguousarray(y.reshape(self.N, T))
        self.has_inf = None          # (tsf_pack_flags: an infinite y among the packed rows; None = not looked at)
        self.integral = None         # every y an integer that fits int32 (the reference's quantity column)
        self.has_nat = None          # a ds that is pandas' NaT


def pack_long_frame(pdf, y_col='y', n_threads=0):
    """fbprophet's setup_dataframe host steps done for every group at once: drop rows whose y
    is NaN (Prophet.fit: history = df[df['y'].notnull()]), sort by ds within the group
    (stable), and lay the groups out contiguously, ascending by (series_id, dim_id).  The row
    movement runs in the native packer (tsf_pack_rows, include/tsf.h); a frame that already is
    in packed order is not copied at all."""
    need = KEYS + ['ds', y_col]
    for c in need:
        if c not in pdf.columns:
            raise ValueError("Dataframe must have columns %r" % (need,))
    # Round 6: the frame's columns go to the packer in their own types wherever it takes them -- the reference's schema is
    # int32 series_id / dim_id / y and datetime64 ds (prophet_modeler.py:12-17), and converting three 7.3 M-row columns to
    # int64 / float64 first was most of this function's time; infinity and NaT are reported by the packer's own pass.
    ycol = pdf[y_col]
    if ycol.dtype in (np.dtype(np.int32), np.dtype(np.float32), np.dtype(np.float64)):
        yv = ycol.to_numpy()
    else:
        yv = pd.to_numeric(ycol).to_numpy(dtype=np.float64, na_value=np.nan)
    dcol = pdf['ds']
    if dcol.dtype == np.dtype('datetime64[ns]'):
        ds_ns = dcol.to_numpy().view(np.int64)
    else:
        ds_ns = ds_to_ns(dcol)
    sid_col, did_col = pdf['series_id'], pdf['dim_id']
    if sid_col.dtype == np.dtype(np.int32) and did_col.dtype == np.dtype(np.int32):
        sid, did = sid_col.to_numpy(), did_col.to_numpy()
    else:
        sid = pd.to_numeric(sid_col).to_numpy(dtype=np.int64)
        did = pd.to_numeric(did_col).to_numpy(dtype=np.int64)
    panel = pack_rows(sid, did, ds_ns, yv, n_threads=n_threads,
                      key_dtypes=(_key_dtype(sid_col), _key_dtype(did_col)))
    if panel.has_inf:
        raise ValueError('Found infinity in column y.')
    if panel.has_nat or (panel.N == 0 and len(ds_ns) and (ds_ns == np.iinfo(np.int64).min).any()):
        raise ValueError('Found NaN in column ds.')
    return panel


def _key_dtype(col):
    return col.dtype if np.issubdtype(col.dtype, np.integer) else np.dtype(np.int64)


def pack_rows(sid, did, ds_ns, y, n_threads=0, key_dtypes=(np.int64, np.int64)):
    """Arrays form of pack_long_frame: sid, did int32 or int64 [n], ds_ns int64 [n], y float64 / float32 / int32 [n] (NaN =
    missing).  A table that already is in packed order is used in place, in those types (PackedPanel.y then has y's
    dtype); otherwise the packed y is float64."""
    L = _lib.load()
    if not (np.asarray(sid).dtype == np.asarray(did).dtype == np.dtype(np.int32)):
        sid = np.ascontiguousarray(sid, dtype=np.int64)
        did = np.ascontiguousarray(did, dtype=np.int64)
    sid, did = np.ascontiguousarray(sid), np.ascontiguousarray(did)
    ds_ns = np.ascontiguousarray(ds_ns, dtype=np.int64)
    y = np.asarray(y)
    if y.dtype not in (np.dtype(np.float64), np.dtype(np.float32), np.dtype(np.int32)):
        y = y.astype(np.float64)
    y = np.ascontiguousarray(y)
    n = len(y)
    if not (len(sid) == len(did) == len(ds_ns) == n):
        raise ValueError('pack_rows: column lengths differ')
    h = ctypes.c_void_p()
    n_rows, n_series, ident = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32()
    rc = L.tsf_pack_rows_typed(n, sid.ctypes.data, did.ctypes.data, sid.dtype.itemsize, ds_ns.ctypes.data, y.ctypes.data,
                               _lib.y_dtype_code(y), int(n_threads), ctypes.byref(h), ctypes.byref(n_rows),
                               ctypes.byref(n_series), ctypes.byref(ident))
    if rc != 0:
        raise _lib.TsfError('tsf_pack_rows failed (%d)' % rc)
    try:
        N, R = n_series.value, n_rows.value
        ksid, kdid = np.empty(N, np.int64), np.empty(N, np.int64)
        offsets = np.empty(N + 1, np.int64)
        span, min_dt, ymax = np.empty(N, np.int64), np.empty(N, np.int64), np.empty(N, np.float64)
        if ident.value:
            ds_out, y_out = ds_ns, y
            rc = L.tsf_pack_fetch(h, ksid.ctypes.data, kdid.ctypes.data, offsets.ctypes.data, None,
                                  None, span.ctypes.data, min_dt.ctypes.data, ymax.ctypes.data)
        else:
            ds_out, y_out = np.empty(R, np.int64), np.empty(R, np.float64)
            rc = L.tsf_pack_fetch(h, ksid.ctypes.data, kdid.ctypes.data, offsets.ctypes.data,
                                  ds_out.ctypes.data, y_out.ctypes.data, span.ctypes.data,
                                  min_dt.ctypes.data, ymax.ctypes.data)
        if rc != 0:
            raise _lib.TsfError('tsf_pack_fetch failed (%d)' % rc)
        f_al, f_inf, f_int, f_nat = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        if L.tsf_pack_flags(h, ctypes.byref(f_al), ctypes.byref(f_inf), ctypes.byref(f_int), ctypes.byref(f_nat)) != 0:
            raise _lib.TsfError('tsf_pack_flags failed')
    finally:
        L.tsf_pack_free(h)
    keys = pd.DataFrame({'series_id': ksid.astype(key_dtypes[0]), 'dim_id': kdid.astype(key_dtypes[1])})
    panel = PackedPanel(keys, offsets, ds_out, y_out, aligned=bool(f_al.value))
    panel.has_inf, panel.integral, panel.has_nat = bool(f_inf.value), bool(f_int.value), bool(f_nat.value)
    panel.stats = (span, min_dt, ymax)
    if R < n:
        # rows with a null y were dropped: they still count for the last history date
        nan = np.isnan(y)
        sid, did = sid.astype(np.int64), did.astype(np.int64)
        nmax = pd.DataFrame({'s': sid[nan], 'd': did[nan], 'ds': ds_ns[nan]}).groupby(['s', 'd'])['ds'].max()
        idx = pd.MultiIndex.from_arrays([ksid, kdid], names=['s', 'd'])
        m = nmax.reindex(idx).to_numpy(dtype=np.float64, na_value=np.nan)
        have = ~np.isnan(m)
        last = panel.last_ds_all.copy()
        last[have] = np.maximum(last[have], nmax.reindex(idx[have]).to_numpy(dtype=np.int64))
        panel.last_ds_all = last
        panel.dropped_keys = [k for k in nmax.index.difference(idx)]
    return panel


def rows_2d(arr, off, members, T):
    """[len(members)][T] matrix of the members' rows of a packed column (all of length T):
    a view when the members are consecutive series, a gathered copy otherwise."""
    members = np.asarray(members, dtype=np.int64)
    n = len(members)
    if n and members[-1] - members[0] + 1 == n and (n == 1 or (np.diff(members) == 1).all()):
        a0 = int(off[members[0]])
        return arr[a0:a0 + n * T].reshape(n, T)
    return arr[off[members][:, None] + np.arange(T, dtype=np.int64)[None, :]]


def group_by_grid(panel, members, min_group=2, keep_single=False):
    """Partition `members` (series indices of a PackedPanel) by identical timestamp vector.
    Returns (groups, rest): groups = list of index arrays (each >= min_group series sharing one
    grid, to be fitted through the aligned entry point: one set of design tables, shared through
    L2 / LDS by every series of the group), rest = the series whose grid nobody shares.
    keep_single: members that ALL share one grid form a group whatever their number (>= 2).
    Results do not depend on the grouping (the evaluation form is a property of the model and the
    aligned / ragged kernels are bit-identical); only the speed does."""
    members = np.asarray(members, dtype=np.int64)
    if len(members) == 0:
        return [], members
    if panel.aligned and len(members) == panel.N and len(members) >= (2 if keep_single else min_group):
        return [np.sort(members)], np.zeros(0, dtype=np.int64)
    off = panel.offsets
    lens = panel.lengths[members]
    if (panel.lengths <= 0).any():           # empty series present: treat everything as ragged
        return [], members
    # cheap necessary condition first: series that share a grid share its length, first and last timestamp.  When no
    # such class is large enough there is nothing to hash (the signature below reads every row of the panel)
    first = panel.ds_ns[off[:-1][members]]
    last = panel.ds_ns[off[1:][members] - 1]
    o3 = np.lexsort((last, first, lens))
    k3 = np.stack([lens[o3], first[o3], last[o3]], axis=1)
    cut = np.flatnonzero((k3[1:] != k3[:-1]).any(axis=1)) + 1
    sizes = np.diff(np.concatenate([[0], cut, [len(members)]]))
    need = 2 if (keep_single and len(sizes) == 1) else min_group
    if sizes.max() < need:
        return [], np.sort(members)
    # 64-bit signature of every grid: length and a position-weighted wrap-around sum of ds
    pos = np.arange(len(panel.ds_ns), dtype=np.uint64) - np.repeat(off[:-1].astype(np.uint64), panel.lengths)
    w = (pos * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0xD1B54A32D192ED03))
    sig_all = np.add.reduceat(panel.ds_ns.astype(np.uint64) * w, off[:-1])
    sig = sig_all[members]
    order = np.lexsort((members, sig, lens))
    ms, ls, ss = members[order], lens[order], sig[order]
    brk = np.flatnonzero((ls[1:] != ls[:-1]) | (ss[1:] != ss[:-1])) + 1
    groups, rest = [], []
    chunks = np.split(ms, brk)
    if keep_single and len(chunks) == 1:
        min_group = 2
    for chunk in chunks:
        if len(chunk) < min_group:              # (before the row-by-row comparison: small groups cost nothing)
            rest.extend(chunk.tolist())
            continue
        T = int(panel.lengths[chunk[0]])
        grid0 = panel.ds_ns[off[chunk[0]]:off[chunk[0]] + T]
        same = np.ones(len(chunk), dtype=bool)
        step = max(1, (1 << 22) // max(T, 1))                 # <= 32 MB of timestamps at a time
        for c0 in range(0, len(chunk), step):
            blk = rows_2d(panel.ds_ns, off, chunk[c0:c0 + step], T)
            same[c0:c0 + step] = (blk == grid0[None, :]).all(axis=1)
        if same.sum() >= min_group:
            groups.append(np.sort(chunk[same]))
        else:
            rest.extend(chunk[same].tolist())
        rest.extend(chunk[~same].tolist())   # signature collision: fit those on their own
    return groups, np.sort(np.asarray(rest, dtype=np.int64))


def per_series_stats(panel):
    """span, smallest non-zero spacing (ns; -1 if none), max y per series -- the inputs of
    fbprophet's set_auto_seasonalities and of the reference's cap = max(y) * cap_multiplier.
    Panels that came out of the native packer carry them already."""
    if getattr(panel, 'stats', None) is not None:
        return panel.stats
    off = panel.offsets
    N = panel.N
    if N == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0)
    first = panel.ds_ns[off[:-1]]
    last = panel.ds_ns[off[1:] - 1]
    big = np.iinfo(np.int64).max
    dd = np.full(len(panel.ds_ns), big, dtype=np.int64)      # dd[i] = ds[i+1]-ds[i] inside a series
    d = np.diff(panel.ds_ns)
    dd[:-1] = np.where(d > 0, d, big)
    dd[off[1:] - 1] = big                                    # last row of every series
    m = np.minimum.reduceat(dd, off[:-1])
    min_dt = np.where(m == big, -1, m).astype(np.int64)
    ymax = np.maximum.reduceat(panel.y, off[:-1])
    return last - first, min_dt, ymax


# ---- model blob ------------------------------------------------------------------------------
# The reference stores pickle.dumps(Prophet object) in a binary column
# (/root/reference/src/jobs/prophet_modeler.py:72-73).  An fbprophet pickle can neither be
# written nor read here; the replacement is a small versioned blob holding exactly what
# predict needs.

def _rec_dtype(n_theta, n_tchange):
    return np.dtype([('y_scale', '<f8'), ('start_ns', '<i8'), ('t_scale_ns', '<i8'),
                     ('last_ds_ns', '<i8'), ('T', '<i4'), ('S', '<i4'), ('i1', '<i4'), ('NT', '<i4'),
                     ('status', '<i4'), ('n_iter', '<i4'), ('n_theta', '<i4'), ('n_tchange', '<i4'),
                     ('theta', '<f8', (int(n_theta),)), ('t_change', '<f8', (int(n_tchange),))])


_REC_FIXED = 64          # bytes before theta in a record


def _prefix(spec_dict):
    hb = json.dumps(spec_dict, sort_keys=True).encode()
    return MAGIC + struct.pack('<II', VERSION, len(hb)) + hb


def dump_models_buffer(spec_dict, theta, y_scale, grid, last_ds_ns, status, n_iter):
    """The blobs of a fitted batch in ONE buffer: uint8 [N][stride], row n = the blob of series n.  Layout of a blob
    (little endian): 'TSFM', u32 version, u32 len(spec json), spec json (the constructor arguments, shared by the
    batch), then one fixed record: y_scale, start_ns, t_scale_ns, last_ds_ns, T, S, i1, NT, status, n_iter, n_theta,
    n_tchange, theta[n_theta], t_change[n_tchange].  grid: 1 entry (aligned batch) or one per series.  Assembled by the
    library (tsf_model_blobs, include/tsf.h); equal strides make the buffer an Arrow binary column as it stands."""
    theta = np.ascontiguousarray(np.atleast_2d(np.asarray(theta, dtype=np.float64)))
    N, nth = theta.shape
    grid = np.ascontiguousarray(grid, dtype=_lib.GRID_DTYPE)
    ntc = int(grid['S'].max()) if len(grid) else 0
    pre = _prefix(spec_dict)

    def col(a, dt):
        return np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=dt), (N,)))
    y_scale, last_ds_ns = col(y_scale, np.float64), col(last_ds_ns, np.int64)
    status, n_iter = col(status, np.int32), col(n_iter, np.int32)
    out = np.empty((N, len(pre) + _REC_FIXED + 8 * (nth + ntc)), dtype=np.uint8)
    if N:
        rc = _lib.load().tsf_model_blobs(N, pre, len(pre), nth, theta.ctypes.data, y_scale.ctypes.data, grid.ctypes.data,
                                         len(grid), last_ds_ns.ctypes.data, status.ctypes.data, n_iter.ctypes.data, ntc,
                                         out.ctypes.data, 0)
        if rc != 0:
            raise _lib.TsfError('tsf_model_blobs failed (%d)' % rc)
    return out


def dump_models(spec_dict, theta, y_scale, grid, last_ds_ns, status, n_iter):
    """One blob (bytes) per series of a fitted batch: the rows of dump_models_buffer."""
    buf = dump_models_buffer(spec_dict, theta, y_scale, grid, last_ds_ns, status, n_iter)
    body = buf.tobytes()
    L = buf.shape[1]
    return [body[i * L:(i + 1) * L] for i in range(buf.shape[0])]


def dump_model(spec_dict, theta, y_scale, grid_row, last_ds_ns, status, n_iter):
    g = np.zeros(1, dtype=_lib.GRID_DTYPE)
    g[0] = grid_row
    return dump_models(spec_dict, [theta], [y_scale], g, [last_ds_ns], [status], [n_iter])[0]


def _split(blob):
    b = blob if isinstance(blob, bytes) else bytes(blob)
    if b[:4] != MAGIC:
        raise ValueError('not a time_series_spark_amd model blob')
    ver, hl = struct.unpack_from('<II', b, 4)
    if ver != VERSION:
        raise ValueError('unsupported model blob version %d' % ver)
    return b[:12 + hl], b[12 + hl:]


def load_models(blobs):
    """Inverse of dump_models for a whole column: returns a list of
    (spec_dict, positions, records) -- one entry per distinct (spec, record shape); positions
    index into `blobs`; records is a structured array (fields as in dump_models).  None
    entries are skipped."""
    # the usual column -- every series of a run fitted with one spec: all blobs the same length with the same prefix --
    # is one reshape of the joined bytes instead of a Python loop over the blobs; a column that arrives as ONE buffer
    # (uint8 [n][L]: dump_models_buffer, or the data buffer of an Arrow binary column with equal strides,
    # model_column_buffer) is not even joined
    if isinstance(blobs, np.ndarray) and blobs.dtype == np.uint8 and blobs.ndim == 2:
        got = _load_uniform(blobs)
        if got is not None:
            return got
        blobs = [bytes(r) for r in blobs]
    n = len(blobs)
    if n > 1 and all(type(b) is bytes for b in blobs):
        L = len(blobs[0])
        if min(map(len, blobs)) == L == max(map(len, blobs)):
            got = _load_uniform(np.frombuffer(b''.join(blobs), dtype=np.uint8).reshape(n, L))
            if got is not None:
                return got
    buckets = {}
    for i, blob in enumerate(blobs):
        if blob is None:
            continue
        pre, body = _split(blob)
        buckets.setdefault((pre, len(body)), ([], []))
        pos, bodies = buckets[(pre, len(body))]
        pos.append(i)
        bodies.append(body)
    out = []
    for (pre, blen), (pos, bodies) in buckets.items():
        if blen < _REC_FIXED:
            raise ValueError('truncated model blob')
        nth, ntc = struct.unpack_from('<ii', bodies[0], _REC_FIXED - 8)
        dt = _rec_dtype(nth, ntc)
        if dt.itemsize != blen:
            raise ValueError('model blob size does not match its header')
        rec = np.frombuffer(b''.join(bodies), dtype=dt)
        if (rec['n_theta'] != nth).any() or (rec['n_tchange'] != ntc).any():
            raise ValueError('inconsistent model blobs')
        spec = json.loads(pre[12:].decode())
        out.append((spec, np.asarray(pos, dtype=np.int64), rec))
    return out


def _load_uniform(arr):
    """load_models for uint8 [n][L] rows that share one prefix (else None)."""
    n, L = arr.shape
    if n == 0:
        return []
    pre, body0 = _split(arr[0].tobytes())
    pl = len(pre)
    if L - pl < _REC_FIXED or not (arr[:, :pl] == arr[0, :pl]).all():
        return None
    nth, ntc = struct.unpack_from('<ii', body0, _REC_FIXED - 8)
    dt = _rec_dtype(nth, ntc)
    if dt.itemsize != L - pl:
        raise ValueError('model blob size does not match its header')
    rec = np.ascontiguousarray(arr[:, pl:]).view(dt).reshape(n)
    if (rec['n_theta'] != nth).any() or (rec['n_tchange'] != ntc).any():
        raise ValueError('inconsistent model blobs')
    return [(json.loads(pre[12:].decode()), np.arange(n, dtype=np.int64), rec)]


def model_column_arrow(buf):
    """uint8 [n][L] (dump_models_buffer) -> pyarrow binary array over the same bytes (no copy, no Python object per
    series): offsets n * L."""
    import pyarrow as pa
    n, L = buf.shape
    buf = np.ascontiguousarray(buf)
    if n * L < 2 ** 31:
        off = (np.arange(n + 1, dtype=np.int64) * L).astype(np.int32)
        return pa.Array.from_buffers(pa.binary(), n, [None, pa.py_buffer(off), pa.py_buffer(buf)])
    off = np.arange(n + 1, dtype=np.int64) * L
    return pa.Array.from_buffers(pa.large_binary(), n, [None, pa.py_buffer(off), pa.py_buffer(buf)])


def model_column_buffer(col):
    """Inverse of model_column_arrow for a model column read from parquet (pyarrow Array / ChunkedArray of binary):
    uint8 [n][L] view of its data buffer when every blob has the same length and none is null -- what load_models takes
    without a Python object per series --, else None (the caller falls back to col.to_pylist())."""
    import pyarrow as pa
    if isinstance(col, pa.ChunkedArray):
        col = col.chunk(0) if col.num_chunks == 1 else col.combine_chunks()
    if len(col) == 0 or col.null_count:
        return None
    wide = pa.types.is_large_binary(col.type)
    if not (wide or pa.types.is_binary(col.type)):
        return None
    _, off_b, data_b = col.buffers()
    off = np.frombuffer(off_b, dtype=np.int64 if wide else np.int32)[col.offset:col.offset + len(col) + 1]
    L = int(off[1] - off[0])
    if L <= 0 or not (np.diff(off) == L).all():
        return None
    return np.frombuffer(data_b, dtype=np.uint8)[int(off[0]):int(off[0]) + len(col) * L].reshape(len(col), L)


def load_model(blob):
    if blob is None:
        return None
    (spec, _, rec), = load_models([blob])
    r = rec[0]
    head = {f: (float(r[f]) if f == 'y_scale' else int(r[f]))
            for f in ('y_scale', 'start_ns', 't_scale_ns', 'last_ds_ns', 'T', 'S', 'i1', 'NT', 'status',
                      'n_iter', 'n_theta')}
    head['spec'] = spec
    head['theta'] = r['theta'].copy()
    head['t_change'] = r['t_change'][:head['S']].copy()
    return head


def grid_from_records(rec):
    g = np.zeros(len(rec), dtype=_lib.GRID_DTYPE)
    for f in ('start_ns', 't_scale_ns', 'T', 'S', 'i1', 'NT'):
        g[f] = rec[f]
    ntc = rec['t_change'].shape[1]
    g['t_change'][:, :ntc] = rec['t_change']
    return g


def grid_from_models(models):
    g = np.zeros(len(models), dtype=_lib.GRID_DTYPE)
    for i, m in enumerate(models):
        for f in ('start_ns', 't_scale_ns', 'T', 'S', 'i1', 'NT'):
            g[i][f] = m[f]
        g[i]['t_change'][:m['S']] = m['t_change']
    return g


def future_dates(last_ds_ns, periods, freq):
    """Prophet.make_future_dataframe(periods, freq, include_history=False) for many series:
    date_range(start=last_date, periods=periods+1, freq) minus entries <= last_date, first
    `periods` kept.  Returns int64 [N][periods]."""
    last_ds_ns = np.asarray(last_ds_ns, dtype=np.int64)
    uniq, inv = np.unique(last_ds_ns, return_inverse=True)
    table = np.zeros((len(uniq), periods), dtype=np.int64)
    for i, v in enumerate(uniq):
        last = pd.Timestamp(int(v))
        dates = pd.date_range(start=last, periods=periods + 1, freq=freq)
        dates = dates[dates > last][:periods]
        if len(dates) != periods:
            raise ValueError('frequency %r yields %d future dates, wanted %d'
                             % (freq, len(dates), periods))
        table[i] = dates.values.astype('datetime64[ns]').astype(np.int64)
    return table[inv.reshape(-1)]
