"""Host-side panel packing: long (series_id, dim_id, ds, y) frames -> the SoA arrays the C-ABI
takes, and back.  This is the work Spark's shuffle + Arrow grouped-map runner does for the
reference (/root/reference/src/jobs/prophet_modeler.py:139-141): bring each series' rows
together; here they end up contiguous in one array instead of one pandas frame per group."""
import io
import json
import struct

import numpy as np
import pandas as pd

from . import forecaster as fc

MAGIC = b'TSFM'
VERSION = 1
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

    def __init__(self, keys, offsets, ds_ns, y):
        self.keys = keys
        self.offsets = offsets
        self.ds_ns = ds_ns
        self.y = y
        self.N = len(offsets) - 1
        lens = np.diff(offsets)
        self.lengths = lens
        self.aligned = False
        self.ds_grid = None
        self.y2d = None
        if self.N > 0 and lens.min() == lens.max() and lens[0] > 0:
            T = int(lens[0])
            grid = ds_ns.reshape(self.N, T)
            if np.all(grid == grid[0]):
                self.aligned = True
                self.ds_grid = np.ascontiguousarray(grid[0])
                self.y2d = np.ascontiguousarray(y.reshape(self.N, T))


def pack_long_frame(pdf, y_col='y'):
    """fbprophet's setup_dataframe host steps done for every group at once: drop rows whose y
    is NaN (Prophet.fit: history = df[df['y'].notnull()]), sort by ds within the group
    (stable), and lay the groups out contiguously."""
    need = KEYS + ['ds', y_col]
    for c in need:
        if c not in pdf.columns:
            raise ValueError("Dataframe must have columns %r" % (need,))
    df = pdf[need]
    yv = pd.to_numeric(df[y_col]).to_numpy(dtype=np.float64, na_value=np.nan)
    if np.isinf(yv).any():
        raise ValueError('Found infinity in column y.')
    ds_ns = ds_to_ns(df['ds'])
    if (ds_ns == np.iinfo(np.int64).min).any():
        raise ValueError('Found NaN in column ds.')
    keep = ~np.isnan(yv)
    sid = df['series_id'].to_numpy()[keep]
    did = df['dim_id'].to_numpy()[keep]
    ds_ns = ds_ns[keep]
    yv = yv[keep]
    order = np.lexsort((ds_ns, did, sid))      # stable: ties keep input order
    sid, did, ds_ns, yv = sid[order], did[order], ds_ns[order], yv[order]
    if len(sid) == 0:
        return PackedPanel(pd.DataFrame({'series_id': [], 'dim_id': []}), np.zeros(1, np.int64),
                           ds_ns, yv)
    new = np.ones(len(sid), dtype=bool)
    new[1:] = (sid[1:] != sid[:-1]) | (did[1:] != did[:-1])
    starts = np.flatnonzero(new)
    offsets = np.concatenate([starts, [len(sid)]]).astype(np.int64)
    keys = pd.DataFrame({'series_id': sid[starts], 'dim_id': did[starts]})
    return PackedPanel(keys, offsets, np.ascontiguousarray(ds_ns), np.ascontiguousarray(yv))


def group_by_grid(panel, members, min_group=2):
    """Partition `members` (series indices of a PackedPanel) by identical timestamp vector.
    Returns (groups, rest): groups = list of index arrays (each >= min_group series sharing one
    grid, to be fitted through the aligned entry point: one set of design tables, shared through
    L2 / LDS by every series of the group), rest = the series whose grid nobody shares.
    Results do not depend on the grouping (the evaluation form is a property of the model and the
    aligned / ragged kernels are bit-identical); only the speed does."""
    members = np.asarray(members, dtype=np.int64)
    if len(members) == 0:
        return [], members
    off = panel.offsets
    lens = panel.lengths[members]
    # 64-bit signature of every grid: length and a position-weighted wrap-around sum of ds
    pos = np.arange(len(panel.ds_ns), dtype=np.uint64) - np.repeat(off[:-1].astype(np.uint64), panel.lengths)
    w = (pos * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0xD1B54A32D192ED03))
    sig_all = np.add.reduceat(panel.ds_ns.astype(np.uint64) * w, off[:-1][panel.lengths > 0]) \
        if (panel.lengths > 0).all() else None
    if sig_all is None:                      # empty series present: treat everything as ragged
        return [], members
    sig = sig_all[members]
    order = np.lexsort((sig, lens))
    ms, ls, ss = members[order], lens[order], sig[order]
    brk = np.flatnonzero((ls[1:] != ls[:-1]) | (ss[1:] != ss[:-1])) + 1
    groups, rest = [], []
    for chunk in np.split(ms, brk):
        if len(chunk) < min_group:
            rest.extend(chunk.tolist())
            continue
        T = int(panel.lengths[chunk[0]])
        grid0 = panel.ds_ns[off[chunk[0]]:off[chunk[0]] + T]
        same = [m for m in chunk if np.array_equal(panel.ds_ns[off[m]:off[m] + T], grid0)]
        diff = [m for m in chunk if not np.array_equal(panel.ds_ns[off[m]:off[m] + T], grid0)]
        if len(same) >= min_group:
            groups.append(np.sort(np.asarray(same, dtype=np.int64)))
        else:
            rest.extend(same)
        rest.extend(diff)                    # signature collision: fit those on their own
    return groups, np.sort(np.asarray(rest, dtype=np.int64))


def per_series_stats(panel):
    """span, smallest non-zero spacing (ns; -1 if none), max y per series -- the inputs of
    fbprophet's set_auto_seasonalities and of the reference's cap = max(y) * cap_multiplier."""
    off = panel.offsets
    N = panel.N
    first = panel.ds_ns[off[:-1]]
    last = panel.ds_ns[off[1:] - 1]
    d = np.diff(panel.ds_ns)
    big = np.iinfo(np.int64).max
    dd = np.where(d > 0, d, big)
    # mask differences that straddle two series
    if N > 1:
        dd[off[1:-1] - 1] = big
    min_dt = np.full(N, -1, dtype=np.int64)
    for n in range(N):          # reduceat needs non-empty segments; lengths can be 1
        a, b = off[n], off[n + 1] - 1
        if b > a:
            m = dd[a:b].min()
            min_dt[n] = m if m != big else -1
    ymax = np.maximum.reduceat(panel.y, off[:-1])
    return last - first, min_dt, ymax


# ---- model blob ------------------------------------------------------------------------------
# The reference stores pickle.dumps(Prophet object) in a binary column
# (/root/reference/src/jobs/prophet_modeler.py:72-73).  An fbprophet pickle can neither be
# written nor read here; the replacement is a small versioned blob holding exactly what
# predict needs.

def dump_model(spec_dict, theta, y_scale, grid_row, last_ds_ns, status, n_iter):
    S = int(grid_row['S'])
    head = {'spec': spec_dict, 'y_scale': float(y_scale), 'start_ns': int(grid_row['start_ns']),
            't_scale_ns': int(grid_row['t_scale_ns']), 'T': int(grid_row['T']), 'S': S,
            'i1': int(grid_row['i1']), 'NT': int(grid_row['NT']), 'last_ds_ns': int(last_ds_ns),
            'status': int(status), 'n_iter': int(n_iter), 'n_theta': int(len(theta))}
    hb = json.dumps(head, sort_keys=True).encode()
    buf = io.BytesIO()
    buf.write(MAGIC)
    buf.write(struct.pack('<II', VERSION, len(hb)))
    buf.write(hb)
    buf.write(np.asarray(theta, dtype='<f8').tobytes())
    buf.write(np.asarray(grid_row['t_change'][:S], dtype='<f8').tobytes())
    return buf.getvalue()


def load_model(blob):
    if blob is None:
        return None
    b = bytes(blob)
    if b[:4] != MAGIC:
        raise ValueError('not a time_series_spark_amd model blob')
    ver, hl = struct.unpack('<II', b[4:12])
    if ver != VERSION:
        raise ValueError('unsupported model blob version %d' % ver)
    head = json.loads(b[12:12 + hl].decode())
    p = 12 + hl
    nt = head['n_theta']
    theta = np.frombuffer(b, dtype='<f8', count=nt, offset=p).copy()
    p += 8 * nt
    tch = np.frombuffer(b, dtype='<f8', count=head['S'], offset=p).copy()
    head['theta'] = theta
    head['t_change'] = tch
    return head


def grid_from_models(models):
    from . import _lib
    g = np.zeros(len(models), dtype=_lib.GRID_DTYPE)
    for i, m in enumerate(models):
        g[i]['start_ns'] = m['start_ns']
        g[i]['t_scale_ns'] = m['t_scale_ns']
        g[i]['T'] = m['T']
        g[i]['S'] = m['S']
        g[i]['i1'] = m['i1']
        g[i]['NT'] = m['NT']
        g[i]['t_change'][:m['S']] = m['t_change']
    return g


def future_dates(last_ds_ns, periods, freq):
    """Prophet.make_future_dataframe(periods, freq, include_history=False) for many series:
    date_range(start=last_date, periods=periods+1, freq) minus entries <= last_date, first
    `periods` kept.  Returns int64 [N][periods]."""
    last_ds_ns = np.asarray(last_ds_ns, dtype=np.int64)
    out = np.zeros((len(last_ds_ns), periods), dtype=np.int64)
    cache = {}
    for i, v in enumerate(last_ds_ns):
        r = cache.get(int(v))
        if r is None:
            last = pd.Timestamp(int(v))
            dates = pd.date_range(start=last, periods=periods + 1, freq=freq)
            dates = dates[dates > last][:periods]
            if len(dates) != periods:
                raise ValueError('frequency %r yields %d future dates, wanted %d'
                                 % (freq, len(dates), periods))
            r = dates.values.astype('datetime64[ns]').astype(np.int64)
            cache[int(v)] = r
        out[i] = r
    return out
