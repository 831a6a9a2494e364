"""Deterministic synthetic panels (SURVEY.md 8d `make_panel`): the reference ships one 816-row
fixture and no benchmark data, so bench.py and the tests generate panels of the shapes named
in BASELINE.json.  Values are positive integers in the fixture's 1e3..1e5 range (the reference
schema stores y as int32, /root/reference/src/jobs/prophet_modeler.py:16)."""
import numpy as np

DAY_NS = 86400 * 10 ** 9
START_NS = 1514764800 * 10 ** 9        # 2018-01-01T00:00:00Z


def daily_grid(T, start_ns=START_NS):
    return start_ns + DAY_NS * np.arange(T, dtype=np.int64)


def make_panel(N, T, kind='linear', seed=751, dtype=np.float64, holidays=None):
    """Returns (ds_ns [T], y [N][T]).  kind: 'linear' (additive seasonality) or 'logistic'
    (saturating trend, multiplicative seasonality -- the reference's own model settings).
    holidays: optional [n_h][T] 0/1 indicator matrix whose effects are added."""
    rng = np.random.default_rng(seed)
    t = np.arange(T, dtype=np.float64)
    u = t / max(T - 1, 1)
    L = np.exp(rng.normal(np.log(3e4), 1.0, size=(N, 1)))
    if kind == 'linear':
        trend = 1.0 + rng.normal(0, 0.3, (N, 1)) * u
        for _ in range(2):
            cp = rng.uniform(0, 0.8, (N, 1))
            trend = trend + rng.normal(0, 0.2, (N, 1)) * np.maximum(u - cp, 0.0)
    elif kind == 'logistic':
        k = rng.uniform(2, 8, (N, 1))
        m = rng.uniform(0.2, 0.8, (N, 1))
        trend = 1.5 / (1.0 + np.exp(-k * (u - m)))
    else:
        raise ValueError(kind)
    seas = np.zeros((N, T))
    for n in range(1, 4):
        seas += rng.normal(0, 0.1, (N, 1)) * np.sin(2 * np.pi * n * t / 7.0)
        seas += rng.normal(0, 0.1, (N, 1)) * np.cos(2 * np.pi * n * t / 7.0)
    for n in range(1, 5):
        seas += rng.normal(0, 0.15, (N, 1)) * np.sin(2 * np.pi * n * t / 365.25)
        seas += rng.normal(0, 0.15, (N, 1)) * np.cos(2 * np.pi * n * t / 365.25)
    if holidays is not None:
        h = np.asarray(holidays, dtype=np.float64)
        seas += rng.normal(0, 0.2, (N, h.shape[0])) @ h
    if kind == 'linear':
        y = L * (trend + seas)
    else:
        y = L * trend * (1.0 + seas)
    y = y + rng.normal(0, 0.05, (N, T)) * L
    y = np.maximum(np.round(y), 1.0)
    return daily_grid(T), y.astype(dtype)


HOLIDAY_DOY = [14, 45, 82, 121, 150, 185, 230, 275, 310, 358, 20, 60, 100, 140, 200]


def holiday_frame(ds_ns, n_holidays=10, lower=-1, upper=1):
    """The holidays frame (Prophet(holidays=...): holiday, ds, lower_window, upper_window) whose indicator columns are
    holiday_matrix's -- what the literal restatement (oracle/fbprophet_restated.ProphetOracle) takes."""
    import pandas as pd
    days = (np.asarray(ds_ns, dtype=np.int64) // DAY_NS).astype(np.int64)
    years = np.unique(days.astype('datetime64[D]').astype('datetime64[Y]').astype(int) + 1970)
    rows = []
    for hi, d0 in enumerate(HOLIDAY_DOY[:n_holidays]):
        for yv in years:
            rows.append(('h%02d' % hi, np.datetime64('%d-01-01' % yv, 'D') + np.timedelta64(d0, 'D'), lower, upper))
    return pd.DataFrame(rows, columns=['holiday', 'ds', 'lower_window', 'upper_window'])


def holiday_matrix(ds_ns, n_holidays=10, lower=-1, upper=1):
    """Indicator columns for `n_holidays` fixed month/day dates per year with windows
    [lower, upper] (cfg4).  Returns (matrix [n_holidays*(upper-lower+1)][T], names)."""
    days = (np.asarray(ds_ns, dtype=np.int64) // DAY_NS).astype(np.int64)
    dates = days.astype('datetime64[D]')
    years = np.unique(dates.astype('datetime64[Y]').astype(int) + 1970)
    doy = HOLIDAY_DOY[:n_holidays]
    cols, names = [], []
    for hi, d0 in enumerate(doy):
        for off in range(lower, upper + 1):
            col = np.zeros(len(days))
            for yv in years:
                base = np.datetime64('%d-01-01' % yv, 'D') + np.timedelta64(d0 + off, 'D')
                col[dates == base] = 1.0
            cols.append(col)
            names.append('h%02d_delim_%s%d' % (hi, '+' if off >= 0 else '-', abs(off)))
    order = np.argsort(names)       # fbprophet sorts holiday columns by name
    return np.array(cols)[order], [names[i] for i in order]
