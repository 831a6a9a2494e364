"""Explicit design columns fbprophet builds on the host: holiday indicator
columns and extra-regressor standardisation.  The kernels take them as `extra` columns
(include/tsf.h: extra [n_extra][T], extra_future); this module makes them from what a Prophet
user passes -- a holidays frame, regressor values -- for whole panels at once.

fbprophet 0.5 behaviour restated here (Prophet.make_holiday_features, Prophet.initialize_scales,
Prophet.setup_dataframe; reached by the reference through Prophet(...).fit(pdf),
/root/reference/src/jobs/prophet_modeler.py:65-66, and model.predict(future_df),
/root/reference/src/jobs/prophet_scorer.py:70):
  * one 0/1 column per (holiday, window offset), offsets lower_window..upper_window around every
    date of the holiday, named "<holiday>_delim_<+|-><|offset|>", columns SORTED BY NAME; a row is 1
    where date(ds) equals the shifted holiday date; a (holiday, offset) pair with no matching row
    still gets its (all-zero) column;
  * one prior scale per holiday (rows of a holiday must agree; default holidays_prior_scale);
  * the columns follow the model's seasonality_mode;
  * a regressor is standardised with the mean / sample std (ddof = 1) of its TRAINING values unless
    it has fewer than two distinct values, or standardize='auto' and its values are exactly {0, 1}.
"""
import numpy as np
import pandas as pd

DAY_NS = 86400 * 10 ** 9


def _day_number(ds_ns):
    """date(ds) as days since the epoch (floor: also right before 1970)."""
    return np.floor_divide(np.asarray(ds_ns, dtype=np.int64), DAY_NS)


RESERVED_NAMES = frozenset([
    'trend', 'additive_terms', 'daily', 'weekly', 'yearly', 'holidays', 'zeros', 'extra_regressors_additive',
    'yhat', 'extra_regressors_multiplicative', 'multiplicative_terms',
    'ds', 'y', 'cap', 'floor', 'y_scaled', 'cap_scaled']) | frozenset(
        n + s for n in ('trend', 'additive_terms', 'daily', 'weekly', 'yearly', 'holidays', 'zeros',
                        'extra_regressors_additive', 'yhat', 'extra_regressors_multiplicative', 'multiplicative_terms')
        for s in ('_lower', '_upper'))


def normalize_holidays(holidays, default_prior_scale=10.0):
    """holidays: fbprophet-style DataFrame [holiday, ds, lower_window?, upper_window?, prior_scale?]
    or a list of dicts with the same keys ('ds' one date or a list of dates: what a YAML config can
    hold).  Returns a JSON-able list [{holiday, days: [int], lower_window, upper_window,
    prior_scale}] with one entry per input row group -- what goes into the model blob so that the
    scorer rebuilds the same columns for future dates."""
    if holidays is None:
        return []
    rows = []
    if isinstance(holidays, pd.DataFrame):
        for _ix, row in holidays.iterrows():
            rows.append({'holiday': row['holiday'], 'ds': [row['ds']],
                         'lower_window': row.get('lower_window', 0), 'upper_window': row.get('upper_window', 0),
                         'prior_scale': row.get('prior_scale', default_prior_scale)})
    else:
        for h in holidays:
            if 'days' in h:                      # already normalised (from a blob)
                rows.append(dict(h))
                continue
            ds = h['ds'] if isinstance(h['ds'], (list, tuple, np.ndarray, pd.Series, pd.DatetimeIndex)) else [h['ds']]
            rows.append({'holiday': h['holiday'], 'ds': list(ds), 'lower_window': h.get('lower_window', 0),
                         'upper_window': h.get('upper_window', 0),
                         'prior_scale': h.get('prior_scale', default_prior_scale)})
    out, scales = [], {}
    for r in rows:
        if 'days' in r:
            days = [int(d) for d in r['days']]
        else:
            days = [int(v) for v in _day_number(pd.DatetimeIndex(pd.to_datetime(list(r['ds']))).asi8)]

        # fbprophet's make_holiday_features parses both windows inside ONE try: if either is missing /
        # NaN / not a number, both are 0
        try:
            lw, uw = int(r['lower_window']), int(r['upper_window'])
        except (TypeError, ValueError):
            lw, uw = 0, 0
        ps = r['prior_scale']
        ps = default_prior_scale if ps is None or (isinstance(ps, float) and np.isnan(ps)) else float(ps)
        if ps <= 0:
            raise ValueError('Prior scale must be > 0')
        name = str(r['holiday'])
        # fbprophet's validate_column_name: the separator of the generated column names and the names
        # of its own components are not allowed as holiday names
        if '_delim_' in name:
            raise ValueError('Name cannot contain "_delim_"')
        if name in RESERVED_NAMES:
            raise ValueError('Name "{}" is reserved.'.format(name))
        if name in scales and scales[name] != ps:
            raise ValueError('Holiday {} does not have consistent prior scale specification.'.format(name))
        scales[name] = ps
        out.append({'holiday': name, 'days': days, 'lower_window': lw, 'upper_window': uw, 'prior_scale': ps})
    return out


def holiday_columns(holidays_norm):
    """-> (names sorted, prior scale per column, days per column): column `names[i]` is 1 on the
    day numbers `days[i]` (sorted unique int64 array)."""
    occ, scale = {}, {}
    for h in holidays_norm:
        for off in range(h['lower_window'], h['upper_window'] + 1):
            key = '{}_delim_{}{}'.format(h['holiday'], '+' if off >= 0 else '-', abs(off))
            occ.setdefault(key, []).extend(d + off for d in h['days'])
            scale[key] = h['prior_scale']
    names = sorted(occ)
    return names, [scale[n] for n in names], [np.unique(np.asarray(occ[n], dtype=np.int64)) for n in names]


def holiday_matrix(ds_ns, days_per_column):
    """[n_columns][...shape of ds_ns] float64 indicator columns for timestamps ds_ns."""
    day = _day_number(ds_ns)
    out = np.zeros((len(days_per_column),) + day.shape)
    for i, d in enumerate(days_per_column):
        out[i] = np.isin(day, d)
    return out


def standardize_regressor(train_values, standardize='auto'):
    """fbprophet's rule for one extra regressor: returns (mu, std) to apply as (x - mu) / std to
    training AND future values; (0.0, 1.0) when the column is left as it is."""
    v = pd.to_numeric(pd.Series(np.asarray(train_values)))
    if v.isnull().any():
        raise ValueError('Found NaN in regressor column')
    uniq = set(v.unique().tolist())
    if len(uniq) < 2:
        standardize = False
    if isinstance(standardize, str) and standardize == 'auto':
        standardize = uniq != {0, 1}
    if not standardize:
        return 0.0, 1.0
    return float(v.mean()), float(v.std())
