"""Predict side of the drop-in boundary: same names, config keys, column order and error
behaviour as /root/reference/src/jobs/prophet_scorer.py.

    forecast_time_series(config)      prophet_scorer.py:18-104   curried grouped-map function
    extract_date                                        :107-108
    ProphetScorer.read_model_dataframe                  :123-128
    ProphetScorer.convert_forecasts                     :130-145
    ProphetScorer.write_forecasts                       :147-150
    ProphetScorer.score                                 :152-165
"""
import logging
import os
from datetime import datetime, timezone

import numpy as np
import pandas as pd

from .. import features, forecaster as fc, panel as pk

FORECAST_COLUMNS = ['series_id', 'dim_id', 'ds', 'yhat']          # prophet_scorer.py:27-32
CONVERTED_COLUMNS = ['created_timestamp', 'series_id', 'dim_id', 'forecast_date',
                     'forecast_timestamp', 'forecast_quantity']


def _empty_forecasts():
    return pd.DataFrame(columns=FORECAST_COLUMNS)


def forecast_panel(config):
    """Batched form of forecast_time_series: the model frame may hold any number of rows
    (one per fitted series); returns periods rows per series."""

    def forecast_panel_fn(pdf):
        if len(pdf.index) == 0:
            return _empty_forecasts()
        frequency = config['forecast']['frequency']
        # 'W' in pandas date_range snaps to Sundays; the reference keeps the weekday (:59-62)
        if frequency == 'W':
            frequency = pd.offsets.Week()
        periods = int(config['forecast']['periods'])
        sids = pdf['series_id'].to_numpy()
        dids = pdf['dim_id'].to_numpy()
        floors = pdf['floor'].to_numpy(dtype=np.float64)
        caps = pdf['cap'].to_numpy(dtype=np.float64)
        blobs = [None if (b is None or isinstance(b, float)) else b for b in pdf['model'].tolist()]
        for i, b in enumerate(blobs):
            if b is None:
                # prophet_scorer.py:51-55
                print(f"For series_id: {int(sids[i])}, dim_id: {int(dids[i])}, no model found")
        pieces = []
        # one launch per distinct model spec (series fitted together share it)
        for spec_dict, idx, rec in pk.load_models(blobs):
            spec = fc.ModelSpec.from_dict(spec_dict)
            theta = np.zeros((len(idx), spec.theta_stride))
            theta[:, :rec['theta'].shape[1]] = rec['theta']
            grid = pk.grid_from_records(rec)
            fut = pk.future_dates(rec['last_ds_ns'], periods, frequency)  # :64-66
            floor, cap = floors[idx], caps[idx]                          # :67-68
            ex = None
            if spec.extra:
                # holiday columns for the future dates (fbprophet rebuilds them from the holidays
                # frame it keeps in the model); any other explicit column has no future values in
                # the reference's scorer (its future frame holds ds, floor, cap only): zeros
                ex = np.zeros((len(idx), len(spec.extra), periods))
                if spec.holidays:
                    names, _scales, days = features.holiday_columns(features.normalize_holidays(spec.holidays))
                    if [e['name'] for e in spec.extra[:len(names)]] != names:
                        # (not an assert: it is the only guard between a blob whose `holidays` and
                        # `extra` disagree and holiday indicators multiplied into the wrong coefficients)
                        raise ValueError('model blob: the holiday columns rebuilt from `holidays` do not match '
                                         'the leading entries of `extra`')
                    ex[:, :len(names), :] = np.moveaxis(features.holiday_matrix(fut, days), 0, 1)
            yhat, yint = fc.predict(spec, theta, rec['y_scale'], grid, fut, floor=floor, cap=cap,
                                    extra_future=ex, want_int=True,      # :70-84
                                    devices=config.get('devices'))
            iv = None
            if (config.get('forecast') or {}).get('intervals'):
                # not in the reference's output (it drops yhat_lower / yhat_upper, :86): opt-in extra
                # columns; the random streams are keyed by (series_id, dim_id), so a series gets the
                # same interval whatever frame it arrives in
                fcfg = config['forecast']
                key = (sids[idx].astype(np.int64) << 32) ^ (dids[idx].astype(np.int64) & 0xffffffff)
                _, lo, hi = fc.predict_intervals(
                    spec, theta, rec['y_scale'], grid, fut, floor=floor, cap=cap, extra_future=ex,
                    series_key=key, uncertainty_samples=int(fcfg.get('uncertainty_samples', 1000)),
                    interval_width=float(fcfg.get('interval_width', 0.8)), seed=int(fcfg.get('seed', 0)))
                iv = (lo, hi)
            for j in np.flatnonzero((np.trunc(yhat) < floor[:, None]).any(axis=1)):
                print(f"Negative forecast values found for series_id: {int(sids[idx[j]])}, "
                      f"dim_id: {int(dids[idx[j]])}")                    # :77-79
            pieces.append((idx, fut, yint, iv))
        if not pieces:
            return _empty_forecasts()

        def cat(parts):                     # one launch (the usual case): its array, not a copy of it
            return parts[0] if len(parts) == 1 else np.concatenate(parts)
        res = pd.DataFrame({
            'series_id': cat([np.repeat(sids[p[0]].astype('int32'), periods) for p in pieces]),
            'dim_id': cat([np.repeat(dids[p[0]].astype('int32'), periods) for p in pieces]),
            'ds': cat([p[1].reshape(-1) for p in pieces]).view('datetime64[ns]'),
            'yhat': cat([p[2].reshape(-1) for p in pieces]).astype('int32', copy=False),
        }, columns=FORECAST_COLUMNS, copy=False)     # (copy=True stacks the three int32 columns into one block: a copy of 900 000 x 3)
        if pieces[0][3] is not None:
            res['yhat_lower'] = np.concatenate([p[3][0].reshape(-1) for p in pieces])
            res['yhat_upper'] = np.concatenate([p[3][1].reshape(-1) for p in pieces])
        return res

    return forecast_panel_fn


def forecast_time_series(config):
    """Forecast using trained time series model (series_id, dim_id) -- prophet_scorer.py:18."""
    batched = forecast_panel(config)

    def forecast_time_series_udf(pdf):
        return batched(pdf)

    return forecast_time_series_udf


def extract_date(datetimestamp):
    return datetimestamp.date().strftime("%Y-%m-%d")                      # :107-108


class ProphetScorer:
    """Forecast quantities using trained models (prophet_scorer.py:114-165), Spark-free."""

    def __init__(self, config, logger=None):
        self.logger = logger or logging.getLogger(self.__class__.__name__)
        self.config = config

    def read_model_dataframe(self, spark=None):
        return pd.read_parquet(self.config['io']['models'])               # :124-126

    @staticmethod
    def convert_forecasts(forecast_df):
        created_timestamp = datetime.now(timezone.utc).replace(microsecond=0).isoformat()
        ds = forecast_df['ds'].values.astype('datetime64[ns]')
        # extract_date per distinct day (the column repeats a few hundred dates)
        inv, days = pd.factorize(ds.astype('datetime64[D]').astype(np.int64))
        names = np.array([extract_date(pd.Timestamp(int(d), unit='D').to_pydatetime()) for d in days],
                         dtype=object)
        out = pd.DataFrame({
            'created_timestamp': created_timestamp,
            'series_id': forecast_df['series_id'].values,
            'dim_id': forecast_df['dim_id'].values,
            'forecast_date': names[inv.reshape(-1)] if len(ds) else names,
            'forecast_timestamp': ds,
            'forecast_quantity': forecast_df['yhat'].values,
        }, columns=CONVERTED_COLUMNS)
        return out

    def write_forecasts(self, output_df):
        """CSV with header, mode='overwrite' (:148-150).  Timestamps are written the way
        Spark 2.4's CSV writer does by default (timestampFormat yyyy-MM-dd'T'HH:mm:ss.SSSXXX)
        with the wall times taken as UTC: 2002-12-28T22:00:00.000Z."""
        import shutil
        import pyarrow as pa
        import pyarrow.csv as pacsv
        path = self.config['io']['forecasts']
        if os.path.isdir(path):
            shutil.rmtree(path)
        os.makedirs(path, exist_ok=True)
        def text_column(codes, values):
            # few distinct strings, many rows: decode a dictionary inside arrow
            d = pa.DictionaryArray.from_arrays(pa.array(codes.astype(np.int32)), pa.array(values, type=pa.string()))
            return d.cast(pa.string())

        cols = {}
        for c in output_df.columns:
            v = output_df[c].values
            if np.issubdtype(v.dtype, np.datetime64):
                u, inv = np.unique(v.astype('datetime64[ns]'), return_inverse=True)
                cols[c] = text_column(inv.reshape(-1), [t + 'Z' for t in np.datetime_as_string(u, unit='ms')])
            elif v.dtype == object:
                codes, uniq = pd.factorize(v)
                cols[c] = text_column(codes, [str(x) for x in uniq])
            else:
                cols[c] = pa.array(v)
        with open(os.path.join(path, 'part-00000.csv'), 'wb') as f:
            f.write((','.join(output_df.columns) + '\n').encode())
            pacsv.write_csv(pa.table(cols), f,
                            write_options=pacsv.WriteOptions(include_header=False, quoting_style='none'))

    def write_converted(self, forecast_df, created_timestamp=None):
        """convert_forecasts + write_forecasts in one native pass (tsf_csv_write_forecasts):
        the same file write_forecasts(convert_forecasts(forecast_df)) produces, formatted by
        threads straight from the forecast columns."""
        import ctypes
        import shutil
        from .. import _lib
        if created_timestamp is None:
            created_timestamp = datetime.now(timezone.utc).replace(microsecond=0).isoformat()
        path = self.config['io']['forecasts']
        if os.path.isdir(path):
            shutil.rmtree(path)
        os.makedirs(path, exist_ok=True)
        ids = [forecast_df[c].values for c in ('series_id', 'dim_id', 'yhat')]
        narrow = all(v.dtype == np.int32 for v in ids)          # forecast_panel's frame: the columns go as they are
        ids = [np.ascontiguousarray(v, dtype=np.int32 if narrow else np.int64) for v in ids]
        ds = forecast_df['ds'].values
        ds = np.ascontiguousarray(ds.view(np.int64) if ds.dtype == np.dtype('datetime64[ns]')
                                  else ds.astype('datetime64[ns]').astype(np.int64))
        cols = [ids[0], ids[1], ds, ids[2]]
        L = _lib.load()
        rc = (L.tsf_csv_write_forecasts_i32 if narrow else L.tsf_csv_write_forecasts)(
            os.fsencode(os.path.join(path, 'part-00000.csv')), created_timestamp.encode(),
            len(forecast_df.index), *[c.ctypes.data for c in cols], 0)
        if rc != 0:
            raise OSError('tsf_csv_write_forecasts failed (%d) for %s' % (rc, path))
        return created_timestamp

    @staticmethod
    def score(spark_session, config):
        scorer = ProphetScorer(config)
        model_df = scorer.read_model_dataframe(spark_session)
        forecast_df = forecast_panel(scorer.config)(model_df)
        # convert_forecasts is a lazy plan in the reference (:162), fused by Spark into the write; here the native
        # sink formats the converted rows straight from the forecast columns (write_converted: the same file
        # write_forecasts(convert_forecasts(forecast_df)) gives, test_host.py), so the converted frame -- 900 000 rows
        # of Python date strings for 10 000 series, 17 ms -- is never built.  Returns None, as the reference does.
        scorer.write_converted(forecast_df)
