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
SINK_PART_ROWS = 1 << 17           # rows per forecast CSV part (ProphetScorer.score)
SINK_WRITERS = 4                   # part files written at the same time
CONVERTED_COLUMNS = ['created_timestamp', 'series_id', 'dim_id', 'forecast_date',
                     'forecast_timestamp', 'forecast_quantity']


def _empty_forecasts():
    return pd.DataFrame(columns=FORECAST_COLUMNS)


def forecast_arrays(config):
    """The predict UDF on columns that never were a DataFrame: (series_id, dim_id, floor, cap, models) -> dict of the
    forecast frame's columns (series_id, dim_id int32; ds int64 ns; yhat int32; yhat_lower / yhat_upper when
    forecast.intervals), `periods` rows per series, or None when there is nothing to forecast.  models: a list of blobs
    (None = no model) or the uint8 [n][L] buffer of a model column whose blobs share one length
    (panel.model_column_buffer: a parquet column as it lies in memory, no Python object per series)."""

    def forecast_arrays_fn(sids, dids, floors, caps, blobs):
        if len(sids) == 0:
            return None
        frequency = config['forecast']['frequency']
        # 'W' in pandas date_range snaps to Sundays; the reference keeps the weekday (:59-62)
        if frequency == 'W':
            frequency = pd.offsets.Week()
        periods = int(config['forecast']['periods'])
        sids, dids = np.asarray(sids), np.asarray(dids)
        floors = np.asarray(floors, dtype=np.float64)
        caps = np.asarray(caps, dtype=np.float64)
        if not isinstance(blobs, np.ndarray):
            blobs = [None if (b is None or isinstance(b, float)) else b for b in blobs]
            for i, b in enumerate(blobs):
                if b is None:
                    # prophet_scorer.py:51-55
                    print(f"For series_id: {int(sids[i])}, dim_id: {int(dids[i])}, no model found")
        pieces = []
        from ..pipeline import Laps
        lap = Laps('forecast_arrays %d' % len(sids))
        # one launch per distinct model spec (series fitted together share it)
        for spec_dict, idx, rec in pk.load_models(blobs):
            lap('load_models')
            spec = fc.ModelSpec.from_dict(spec_dict)
            theta = np.zeros((len(idx), spec.theta_stride))
            theta[:, :rec['theta'].shape[1]] = rec['theta']
            grid = pk.grid_from_records(rec)
            lap('theta + grid')
            fut = pk.future_dates(rec['last_ds_ns'], periods, frequency)  # :64-66
            lap('future_dates')
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
            if len(fut) and (fut == fut[0]).all():
                fut = np.ascontiguousarray(fut[0])           # one future grid for the batch: one shared design table on the device
            yhat, yint = fc.predict(spec, theta, rec['y_scale'], grid, fut, floor=floor, cap=cap,
                                    extra_future=ex if (ex is None or fut.ndim == 2) else np.ascontiguousarray(ex[0]),
                                    want_int=True,      # :70-84
                                    devices=config.get('devices'))
            lap('predict')
            iv = None
            if (config.get('forecast') or {}).get('intervals'):
                # not in the reference's output (it drops yhat_lower / yhat_upper, :86): opt-in extra
                # columns; the random streams are keyed by (series_id, dim_id), so a series gets the
                # same interval whatever frame it arrives in
                fcfg = config['forecast']
                key = (sids[idx].astype(np.int64) << 32) ^ (dids[idx].astype(np.int64) & 0xffffffff)
                _, lo, hi = fc.predict_intervals(
                    spec, theta, rec['y_scale'], grid, fut, floor=floor, cap=cap,
                    extra_future=ex if (ex is None or fut.ndim == 2) else np.ascontiguousarray(ex[0]),
                    series_key=key, uncertainty_samples=int(fcfg.get('uncertainty_samples', 1000)),
                    interval_width=float(fcfg.get('interval_width', 0.8)), seed=int(fcfg.get('seed', 0)))
                iv = (lo, hi)
            for j in np.flatnonzero((np.trunc(yhat) < floor[:, None]).any(axis=1)):
                print(f"Negative forecast values found for series_id: {int(sids[idx[j]])}, "
                      f"dim_id: {int(dids[idx[j]])}")                    # :77-79
            if fut.ndim == 1:
                fut = np.broadcast_to(fut, (len(idx), periods))
            pieces.append((idx, fut, yint, iv))
            lap('negative check')
        if not pieces:
            return None

        def cat(parts):                     # one launch (the usual case): its array, not a copy of it
            return parts[0] if len(parts) == 1 else np.concatenate(parts)
        res = {
            'series_id': cat([np.repeat(sids[p[0]].astype('int32'), periods) for p in pieces]),
            'dim_id': cat([np.repeat(dids[p[0]].astype('int32'), periods) for p in pieces]),
            'ds': cat([np.ascontiguousarray(p[1]).reshape(-1) for p in pieces]),
            'yhat': cat([p[2].reshape(-1) for p in pieces]).astype('int32', copy=False),
        }
        if pieces[0][3] is not None:
            res['yhat_lower'] = np.concatenate([p[3][0].reshape(-1) for p in pieces])
            res['yhat_upper'] = np.concatenate([p[3][1].reshape(-1) for p in pieces])
        lap('columns')
        lap.__exit__()
        return res

    return forecast_arrays_fn


def forecast_panel(config):
    """Batched form of forecast_time_series: the model frame may hold any number of rows
    (one per fitted series); returns periods rows per series."""
    arrays = forecast_arrays(config)

    def forecast_panel_fn(pdf):
        if len(pdf.index) == 0:
            return _empty_forecasts()
        cols = arrays(pdf['series_id'].to_numpy(), pdf['dim_id'].to_numpy(), pdf['floor'].to_numpy(dtype=np.float64),
                      pdf['cap'].to_numpy(dtype=np.float64), pdf['model'].tolist())
        if cols is None:
            return _empty_forecasts()
        cols['ds'] = cols['ds'].view('datetime64[ns]')
        # (copy=True stacks the three int32 columns into one block: a copy of 900 000 x 3)
        return pd.DataFrame(cols, columns=FORECAST_COLUMNS + [c for c in ('yhat_lower', 'yhat_upper') if c in cols], copy=False)

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

    def _clear_forecasts(self):
        """-> function that waits until the previous run's files are gone (pipeline.clear_directory)"""
        from ..pipeline import clear_directory
        return clear_directory(self.config['io']['forecasts'])

    def _write_converted_part(self, k, cols, created_timestamp, lo=0, hi=None):
        """part-<k>.csv (with its own header, as each of Spark's part files has: :148-150) from rows [lo, hi) of forecast
        columns: series_id, dim_id, yhat int32 (or int64) and ds int64 ns / datetime64[ns]."""
        from .. import _lib
        if lo or hi is not None:
            cols = {c: np.asarray(cols[c])[lo:hi] for c in FORECAST_COLUMNS}
        ids = [np.asarray(cols[c]) for c in ('series_id', 'dim_id', 'yhat')]
        narrow = all(v.dtype == np.int32 for v in ids)          # forecast_panel's frame: the columns go as they are
        ids = [np.ascontiguousarray(v, dtype=np.int32 if narrow else np.int64) for v in ids]
        ds = np.asarray(cols['ds'])
        if ds.dtype != np.int64:
            ds = ds.view(np.int64) if ds.dtype == np.dtype('datetime64[ns]') else ds.astype('datetime64[ns]').astype(np.int64)
        ds = np.ascontiguousarray(ds)
        L = _lib.load()
        path = os.path.join(self.config['io']['forecasts'], 'part-%05d.csv' % k)
        rc = (L.tsf_csv_write_forecasts_i32 if narrow else L.tsf_csv_write_forecasts)(
            os.fsencode(path), created_timestamp.encode(), len(ds), ids[0].ctypes.data, ids[1].ctypes.data,
            ds.ctypes.data, ids[2].ctypes.data, 0)
        if rc != 0:
            raise OSError('tsf_csv_write_forecasts failed (%d) for %s' % (rc, path))

    def write_converted(self, forecast_df, created_timestamp=None):
        """convert_forecasts + write_forecasts in one native pass (tsf_csv_write_forecasts):
        the same file write_forecasts(convert_forecasts(forecast_df)) produces, formatted by
        threads straight from the forecast columns."""
        if created_timestamp is None:
            created_timestamp = datetime.now(timezone.utc).replace(microsecond=0).isoformat()
        wait = self._clear_forecasts()
        self._write_converted_part(0, {c: forecast_df[c].values for c in FORECAST_COLUMNS}, created_timestamp)
        wait()
        return created_timestamp

    def _model_chunks(self):
        """The model directory (:124-126) one parquet row group at a time: (series_id, dim_id, floor, cap, models)
        column tuples, models as panel.model_column_buffer gives them (the column's own bytes) or a list of blobs."""
        import pyarrow.parquet as pq
        path = self.config['io']['models']
        if os.path.isdir(path):
            parts = sorted(os.path.join(path, f) for f in os.listdir(path)
                           if not f.startswith(('_', '.')) and f.endswith('.parquet'))
        else:
            parts = [path]
        for part in parts:
            pf = pq.ParquetFile(part)
            for g in range(pf.num_row_groups):
                t = pf.read_row_group(g, columns=['series_id', 'dim_id', 'floor', 'cap', 'model'])
                if t.num_rows == 0:
                    continue
                models = pk.model_column_buffer(t.column('model'))
                if models is None:
                    models = t.column('model').to_pylist()
                yield (t.column('series_id').to_numpy(), t.column('dim_id').to_numpy(),
                       t.column('floor').to_numpy().astype(np.float64), t.column('cap').to_numpy().astype(np.float64),
                       models, t)           # (t keeps the buffers the views point into)

    @staticmethod
    def score(spark_session, config):
        """Forecast every model of io.models into io.forecasts (:152-165).  Round 6: a pipeline over the row groups of
        the model parts (time_series_spark_amd/pipeline.py): the models of group k + 1 are read while group k is on the
        GPU and the forecasts of group k - 1 are formatted and written -- one CSV part per group, each with the header,
        as Spark writes one per task.  convert_forecasts is a lazy plan in the reference (:162), fused by Spark into
        the write; here the native sink formats the converted rows straight from the forecast columns
        (_write_converted_part: the same rows write_forecasts(convert_forecasts(forecast_df)) gives, test_host.py), so
        the converted frame -- 900 000 rows of Python date strings for 10 000 series -- is never built.  Returns None, as
        the reference does."""
        from .. import pipeline
        scorer = ProphetScorer(config)
        created = datetime.now(timezone.utc).replace(microsecond=0).isoformat()
        arrays = forecast_arrays(scorer.config)
        gone = scorer._clear_forecasts()
        n_parts = []
        import concurrent.futures
        writers = concurrent.futures.ThreadPoolExecutor(max_workers=SINK_WRITERS)

        def sink(cols):
            # Buffered writes to ONE file take turns on its inode lock, however many threads issue them (measured: a
            # 370 000-row part left the sink's 32 formatting threads waiting for one another's pwrite, 15 of the scorer's
            # 25 ms on 10 000 series): a chunk goes out as several part files, SINK_PART_ROWS rows each, written side by side.
            if cols is not None:
                n = len(cols['yhat'])
                cuts = list(range(0, n, SINK_PART_ROWS)) + [n]
                futs = [writers.submit(scorer._write_converted_part, len(n_parts) + i, cols, created, a, b)
                        for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))]
                n_parts.extend([1] * len(futs))
                for f in futs:
                    f.result()

        try:
            pipeline.run_pipeline(scorer._model_chunks(), [lambda c: arrays(*c[:5]), sink])
        finally:
            writers.shutdown()
            gone()
        if not n_parts:             # no model at all: the header alone, as an empty frame's CSV has
            scorer._write_converted_part(0, {'series_id': np.zeros(0, np.int32), 'dim_id': np.zeros(0, np.int32),
                                             'ds': np.zeros(0, np.int64), 'yhat': np.zeros(0, np.int32)}, created)
