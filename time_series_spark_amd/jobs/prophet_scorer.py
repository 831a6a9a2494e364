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

from .. import forecaster as fc, panel as pk

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
        models, rows = [], []
        for i in range(len(pdf.index)):
            r = pdf.iloc[i]
            series_id, dim_id = int(r['series_id']), int(r['dim_id'])
            m = pk.load_model(r['model']) if r['model'] is not None else None
            if m is None:
                # prophet_scorer.py:51-55
                print(f"For series_id: {series_id}, dim_id: {dim_id}, no model found")
                continue
            models.append(m)
            rows.append((series_id, dim_id, float(r['floor']), float(r['cap'])))
        if not models:
            return _empty_forecasts()
        out = []
        # one launch per distinct model spec (series fitted together share it)
        groups = {}
        for i, m in enumerate(models):
            groups.setdefault(repr(sorted(m['spec'].items(), key=lambda kv: kv[0])), []).append(i)
        for _, idx in groups.items():
            spec = fc.ModelSpec.from_dict(models[idx[0]]['spec'])
            stride = spec.theta_stride
            theta = np.zeros((len(idx), stride))
            for j, i in enumerate(idx):
                theta[j, :len(models[i]['theta'])] = models[i]['theta']
            y_scale = np.array([models[i]['y_scale'] for i in idx])
            grid = pk.grid_from_models([models[i] for i in idx])
            last = np.array([models[i]['last_ds_ns'] for i in idx], dtype=np.int64)
            fut = pk.future_dates(last, periods, frequency)              # :64-66
            floor = np.array([rows[i][2] for i in idx])                  # :67
            cap = np.array([rows[i][3] for i in idx])                    # :68
            ex = np.zeros((len(idx), len(spec.extra), periods)) if spec.extra else None
            yhat, yint = fc.predict(spec, theta, y_scale, grid, fut, floor=floor, cap=cap,
                                    extra_future=ex, want_int=True)      # :70-84
            for j, i in enumerate(idx):
                if (np.trunc(yhat[j]) < floor[j]).any():
                    print(f"Negative forecast values found for series_id: {rows[i][0]}, "
                          f"dim_id: {rows[i][1]}")                       # :77-79
                out.append(pd.DataFrame({'series_id': rows[i][0], 'dim_id': rows[i][1],
                                         'ds': fut[j].astype('datetime64[ns]'), 'yhat': yint[j]}))
        res = pd.concat(out, ignore_index=True)[FORECAST_COLUMNS]
        res['series_id'] = res['series_id'].astype('int32')
        res['dim_id'] = res['dim_id'].astype('int32')
        res['yhat'] = res['yhat'].astype('int32')
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
        out = pd.DataFrame({
            'created_timestamp': created_timestamp,
            'series_id': forecast_df['series_id'].values,
            'dim_id': forecast_df['dim_id'].values,
            'forecast_date': [extract_date(pd.Timestamp(v).to_pydatetime())
                              for v in forecast_df['ds'].values],
            'forecast_timestamp': forecast_df['ds'].values,
            'forecast_quantity': forecast_df['yhat'].values,
        }, columns=CONVERTED_COLUMNS)
        return out

    def write_forecasts(self, output_df):
        """CSV with header, mode='overwrite' (:148-150)."""
        import shutil
        path = self.config['io']['forecasts']
        if os.path.isdir(path):
            shutil.rmtree(path)
        os.makedirs(path, exist_ok=True)
        output_df.to_csv(os.path.join(path, 'part-00000.csv'), index=False,
                         date_format='%Y-%m-%dT%H:%M:%S.%f')

    @staticmethod
    def score(spark_session, config):
        scorer = ProphetScorer(config)
        model_df = scorer.read_model_dataframe(spark_session)
        forecast_df = forecast_panel(scorer.config)(model_df)
        converted_df = scorer.convert_forecasts(forecast_df)
        scorer.write_forecasts(converted_df)
        return converted_df
