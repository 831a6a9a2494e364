"""CPU tests of the host layer: panel packing, fbprophet's auto-seasonality rules, model blob,
future frames, the reference's exact-value conversion test re-expressed without Spark
(/root/reference/tests/unit/prophet_scorer_test.py:70-80), CSV/parquet IO, and that the C-ABI
library loads and exports every symbol include/tsf.h declares (no compute: no GPU here)."""
import ctypes
import os
import re
from datetime import datetime

import numpy as np
import pandas as pd
import pytest

from tests import helpers
from time_series_spark_amd import _lib, forecaster as fc, panel as pk, parallel
from time_series_spark_amd.jobs import prophet_modeler as pm, prophet_scorer as ps


def test_library_loads_and_exports_every_declared_symbol(built):
    L = _lib.load()
    # include/tsf.h: the drop-in surface; include/tsf_dev.h: what tests and measurements need on top of it
    hdr = open(os.path.join(helpers.ROOT, 'include', 'tsf.h')).read()
    dev = open(os.path.join(helpers.ROOT, 'include', 'tsf_dev.h')).read()
    surface = set(re.findall(r'\b(tsf_[a-z_A-Z0-9]+)\s*\(', hdr))
    declared = surface | set(re.findall(r'\b(tsf_[a-z_A-Z0-9]+)\s*\(', dev))
    assert declared == set(_lib.EXPORTS)
    assert not surface & {'tsf_set_option', 'tsf_get_option', 'tsf_eval', 'tsf_design', 'tsf_profile_read'}
    for sym in declared:
        assert hasattr(L, sym), sym
    assert L.tsf_spec_size() == ctypes.sizeof(_lib.TsfSpec)
    s = _lib.default_spec()
    assert (s.n_changepoints, s.history, s.max_iter) == (25, 5, 10000)
    assert s.changepoint_range == 0.8 and s.changepoint_prior_scale == 0.05
    assert (s.init_alpha, s.tol_obj, s.tol_rel_obj, s.tol_grad, s.tol_rel_grad, s.tol_param) == \
        (1e-3, 1e-12, 1e4, 1e-8, 1e7, 1e-8)
    # evaluation form of the likelihood (include/tsf.h): auto, re-centre every 128 / at ratio 1
    assert (s.eval_form, s.recenter_every, s.recenter_ratio) == (_lib.EVAL_AUTO, 128, 1.0)
    # what a fit converges to: Stan's stopping rule by default (= Prophet.fit); the maximum a posteriori estimate on request
    assert (s.converge, s.map_max_iter, s.map_tol) == (_lib.CONVERGE_STAN, 10000, 1e-7)
    cm = fc.ModelSpec(growth='logistic', seasonalities=[helpers.WEEKLY], converge=_lib.CONVERGE_MAP, map_max_iter=500).to_c()
    assert (cm.converge, cm.map_max_iter) == (_lib.CONVERGE_MAP, 500)
    assert pm._spec_opts({'converge': 'map', 'map_tol': 1e-6}) == {'converge': _lib.CONVERGE_MAP, 'map_tol': 1e-6}
    assert pm._spec_opts({'converge': 'stan'}) == {'converge': _lib.CONVERGE_STAN}
    with pytest.raises(ValueError):
        pm._spec_opts({'converge': 'both'})
    c = fc.ModelSpec(growth='linear', seasonalities=[helpers.WEEKLY], eval_form=_lib.EVAL_RESIDUAL,
                     recenter_every=8).to_c()
    assert (c.eval_form, c.recenter_every) == (_lib.EVAL_RESIDUAL, 8)
    assert helpers.uses_quadratic_form(fc.ModelSpec(growth='linear', seasonalities=[helpers.WEEKLY]))
    assert not helpers.uses_quadratic_form(fc.ModelSpec(growth='logistic', seasonalities=[helpers.WEEKLY]))
    assert not helpers.uses_quadratic_form(fc.ModelSpec(growth='linear', seasonality_mode='multiplicative',
                                                        seasonalities=[helpers.WEEKLY]))


def test_no_cpu_fallback(built):
    # on a box without a GPU the product path must fail loudly, not fall back
    if _lib.load().tsf_device_count() > 0:
        pytest.skip('GPU present')
    with pytest.raises(_lib.TsfError):
        _lib.Context(0)


def test_pack_long_frame_sorts_drops_nan_keeps_duplicates():
    df = pd.DataFrame({
        'series_id': [1, 1, 1, 1, 2, 2, 1],
        'dim_id': [5, 5, 5, 5, 5, 5, 4],
        'ds': pd.to_datetime(['2020-01-03', '2020-01-01', '2020-01-02', '2020-01-02',
                              '2020-01-01', '2020-01-02', '2020-01-01']),
        'y': [3.0, 1.0, np.nan, 2.5, 10.0, 11.0, 7.0]})
    p = pk.pack_long_frame(df)
    assert p.N == 3
    assert list(p.keys['series_id']) == [1, 1, 2] and list(p.keys['dim_id']) == [4, 5, 5]
    assert list(p.offsets) == [0, 1, 4, 6]
    assert list(p.y) == [7.0, 1.0, 2.5, 3.0, 10.0, 11.0]
    assert not p.aligned
    df2 = pd.DataFrame({'series_id': [1] * 3 + [2] * 3, 'dim_id': 0,
                        'ds': list(pd.date_range('2020-01-01', periods=3)) * 2, 'y': np.arange(6.0)})
    p2 = pk.pack_long_frame(df2)
    assert p2.aligned and p2.y2d.shape == (2, 3)
    with pytest.raises(ValueError):
        pk.pack_long_frame(pd.DataFrame({'series_id': [1], 'dim_id': [1],
                                         'ds': pd.to_datetime(['2020-01-01']), 'y': [np.inf]}))


def test_trailing_null_rows_count_for_the_last_history_date_and_all_null_groups_are_reported():
    """fbprophet keeps null-y rows in history_dates (Prophet.fit) and make_future_dataframe starts
    at history_dates.max() (oracle/fbprophet_restated.py make_future_dataframe), so a series that
    ends in null-y rows forecasts from the LAST row, not from the last non-null one; a group with
    no non-null y at all makes Prophet.fit raise ValueError."""
    from oracle.fbprophet_restated import ProphetOracle
    days = pd.date_range('2020-01-01', periods=8)
    df = pd.DataFrame({
        'series_id': [1] * 8 + [2] * 3 + [3] * 2,
        'dim_id': 0,
        'ds': list(days) + list(days[:3]) + list(days[:2]),
        'y': [1.0, 2.0, 3.0, np.nan, 5.0, 6.0, np.nan, np.nan] + [1.0, np.nan, 3.0] + [np.nan, np.nan]})
    p = pk.pack_long_frame(df)
    assert p.N == 2 and list(p.lengths) == [5, 2]
    assert list(p.last_ds_all) == [days[7].value, days[2].value]          # not days[5] for series 1
    assert p.dropped_keys == [(3, 0)]
    m = ProphetOracle()
    m.stan_data(df[df['series_id'] == 1][['ds', 'y']])
    lit = m.make_future_dataframe(3, freq='D', include_history=False)['ds']
    assert list(pk.future_dates(p.last_ds_all[:1], 3, 'D')[0]) == [t.value for t in lit]
    # packed input without nulls: nothing to merge
    p2 = pk.pack_long_frame(df.dropna())
    assert list(p2.last_ds_all) == [days[5].value, days[2].value] and p2.dropped_keys == []


def test_group_with_only_null_y_raises_like_prophet_fit():
    from time_series_spark_amd.jobs import prophet_modeler as pm
    days = pd.date_range('2020-01-01', periods=4)
    df = pd.DataFrame({'series_id': [1] * 4 + [2] * 4, 'dim_id': 7, 'ds': list(days) * 2,
                       'y': [1.0, 2.0, 3.0, 4.0] + [np.nan] * 4})
    with pytest.raises(ValueError, match='less than 2 non-NaN rows'):
        pm.model_panel({'model': {'floor': 0, 'cap_multiplier': 1.1}})(df)


def test_auto_seasonality_rules():
    day = fc.DAY_NS
    names = lambda s: [x['name'] for x in s]
    assert names(fc.ModelSpec.auto_seasonalities(day * np.arange(730))) == ['weekly']      # F8
    assert names(fc.ModelSpec.auto_seasonalities(day * np.arange(731))) == ['yearly', 'weekly']
    assert names(fc.ModelSpec.auto_seasonalities(day * np.arange(731), yearly=False)) == ['weekly']
    assert names(fc.ModelSpec.auto_seasonalities(day * np.arange(90))) == ['weekly']
    assert names(fc.ModelSpec.auto_seasonalities(7 * day * np.arange(200))) == ['yearly']
    hourly = (day // 24) * np.arange(24 * 20)
    assert names(fc.ModelSpec.auto_seasonalities(hourly)) == ['weekly', 'daily']
    assert names(fc.ModelSpec.auto_seasonalities(day * np.arange(730), yearly=True)) == ['yearly', 'weekly']
    s = fc.ModelSpec.auto_seasonalities(day * np.arange(800))
    assert [(x['period'], x['fourier_order']) for x in s] == [(365.25, 10), (7, 3)]
    # the reference fixture: Thu-Sun 11:15 / 21:45 observations -> weekly + daily
    g = np.load(helpers.GOLDEN + '/fixture_751.npz')
    df = pd.DataFrame({'series_id': 751, 'dim_id': g['raw_dim_id'],
                       'ds': g['raw_ds_ns'].astype('datetime64[ns]'), 'y': g['raw_y']})
    p = pk.pack_long_frame(df)
    assert list(p.lengths) == [410, 406]
    span, min_dt, ymax = pk.per_series_stats(p)
    assert names(fc.ModelSpec.auto_from_stats(int(span[0]), int(min_dt[0]))) == ['weekly', 'daily']
    assert np.allclose(ymax * 1.1, [103591.4, 140054.2])


def test_model_blob_roundtrip():
    spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative',
                        seasonalities=[dict(helpers.WEEKLY)], tol_grad=1e-9)
    grid = np.zeros(1, dtype=_lib.GRID_DTYPE)
    grid[0]['start_ns'] = 123; grid[0]['t_scale_ns'] = 456; grid[0]['T'] = 100; grid[0]['S'] = 25
    grid[0]['t_change'][:25] = np.linspace(0.03, 0.8, 25)
    theta = np.arange(34.0)
    b = pk.dump_model(spec.to_dict(), theta, 2.5, grid[0], 999, 31, 77)
    m = pk.load_model(b)
    assert np.array_equal(m['theta'], theta) and m['y_scale'] == 2.5 and m['last_ds_ns'] == 999
    assert np.array_equal(m['t_change'], grid[0]['t_change'][:25])
    s2 = fc.ModelSpec.from_dict(m['spec'])
    assert s2.to_dict() == spec.to_dict() and s2.lbfgs == {'tol_grad': 1e-9}
    assert pk.load_model(None) is None
    with pytest.raises(ValueError):
        pk.load_model(b'nope' + b[4:])
    g2 = pk.grid_from_models([m])
    assert g2[0]['S'] == 25 and g2[0]['start_ns'] == 123


def test_future_dates_match_make_future_dataframe():
    last = pd.Timestamp('2002-12-28 21:45:00').value
    f = pk.future_dates([last], 40, '15min')[0].astype('datetime64[ns]')
    assert f[0] == np.datetime64('2002-12-28T22:00:00') and len(f) == 40
    assert f[-1] == np.datetime64('2002-12-28T22:00:00') + np.timedelta64(39 * 15, 'm')
    # 'W' would snap to Sundays; the reference substitutes pd.offsets.Week() (prophet_scorer.py:59-62)
    sat = pd.Timestamp('2002-12-28').value
    w = pk.future_dates([sat], 2, pd.offsets.Week())[0].astype('datetime64[ns]')
    assert list(w) == [np.datetime64('2003-01-04'), np.datetime64('2003-01-11')]
    d = pk.future_dates([sat], 3, 'D')[0].astype('datetime64[ns]').astype('datetime64[D]')
    assert list(d) == [np.datetime64('2002-12-29'), np.datetime64('2002-12-30'), np.datetime64('2002-12-31')]


def test_convert_forecasts():
    # /root/reference/tests/unit/prophet_scorer_test.py:55-80 without Spark
    fdf = pd.DataFrame({'series_id': np.array([101], dtype='int32'), 'dim_id': np.array([66], dtype='int32'),
                        'ds': [datetime.strptime('2015-07-05 10:15:00', '%Y-%m-%d %H:%M:%S')],
                        'yhat': np.array([873242], dtype='int32')})
    out = ps.ProphetScorer.convert_forecasts(fdf)
    timestamp_regex = re.compile(r'^([0-9]{4})-(1[0-2]|0[1-9])-(3[01]|0[1-9]|[12][0-9])T'
                                 r'(2[0-3]|[01][0-9]):([0-5][0-9]):([0-5][0-9])(\+00:00)$')
    row = out.iloc[0]
    assert timestamp_regex.match(row.iloc[0])
    assert row.iloc[1] == 101 and row.iloc[2] == 66
    assert row.iloc[3] == '2015-07-05'
    assert pd.Timestamp(row.iloc[4]).to_pydatetime() == datetime(2015, 7, 5, 10, 15)
    assert row.iloc[5] == 873242
    assert list(out.columns) == ['created_timestamp', 'series_id', 'dim_id', 'forecast_date',
                                 'forecast_timestamp', 'forecast_quantity']


def test_read_input_dataframe_hive_layout(tmp_path):
    # /root/reference/tests/unit/prophet_modeler_test.py:52-56: the fixture's layout and counts
    g = np.load(helpers.GOLDEN + '/fixture_751.npz')
    d = tmp_path / 'model-input' / 'series_id=751'
    d.mkdir(parents=True)
    pd.DataFrame({'dim_id': g['raw_dim_id'],
                  'ds': pd.Series(g['raw_ds_ns'].astype('datetime64[ns]')).dt.strftime('%Y-%m-%d %H:%M:%S'),
                  'y': g['raw_y']}).to_csv(d / 'sample-model-input.csv', header=False, index=False)
    modeler = pm.ProphetModeler({'io': {'input': str(tmp_path / 'model-input'), 'models': str(tmp_path / 'models')},
                                 'model': {'floor': 0, 'cap_multiplier': 1.1}})
    df = modeler.read_input_dataframe(None)
    assert list(df.columns) == ['series_id', 'dim_id', 'ds', 'y']
    assert df['series_id'].nunique() == 1 and df['dim_id'].nunique() == 2 and len(df) == 816
    assert str(df['y'].dtype) == 'int32' and str(df['ds'].dtype).startswith('datetime64')


def test_input_discovery_follows_spark_rules(tmp_path):
    """spark.read.csv(path) reads every non-hidden file under the path (prophet_modeler.py:102-116):
    part files without an extension count, `_SUCCESS` / `.crc` files and `_temporary` directories do
    not; .gz / .deflate parts are inflated by the reader (as Spark's Hadoop codecs do transparently), a part in
    another codec raises instead of being skipped silently, a corrupt stream raises naming the file."""
    root = tmp_path / 'in'
    for sid in (7, 12):
        d = root / ('series_id=%d' % sid)
        d.mkdir(parents=True)
        (d / 'part-00000').write_text('1,2020-01-01 00:00:00,5\n1,2020-01-02 00:00:00,6\n')
        (d / 'part-00001.csv').write_text('1,2020-01-03 00:00:00,7\n')
        (d / '.part-00000.crc').write_text('x')
    (root / '_SUCCESS').write_text('')
    (root / '_temporary').mkdir()
    (root / '_temporary' / 'part-00009').write_text('garbage')
    files, part = pm.find_model_input(str(root))
    assert [os.path.basename(f) for f in files] == ['part-00000', 'part-00001.csv'] * 2
    assert part == [7, 7, 12, 12]
    sid, did, ds, y = pm.read_model_input(files, str(root), part_sid=part)
    assert sorted(set(sid.tolist())) == [7, 12] and len(y) == 6 and y.sum() == 36
    import gzip
    import zlib
    more = '1,2020-01-04 00:00:00,8\n1,2020-01-05 00:00:00,\n1,2020-01-06 00:00:00,9\n'
    (root / 'series_id=7' / 'part-00002.csv.gz').write_bytes(gzip.compress(more[:24].encode()) + gzip.compress(more[24:].encode()))   # two members
    (root / 'series_id=12' / 'part-00002.deflate').write_bytes(zlib.compress(b'2,2020-01-04 00:00:00,100\n'))
    big = ''.join('3,2021-%02d-%02d 00:00:00,%d\n' % (1 + i // 28, 1 + i % 28, i) for i in range(300)) * 200   # > one output guess
    (root / 'series_id=12' / 'part-00003.gz').write_bytes(gzip.compress(big.encode()))
    files, part = pm.find_model_input(str(root))
    assert len(files) == 7
    sid, did, ds, y = pm.read_model_input(files, str(root), part_sid=part)
    assert len(y) == 6 + 3 + 1 + 60000 and np.nansum(y[sid == 7]) == 18 + 17 and np.isnan(y).sum() == 1
    assert np.nansum(y[(sid == 12) & (did == 2)]) == 100 and np.nansum(y[did == 3]) == 200 * sum(range(300))
    (root / 'series_id=7' / 'part-00004.csv.gz').write_bytes(gzip.compress(more.encode())[:-9])     # truncated
    with pytest.raises(OSError, match='corrupt or truncated compressed stream in .*part-00004.csv.gz'):
        pm.read_model_input(*pm.find_model_input(str(root))[:1], str(root))
    os.remove(root / 'series_id=7' / 'part-00004.csv.gz')
    # (round-4 advice) the codec follows the SUFFIX, as Spark's does: a plain part whose first bytes happen to look like
    # a zlib header ("x^") is parsed as text -- a malformed record in FAILFAST mode, not a "corrupt compressed stream"
    (root / 'series_id=7' / 'part-00006.csv').write_bytes(b'x^1,2020-02-01 00:00:00,5\n')
    with pytest.raises(ValueError) as ei:
        pm.read_model_input(*pm.find_model_input(str(root))[:1], str(root))
    assert 'compressed' not in str(ei.value) and 'part-00006.csv' in str(ei.value)
    os.remove(root / 'series_id=7' / 'part-00006.csv')
    # ... and the views the in-place reader hands out are read-only
    cols = pm.read_model_input_dir(str(root))
    assert not cols[3].flags.writeable
    (root / 'series_id=7' / 'part-00005.csv.bz2').write_bytes(b'BZh9')
    with pytest.raises(ValueError, match='compressed input file'):
        pm.find_model_input(str(root))


def test_native_discovery_and_in_place_reader_match_the_python_walk(tmp_path):
    """read_model_input_dir (tsf_csv_discover + tsf_csv_read on its path list, rows parsed straight into one block
    that numpy adopts) against read_model_input(*find_model_input(root)): the same rows in the same order on a tree
    with partition and plain directories, nested directories, hidden files, a symlinked directory, compressed
    parts, blank lines (the block then has gaps to close) and, in PERMISSIVE mode, dropped records; the same
    errors for a foreign codec, a non-integer partition value and a malformed line."""
    import gzip
    root = tmp_path / 'in'
    rng = np.random.default_rng(5)

    def lines(dim, n, start):
        return ''.join('%d,2021-%02d-%02d 00:00:00,%d\n' % (dim, 1 + (start + i) // 28 % 12, 1 + (start + i) % 28, rng.integers(1, 99))
                       for i in range(n))
    for sid in (12, 7, 100, 3):
        d = root / ('series_id=%d' % sid)
        d.mkdir(parents=True)
        (d / 'part-00000').write_text(lines(1, 40, sid))
        (d / 'part-00001.csv').write_text(lines(2, 5, sid) + '\n\n' + lines(2, 3, sid + 9) + '   \n')      # blank lines
        (d / '.part-00000.crc').write_text('x')
        (d / '_SUCCESS').write_text('')
    (root / 'series_id=7' / 'sub').mkdir()
    (root / 'series_id=7' / 'sub' / 'part-00002.gz').write_bytes(gzip.compress(lines(3, 6, 2).encode()))
    (root / '_temporary').mkdir()
    (root / '_temporary' / 'part-00009').write_text('garbage')
    (root / 'loose').mkdir()
    (root / 'loose' / 'a.csv').write_text(''.join('55,' + l for l in lines(4, 4, 0).splitlines(True)))        # 4 columns
    (tmp_path / 'elsewhere' / 'series_id=41').mkdir(parents=True)
    (tmp_path / 'elsewhere' / 'series_id=41' / 'p.csv').write_text(lines(5, 7, 1))
    os.symlink(tmp_path / 'elsewhere' / 'series_id=41', root / 'series_id=41')
    files, part = pm.find_model_input(str(root))
    want = pm.read_model_input(files, str(root), part_sid=part)
    got = pm.read_model_input_dir(str(root))
    assert len(got[3]) == 4 * 48 + 6 + 4 + 7
    for a, b in zip(want, got):
        assert a.dtype == b.dtype and np.array_equal(a, b, equal_nan=True)
    assert list(np.unique(got[0])) == [3, 7, 12, 41, 55, 100] and list(got[0][:48]) == [3] * 48       # partition order
    for nt in (1, 3):
        for a, b in zip(got, pm.read_model_input_dir(str(root), n_threads=nt)):
            assert np.array_equal(a, b, equal_nan=True)
    # the arrays outlive every reference to the native table but their own
    import gc
    y = got[3]
    del got, want
    gc.collect()
    assert y.sum() > 0
    # a single file as root; an empty directory
    one = pm.read_model_input_dir(str(root / 'loose' / 'a.csv'))
    assert list(one[0]) == [55] * 4 and list(one[1]) == [4] * 4
    (tmp_path / 'empty').mkdir()
    assert len(pm.read_model_input_dir(str(tmp_path / 'empty'))[3]) == 0
    # PERMISSIVE: dropped records leave gaps inside the block
    (root / 'series_id=3' / 'part-00003.csv').write_text('1,2021-01-01 00:00:00,5\n1,bad,6\nx,2021-01-03 00:00:00,7\n1,2021-01-04 00:00:00,8\n')
    with pytest.raises(ValueError, match=r'series_id=3/part-00003.csv line 2 '):
        pm.read_model_input_dir(str(root))
    stats = {}
    got = pm.read_model_input_dir(str(root), mode='PERMISSIVE', stats=stats)
    stats2 = {}
    files, part = pm.find_model_input(str(root))
    want = pm.read_model_input(files, str(root), part_sid=part, mode='PERMISSIVE', stats=stats2)
    assert stats == stats2 == {'malformed': 2}
    for a, b in zip(want, got):
        assert np.array_equal(a, b, equal_nan=True)
    assert got[3][got[0] == 3][-2:].tolist() == [5.0, 8.0]
    os.remove(root / 'series_id=3' / 'part-00003.csv')
    # errors
    (root / 'series_id=12' / 'part-9.csv.zst').write_bytes(b'xx')
    with pytest.raises(ValueError, match='compressed input file .*part-9.csv.zst'):
        pm.read_model_input_dir(str(root))
    os.remove(root / 'series_id=12' / 'part-9.csv.zst')
    (root / 'series_id=abc').mkdir()
    with pytest.raises(ValueError, match='series_id=abc'):
        pm.read_model_input_dir(str(root))


def test_shard_indices_partition():
    """series i on rank i mod world: the shards of all ranks partition the panel, sizes differ by at most one,
    shard_rank inverts it."""
    for n, w in [(10, 3), (10000, 8), (7, 8), (100000, 8), (1, 1)]:
        shards = [parallel.shard_indices(n, r, w) for r in range(w)]
        allidx = np.concatenate(shards)
        assert len(allidx) == n and np.array_equal(np.sort(allidx), np.arange(n))
        sizes = [len(x) for x in shards]
        assert max(sizes) - min(sizes) <= 1
        for i in range(0, n, max(1, n // 17)):
            assert i in shards[parallel.shard_rank(i, w)]


def test_spec_validation_messages():
    with pytest.raises(ValueError):
        fc.ModelSpec(growth='flat')
    with pytest.raises(ValueError):
        fc.ModelSpec(seasonality_mode='both')
    with pytest.raises(TypeError):
        fc.ModelSpec(tolerance=1)
    s = fc.ModelSpec(seasonalities=[dict(helpers.YEARLY), dict(helpers.WEEKLY)])
    assert s.K == 26 and s.theta_stride == 54


def test_drivers_usage_message(capsys):
    """/root/reference/src/modeler_spark_driver.py:9-11: wrong argument count -> message, exit 1."""
    from time_series_spark_amd import modeler_driver, scorer_driver
    assert modeler_driver.main(['prog']) == 1 and scorer_driver.main(['prog', 'a', 'b']) == 1
    assert capsys.readouterr().out.count('arg1 must be the config YAML') == 2


def test_group_by_grid_partitions_by_identical_timestamps():
    ds = pd.date_range('2020-01-01', periods=12, freq='D')
    rows = []
    grids = [ds, ds, ds + pd.Timedelta(hours=1), ds, ds[:7], ds[:7], ds[2:]]
    for sid, g in enumerate(grids):
        rows += [(sid, 0, t, float(sid + 1)) for t in g]
    p = pk.pack_long_frame(pd.DataFrame(rows, columns=['series_id', 'dim_id', 'ds', 'y']))
    groups, rest = pk.group_by_grid(p, np.arange(p.N))
    assert sorted(g.tolist() for g in groups) == [[0, 1, 3], [4, 5]] and rest.tolist() == [2, 6]
    groups, rest = pk.group_by_grid(p, np.array([0, 2, 4]))          # nobody shares within the subset
    assert groups == [] and rest.tolist() == [0, 2, 4]
    groups, rest = pk.group_by_grid(p, np.arange(p.N), min_group=3)
    assert [g.tolist() for g in groups] == [[0, 1, 3]] and rest.tolist() == [2, 4, 5, 6]
    # the job layer's policy: groups from min_group series on -- nothing is hashed or compared when no class of equal
    # (length, first, last) is that large -- but members that ALL share one grid stay a group whatever their number
    groups, rest = pk.group_by_grid(p, np.arange(p.N), min_group=4096, keep_single=True)
    assert groups == [] and rest.tolist() == list(range(p.N))
    groups, rest = pk.group_by_grid(p, np.array([0, 1, 3]), min_group=4096, keep_single=True)
    assert [g.tolist() for g in groups] == [[0, 1, 3]] and rest.tolist() == []
    groups, rest = pk.group_by_grid(p, np.array([0, 1, 3]), min_group=4096)
    assert groups == [] and rest.tolist() == [0, 1, 3]
    # same length, first and last timestamp, different interior: the cheap classes match, the rows do not
    g2 = ds.to_series().reset_index(drop=True)
    g2[5] = g2[5] + pd.Timedelta(hours=3)
    rows2 = [(0, 0, t, 1.0) for t in ds] + [(1, 0, t, 2.0) for t in g2]
    p2 = pk.pack_long_frame(pd.DataFrame(rows2, columns=['series_id', 'dim_id', 'ds', 'y']))
    groups, rest = pk.group_by_grid(p2, np.arange(2), keep_single=True)
    assert groups == [] and rest.tolist() == [0, 1]


@pytest.mark.parametrize('case', ['shuffled', 'grouped_unsorted', 'packed', 'nan_rows', 'single_rows',
                                  'empty', 'wide_keys'])
def test_native_packer_matches_numpy_order_contract(case):
    rng = np.random.default_rng(11)
    if case == 'empty':
        sid = did = ds = np.zeros(0, np.int64); y = np.zeros(0)
    else:
        G = 300
        lens = rng.integers(1, 40, G)
        if case == 'single_rows':
            lens[:] = 1
        sid = np.repeat(rng.integers(-5, 60, G), lens).astype(np.int64)
        did = np.repeat(rng.integers(0, 4, G), lens).astype(np.int64)
        if case == 'wide_keys':
            sid = sid * (1 << 40) + 7
            did = did - (1 << 50)
        n = len(sid)
        ds = rng.integers(0, 50, n).astype(np.int64) * 3600_000_000_000    # many ties
        y = rng.normal(size=n).round(2)
        if case in ('nan_rows', 'shuffled'):
            y[rng.random(n) < 0.2] = np.nan
            y[sid == sid[0]] = np.nan                                       # a series that vanishes
        if case in ('shuffled', 'nan_rows', 'wide_keys'):
            p = rng.permutation(n); sid, did, ds, y = sid[p], did[p], ds[p], y[p]
        if case == 'packed':
            o = np.lexsort((ds, did, sid)); sid, did, ds, y = sid[o], did[o], ds[o], y[o]
    ks, kd, off, dso, yo = helpers.pack_reference(sid, did, ds, y)
    for nt in (1, 4):
        p = pk.pack_rows(sid, did, ds, y, n_threads=nt)
        assert np.array_equal(p.keys['series_id'].values, ks) and np.array_equal(p.keys['dim_id'].values, kd)
        assert np.array_equal(p.offsets, off) and np.array_equal(p.ds_ns, dso) and np.array_equal(p.y, yo)
        # the statistics that come with it == the numpy statement of them
        span, min_dt, ymax = p.stats
        p.stats = None
        s2, m2, y2 = pk.per_series_stats(p)
        assert np.array_equal(span, s2) and np.array_equal(min_dt, m2) and np.array_equal(ymax, y2)
    if case == 'packed':
        assert p.ds_ns is ds or np.shares_memory(p.ds_ns, ds)               # no copy made


def test_native_csv_reader_matches_pandas(tmp_path):
    """tsf_csv_read (the Spark CSV reader's stand-in, prophet_modeler.py:102-116) against
    pandas.read_csv on the same files: partition directories, timestamp spellings, nulls."""
    rng = np.random.default_rng(5)
    root = tmp_path / 'in'
    want = []
    spell = ['%Y-%m-%d %H:%M:%S', '%Y-%m-%dT%H:%M:%S', '%Y-%m-%d %H:%M:%S.%f', '%Y-%m-%d', '%Y-%m-%d %H:%M']
    for k, sid in enumerate([751, 3, -4, 2000000000]):
        d = root / ('series_id=%d' % sid)
        d.mkdir(parents=True)
        n = 50 + 7 * k
        ts = pd.to_datetime('1969-06-01') + pd.to_timedelta(rng.integers(0, 40000, n) * 3600 + k, unit='s')
        if spell[k] == '%Y-%m-%d':
            ts = ts.normalize()
        if spell[k] == '%Y-%m-%d %H:%M':
            ts = ts.floor('min')
        if 'f' in spell[k]:
            ts = ts + pd.to_timedelta(rng.integers(0, 10 ** 6, n), unit='us')
        did = rng.integers(1, 4, n)
        q = rng.integers(-5, 10 ** 6, n).astype(float)
        if k == 1:
            q[::9] = np.nan                                  # null quantity
        lines = ['%d,%s,%s' % (a, b, '' if np.isnan(c) else '%d' % c)
                 for a, b, c in zip(did, ts.strftime(spell[k]), q)]
        eol = '\r\n' if k == 2 else '\n'
        text = eol.join(lines) + (eol if k != 3 else '') + (eol if k == 0 else '')   # blank line / no EOL at EOF
        (d / 'part-0.csv').write_bytes(text.encode())
        want.append(pd.DataFrame({'series_id': sid, 'dim_id': did, 'ds': ts.values, 'y': q}))
    want = pd.concat(want, ignore_index=True)
    files = sorted(str(p) for p in root.rglob('*.csv'))
    order = [int(f.split('series_id=')[1].split('/')[0]) for f in files]
    want = pd.concat([want[want.series_id == s] for s in order], ignore_index=True)
    for nt in (1, 3):
        sid, did, ds_ns, y = pm.read_model_input(files, str(root), n_threads=nt)
        assert np.array_equal(sid, want.series_id.values) and np.array_equal(did, want.dim_id.values)
        assert np.array_equal(ds_ns, want.ds.values.astype('datetime64[ns]').astype(np.int64))
        assert np.array_equal(y, want.y.values, equal_nan=True)
    # and the same bytes through pandas
    for f, s in zip(files, order):
        ref = pd.read_csv(f, header=None, names=['dim_id', 'ds', 'y'])
        got = pm.read_model_input([f], str(root))
        assert np.array_equal(got[2], pk.ds_to_ns(pd.to_datetime(ref['ds'])))
        assert np.array_equal(got[3], ref['y'].values.astype(float), equal_nan=True)
    # a file outside any partition directory carries series_id itself
    flat = tmp_path / 'flat.csv'
    flat.write_text('7,1,2020-02-29 00:00:00,5\n7,1,"2020-03-01 12:30:00", 6 \n')
    sid, did, ds_ns, y = pm.read_model_input([str(flat)], str(flat))
    assert list(sid) == [7, 7] and list(y) == [5, 6]
    assert ds_ns[1] == pd.Timestamp('2020-03-01 12:30:00').value
    df = pm.ProphetModeler({'io': {'input': str(flat)}}).read_input_dataframe()
    assert str(df['y'].dtype) == 'int32' and len(df) == 2
    # nulls survive as NaN (fbprophet drops them at fit time); the frame keeps them
    df = pm.ProphetModeler({'io': {'input': str(root)}}).read_input_dataframe()
    assert df['y'].isna().sum() == np.isnan(want.y.values).sum() and len(df) == len(want)
    # errors name the file and the line
    bad = tmp_path / 'bad' / 'series_id=1'
    bad.mkdir(parents=True)
    (bad / 'x.csv').write_text('1,2020-01-01 00:00:00,5\n1,2020-02-30 00:00:00,5\n')
    with pytest.raises(ValueError, match='line 2'):
        pm.read_model_input([str(bad / 'x.csv')], str(tmp_path / 'bad'))
    (bad / 'y.csv').write_text('1,2020-01-01 00:00:00\n')
    with pytest.raises(ValueError, match='line 1'):
        pm.read_model_input([str(bad / 'y.csv')], str(tmp_path / 'bad'))
    with pytest.raises(OSError):
        pm.read_model_input([str(bad / 'missing.csv')], str(tmp_path / 'bad'))


def test_write_forecasts_csv_and_model_parquet_round_trip(tmp_path):
    # prophet_scorer.py:147-150 (CSV with header) and prophet_modeler.py:118-125 / scorer :123-128
    cfg = {'io': {'forecasts': str(tmp_path / 'fc'), 'models': str(tmp_path / 'models')}}
    fdf = pd.DataFrame({'series_id': np.array([5, 5, 9], dtype='int32'), 'dim_id': np.array([1, 1, 2], dtype='int32'),
                        'ds': pd.to_datetime(['2002-12-28 22:00:00', '2002-12-28 22:15:00', '1969-12-31 23:59:59.5'], format='ISO8601'),
                        'yhat': np.array([10, -3, 7], dtype='int32')})
    sc = ps.ProphetScorer(cfg)
    conv = sc.convert_forecasts(fdf)
    assert list(conv['forecast_date']) == ['2002-12-28', '2002-12-28', '1969-12-31']
    sc.write_forecasts(conv)
    sc.write_forecasts(conv)                                   # overwrite, not append
    lines = open(tmp_path / 'fc' / 'part-00000.csv').read().splitlines()
    assert lines[0] == 'created_timestamp,series_id,dim_id,forecast_date,forecast_timestamp,forecast_quantity'
    assert len(lines) == 4
    f = lines[2].split(',')
    assert f[1:] == ['5', '1', '2002-12-28', '2002-12-28T22:15:00.000Z', '-3']
    assert lines[3].split(',')[4] == '1969-12-31T23:59:59.500Z'
    back = pd.read_csv(tmp_path / 'fc' / 'part-00000.csv')
    assert np.array_equal(back['forecast_quantity'].values, [10, -3, 7])
    # models: blobs survive parquet, floor/cap become float32 (prophet_modeler.py:35-36)
    spec = fc.ModelSpec(seasonalities=[dict(helpers.WEEKLY)])
    grid = np.zeros(1, dtype=_lib.GRID_DTYPE); grid['S'] = 3; grid['T'] = 50; grid['t_change'][0, :3] = [.2, .4, .6]
    blobs = pk.dump_models(spec.to_dict(), np.arange(2 * spec.theta_stride, dtype=float).reshape(2, -1),
                           [2.0, 3.0], grid, [100, 200], [31, 32], [7, 8])
    mdf = pd.DataFrame({'series_id': [1, 2], 'dim_id': [1, 1], 'floor': [0, 0], 'cap': [1.1, 16777217.0],
                        'model': blobs})
    mo = pm.ProphetModeler(cfg)
    mo.persist_models(mdf)
    got = sc.read_model_dataframe()
    assert str(got['cap'].dtype) == 'float32' and got['cap'][1] == np.float32(16777217.0)
    (sd, pos, rec), = pk.load_models(list(got['model']) + [None])
    assert list(pos) == [0, 1] and list(rec['y_scale']) == [2.0, 3.0] and list(rec['last_ds_ns']) == [100, 200]
    assert list(rec['status']) == [31, 32] and rec['theta'][1, -1] == 2 * spec.theta_stride - 1
    assert np.array_equal(pk.grid_from_records(rec)['t_change'][:, :3], [[.2, .4, .6]] * 2)
    assert sd == spec.to_dict()


def test_device_list_and_block_cuts(monkeypatch):
    monkeypatch.delenv('TSF_DEVICES', raising=False)
    assert fc.resolve_devices(None) is None and fc.resolve_devices([2]) is None
    assert fc.resolve_devices('0, 1,3') == [0, 1, 3] and fc.resolve_devices([0, 0]) == [0, 0]
    monkeypatch.setenv('TSF_DEVICES', '4,5')
    assert fc.resolve_devices(None) == [4, 5]
    assert list(fc._cuts(np.ones(10, int), 4)) == [0, 3, 5, 8, 10]
    c = fc._cuts(np.array([5, 5, 5, 5, 100, 5, 5, 5]), 3)      # one long series: a block may be empty
    assert c[0] == 0 and c[-1] == 8 and (np.diff(c) >= 0).all()
    rows = np.random.default_rng(0).integers(2, 900, 5000)
    c = fc._cuts(rows, 8)
    per = np.add.reduceat(rows, c[:-1])
    assert per.max() - per.min() <= 2 * rows.max()


def test_interleaved_device_split_round_trips():
    """In-process multi-GPU: series i -> device i mod G, merged back in series order."""
    N, G = 11, 3
    spec = fc.ModelSpec(seasonalities=[{'name': 'weekly', 'period': 7, 'fourier_order': 3}])
    full = np.arange(N, dtype=np.float64)
    parts = []
    for d in range(G):
        idx = full[d::G]
        parts.append(fc.FitResult(spec, np.stack([idx, idx + 0.5], axis=1), idx * 2, idx * 3,
                                  idx.astype(np.int32), idx.astype(np.int32) + 1, idx.astype(np.int32) + 2,
                                  np.zeros(1, dtype=_lib.GRID_DTYPE)))
    m = fc._merge_interleaved(spec, parts, N, G)
    assert np.array_equal(m.theta[:, 0], full) and np.array_equal(m.theta[:, 1], full + 0.5)
    assert np.array_equal(m.y_scale, 2 * full) and np.array_equal(m.status, full.astype(np.int32))
    assert np.array_equal(m.n_eval, full.astype(np.int32) + 2) and len(m.grid) == 1


def test_native_forecast_sink_writes_the_same_file_as_the_frame_path(tmp_path):
    rng = np.random.default_rng(2)
    n = 200000                                               # several 65 536-row blocks, 2+ threads
    ds = (np.datetime64('1965-03-01T00:00:00', 'ns')
          + rng.integers(0, 70 * 365 * 86400, n).astype('timedelta64[s]')
          + rng.integers(0, 1000, n).astype('timedelta64[ms]'))
    fdf = pd.DataFrame({'series_id': rng.integers(-3, 2 ** 31 - 1, n).astype('int32'),
                        'dim_id': rng.integers(0, 500, n).astype('int32'), 'ds': ds,
                        'yhat': rng.integers(-2 ** 31, 2 ** 31 - 1, n).astype('int32')})
    a = ps.ProphetScorer({'io': {'forecasts': str(tmp_path / 'a')}})
    b = ps.ProphetScorer({'io': {'forecasts': str(tmp_path / 'b')}})
    conv = a.convert_forecasts(fdf)
    a.write_forecasts(conv)
    b.write_converted(fdf, conv['created_timestamp'].iloc[0])
    assert open(tmp_path / 'a' / 'part-00000.csv', 'rb').read() == open(tmp_path / 'b' / 'part-00000.csv', 'rb').read()
    b.write_converted(fdf.iloc[:0])                          # empty: header only
    assert open(tmp_path / 'b' / 'part-00000.csv').read().count('\n') == 1


def test_native_packer_property_random_tables():
    """Random small tables (few keys, many ties, NaNs anywhere, any thread count): the native
    packer equals the numpy statement of the order contract."""
    from hypothesis import given, settings, strategies as st

    row = st.tuples(st.integers(-2, 3), st.integers(0, 2), st.integers(-3, 3),
                    st.one_of(st.just(float('nan')), st.floats(-5, 5, allow_nan=False, width=32)))

    @settings(max_examples=150, deadline=None)
    @given(st.lists(row, min_size=0, max_size=60), st.integers(1, 5))
    def check(rows, nt):
        a = np.array(rows, dtype=np.float64).reshape(-1, 4)
        sid, did, ds, y = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2].astype(np.int64), a[:, 3].copy()
        ks, kd, off, dso, yo = helpers.pack_reference(sid, did, ds, y)
        p = pk.pack_rows(sid, did, ds, y, n_threads=nt)
        assert np.array_equal(p.keys['series_id'].values, ks) and np.array_equal(p.keys['dim_id'].values, kd)
        assert np.array_equal(p.offsets, off) and np.array_equal(p.ds_ns, dso) and np.array_equal(p.y, yo)

    check()


def test_c_abi_is_plain_c_and_host_stages_run_from_c(built, tmp_path):
    """include/tsf.h compiles as strict C99 and a C program (no Python, no GPU) drives
    reader -> packer -> sink through the shared library."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / 'abi_host_stages')
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(['gcc', '-std=c99', '-pedantic', '-Wall', '-Wextra', '-Werror',
                           '-I', os.path.join(root, 'include'), os.path.join(root, 'tests', 'c', 'abi_host_stages.c'),
                           '-o', exe, '-L', libdir, '-ltsf_amd', '-Wl,-rpath,' + libdir])
    (tmp_path / 'in.csv').write_text('2,2020-01-03 00:00:00,5\n1,2020-01-02 00:00:00,9\n2,2020-01-01 12:00:00,\n'
                                     '1,2020-01-01 00:00:00,4\n2,2020-01-01 00:00:00,6\n')
    out = subprocess.check_output([exe, str(tmp_path / 'in.csv'), str(tmp_path / 'out.csv')]).decode()
    assert out.split() == ['rows_in=5', 'rows=4', 'series=2', 'identity=0', 'first_key=7/1',
                           'span0=86400000000000', 'min_dt0=86400000000000', 'ymax0=9.0']
    lines = (tmp_path / 'out.csv').read_text().splitlines()
    assert lines[1:] == ['2020-01-01T00:00:00+00:00,7,1,2020-01-01,2020-01-01T00:00:00.000Z,4',
                         '2020-01-01T00:00:00+00:00,7,1,2020-01-02,2020-01-02T00:00:00.000Z,9',
                         '2020-01-01T00:00:00+00:00,7,2,2020-01-01,2020-01-01T00:00:00.000Z,6',
                         '2020-01-01T00:00:00+00:00,7,2,2020-01-03,2020-01-03T00:00:00.000Z,5']


def test_job_layer_follows_fbprophets_optimiser_rule(monkeypatch):
    """Prophet.fit: Newton below 100 rows, L-BFGS from 100 rows on, and Newton once more for the
    series whose L-BFGS run ended in a RuntimeError (UPSTREAM-RECALL fbprophet 0.5).
    GPU calls replaced by recorders (the kernels themselves are compared with the oracle in
    tests/test_gpu_parity.py)."""
    calls = []

    def result(spec, N, grid_n, status):
        g = np.zeros(grid_n, dtype=_lib.GRID_DTYPE)
        g['S'] = 3
        return fc.FitResult(spec, np.zeros((N, spec.theta_stride)), np.ones(N), np.zeros(N),
                            np.asarray(status, np.int32), np.ones(N, np.int32), np.ones(N, np.int32), g)

    def fake_aligned(spec, ds_ns, y, floor=None, cap=None, extra=None, ctx=None, devices=None):
        algo = spec.lbfgs.get('algorithm')
        calls.append(('aligned', algo, y.shape))
        st = np.full(len(y), 31 if algo == _lib.ALGO_LBFGS else 60)
        if algo == _lib.ALGO_LBFGS:
            st[0] = -1                                       # first series: line search failed
        return result(spec, len(y), 1, st)

    def fake_ragged(spec, offsets, ds_ns, y, floor=None, cap=None, extra=None, ctx=None, devices=None):
        algo = spec.lbfgs.get('algorithm')
        calls.append(('ragged', algo, tuple(np.diff(offsets))))
        return result(spec, len(offsets) - 1, len(offsets) - 1, np.full(len(offsets) - 1, 60 if algo else 31))

    monkeypatch.setattr(fc, 'fit_aligned', fake_aligned)
    monkeypatch.setattr(fc, 'fit_ragged', fake_ragged)
    day = np.datetime64('2020-01-01', 'ns') + np.arange(150).astype('timedelta64[D]')
    rows = []
    for sid, T in enumerate([150, 150, 150, 60, 60, 99]):
        rows += [(sid, 1, d, 5 + (i * (sid + 1)) % 7) for i, d in enumerate(day[:T])]
    df = pd.DataFrame(rows, columns=['series_id', 'dim_id', 'ds', 'y'])
    cfg = {'model': {'floor': 0, 'cap_multiplier': 1.1, 'prophet': {'growth': 'linear', 'seasonality_mode': 'additive',
                                                                     'min_aligned_group': 2}}}     # (default 4 096: see below)
    out = pm.model_panel(cfg)(df)
    assert len(out) == 6                                     # the failed L-BFGS series came back through Newton
    assert calls == [('aligned', _lib.ALGO_LBFGS, (3, 150)),       # T >= 100 together
                     ('ragged', _lib.ALGO_NEWTON, (150,)),         # ... retry of the RuntimeError
                     ('aligned', _lib.ALGO_NEWTON, (2, 60)),       # T < 100: Newton, grouped by grid
                     ('ragged', _lib.ALGO_NEWTON, (99,))]
    (sd, pos, rec), = pk.load_models(list(out['model']))     # one spec for all: no optimiser in it
    assert 'algorithm' not in sd['lbfgs'] and list(rec['status']) == [60, 31, 31, 60, 60, 60]
    # default grouping policy: a group of series sharing their timestamps is a launch of its own only from 4 096 series
    # on (or when it is the whole bucket): the two 60-row series join the ragged call of the 99-row one
    calls.clear()
    del cfg['model']['prophet']['min_aligned_group']
    assert len(pm.model_panel(cfg)(df)) == 6
    assert calls == [('aligned', _lib.ALGO_LBFGS, (3, 150)), ('ragged', _lib.ALGO_NEWTON, (150,)),
                     ('ragged', _lib.ALGO_NEWTON, (60, 60, 99))]
    cfg['model']['prophet']['min_aligned_group'] = 2
    calls.clear()
    cfg['model']['prophet']['algorithm'] = 'lbfgs'
    out = pm.model_panel(cfg)(df)
    assert len(out) == 4 and all(c[1] == _lib.ALGO_LBFGS for c in calls)   # one failure per aligned call, no retry
    cfg['model']['prophet']['algorithm'] = 'bfgs'
    with pytest.raises(ValueError):
        pm.model_panel(cfg)(df)


def test_a_rerun_schedules_its_launches_from_the_previous_runs_models(monkeypatch, tmp_path):
    """ProphetModeler.model overwrites io.models -- with model.schedule_from_previous_models after reading what the
    previous run left there: every model blob carries its iteration count, and the next run hands those counts to the
    library as scheduling hints
    (tsf_set_cost_hints: longest fits first).  GPU calls replaced by recorders; that hints never change a result is
    tests/test_gpu_parity.py::test_cost_hints_change_the_order_of_the_launch_and_nothing_else."""
    seen = []

    def fake_aligned(spec, ds_ns, y, floor=None, cap=None, extra=None, ctx=None, devices=None, cost_hints=None):
        seen.append(None if cost_hints is None else np.asarray(cost_hints).copy())
        N = len(y)
        g = np.zeros(1, dtype=_lib.GRID_DTYPE)
        g['S'] = 3
        n_iter = (100 + 10 * np.arange(N)).astype(np.int32)          # series i "took" 100 + 10 i iterations
        return fc.FitResult(spec, np.zeros((N, spec.theta_stride)), np.ones(N), np.zeros(N),
                            np.full(N, 31, np.int32), n_iter, 2 * n_iter, g)

    monkeypatch.setattr(fc, 'fit_aligned', fake_aligned)
    root = tmp_path / 'in'
    day = np.datetime64('2020-01-01', 'ns') + np.arange(120).astype('timedelta64[D]')
    for sid in (3, 1, 2):
        d = root / ('series_id=%d' % sid)
        d.mkdir(parents=True)
        with open(d / 'part-00000.csv', 'w') as fh:
            for i, t in enumerate(day):
                fh.write('7,%s,%d\n' % (str(t)[:10], 5 + (i * sid) % 9))
    cfg = {'io': {'input': str(root), 'models': str(tmp_path / 'models')},
           'model': {'floor': 0, 'cap_multiplier': 1.1, 'schedule_from_previous_models': True,
                     'prophet': {'growth': 'linear', 'seasonality_mode': 'additive', 'algorithm': 'lbfgs'}}}
    first = pm.ProphetModeler.model(None, cfg)
    assert len(first) == 3 and seen == [None]                        # nothing to learn from yet
    prev = pm.previous_run_cost(cfg['io']['models'])
    assert sorted(prev['cost']) == [100, 110, 120]
    # persist_models left the counts beside the parquet part (read in place of 10 000 blobs); without the
    # sidecar, or with a part that is not the one it was written for, the blobs give the same frame
    side = os.path.join(cfg['io']['models'], pm.COST_SIDECAR)
    assert os.path.isfile(side)
    assert list(pd.read_parquet(cfg['io']['models']).columns) == ['series_id', 'dim_id', 'floor', 'cap', 'model']
    os.rename(side, side + '.away')
    slow = pm.previous_run_cost(cfg['io']['models'])
    os.rename(side + '.away', side)
    for c in ('series_id', 'dim_id', 'cost'):
        assert np.array_equal(prev[c].to_numpy(), slow[c].to_numpy())
    seen.clear()
    # second run: one series gone, one new
    import shutil
    shutil.rmtree(root / 'series_id=2')
    d = root / 'series_id=9'
    d.mkdir()
    with open(d / 'part-00000.csv', 'w') as fh:
        for i, t in enumerate(day):
            fh.write('7,%s,%d\n' % (str(t)[:10], 3 + i % 4))
    second = pm.ProphetModeler.model(None, cfg)
    assert len(second) == 3 and len(seen) == 1
    order = list(second['series_id'])
    want = {int(r.series_id): int(r.cost) for r in prev.itertuples()}
    med = int(np.median(prev['cost']))
    assert list(seen[0]) == [want.get(s, med) for s in order]        # known series: their count; the new one: the median
    seen.clear()
    cfg['model']['schedule_from_previous_models'] = False
    pm.ProphetModeler.model(None, cfg)
    assert seen == [None]
    # a foreign file where the models should be must not fail the run
    cfg['model']['schedule_from_previous_models'] = True
    with open(os.path.join(cfg['io']['models'], 'part-00000.parquet'), 'wb') as fh:
        fh.write(b'not parquet')
    seen.clear()
    assert len(pm.ProphetModeler.model(None, cfg)) == 3 and seen == [None]


def test_native_csv_reader_cuts_big_files_into_segments(tmp_path):
    """One 10 MB file is parsed by several threads (cut at line ends); same rows, same order,
    and a parse error deep inside still reports its line number in the file."""
    n = 400000
    rng = np.random.default_rng(9)
    ts = pd.Timestamp('2015-01-01') + pd.to_timedelta(np.arange(n) * 900, unit='s')
    q = rng.integers(0, 10 ** 6, n)
    sid = rng.integers(1, 50, n)
    text = '\n'.join('%d,7,%s,%d' % t for t in zip(sid, ts.strftime('%Y-%m-%d %H:%M:%S'), q)) + '\n'
    f = tmp_path / 'big.csv'
    f.write_text(text)
    assert f.stat().st_size > 9 * 2 ** 20
    for nt in (1, 4):
        s, d, ds_ns, y = pm.read_model_input([str(f)], str(f), n_threads=nt)
        assert np.array_equal(s, sid) and (d == 7).all() and np.array_equal(y, q.astype(float))
        assert np.array_equal(ds_ns, ts.values.astype('datetime64[ns]').astype(np.int64))
    lines = text.split('\n')
    lines[n - 5] = lines[n - 5].replace('-', '/', 1)         # a bad timestamp near the end
    f.write_text('\n'.join(lines))
    with pytest.raises(ValueError, match='line %d ' % (n - 4)):
        pm.read_model_input([str(f)], str(f), n_threads=4)


def test_holiday_columns_match_the_literal_make_holiday_features():
    """time_series_spark_amd/features.py against the method-by-method restatement of
    Prophet.make_holiday_features (oracle/fbprophet_restated.py): names, order, prior scales,
    indicator values -- intraday timestamps, duplicate timestamps, windows, a holiday with no
    matching row, dates before 1970."""
    from time_series_spark_amd import features
    from oracle.fbprophet_restated import ProphetOracle
    rng = np.random.default_rng(5)
    ds = pd.to_datetime(np.sort(rng.integers(-40, 400, 300)) * 86400 * 10 ** 9 + rng.integers(0, 86400, 300) * 10 ** 9)
    hol = pd.DataFrame({'holiday': ['xmas', 'xmas', 'sale', 'sale', 'never', 'old'],
                        'ds': pd.to_datetime(['1970-12-25', '1971-12-25', '1970-03-01', '1970-06-01', '1990-01-01',
                                              '1969-12-20']),
                        'lower_window': [-2, -2, 0, 0, -1, 0], 'upper_window': [1, 1, 3, 3, 1, 2],
                        'prior_scale': [5.0, 5.0, np.nan, np.nan, 2.0, 10.0]})
    m = ProphetOracle(holidays=hol, holidays_prior_scale=7.0)
    lit, lit_scales, _names = m.make_holiday_features(pd.Series(ds), hol)
    norm = features.normalize_holidays(hol, 7.0)
    names, scales, days = features.holiday_columns(norm)
    assert names == list(lit.columns) and scales == list(lit_scales)
    X = features.holiday_matrix(ds.asi8, days)
    assert np.array_equal(X, lit.values.T)
    assert X.sum() > 0 and not X[names.index('never_delim_+0')].any()
    # the normalised form survives JSON (it travels in the model blob) and a list-of-dicts config
    import json
    again = features.normalize_holidays(json.loads(json.dumps(norm)))
    assert features.holiday_columns(again)[0] == names
    cfg = [{'holiday': 'xmas', 'ds': ['1970-12-25', '1971-12-25'], 'lower_window': -2, 'upper_window': 1,
            'prior_scale': 5.0}]
    n2, s2, d2 = features.holiday_columns(features.normalize_holidays(cfg))
    i = [names.index(n) for n in n2]
    assert np.array_equal(features.holiday_matrix(ds.asi8, d2), X[i]) and s2 == [5.0] * 4
    with pytest.raises(ValueError, match='consistent prior scale'):
        features.normalize_holidays(pd.DataFrame({'holiday': ['a', 'a'], 'ds': pd.to_datetime(['2020-01-01'] * 2),
                                                  'prior_scale': [1.0, 2.0]}))


def test_regressor_standardisation_rule():
    from time_series_spark_amd import features
    from oracle.fbprophet_restated import ProphetOracle
    rng = np.random.default_rng(6)
    ds = pd.date_range('2020-01-01', periods=50)
    for vals in (rng.normal(3, 2, 50), rng.integers(0, 2, 50).astype(float), np.full(50, 4.0),
                 rng.integers(0, 3, 50).astype(float)):
        m = ProphetOracle()
        m.add_regressor('x')
        df = pd.DataFrame({'ds': ds, 'y': rng.normal(10, 1, 50), 'x': vals})
        out = m.setup_dataframe(df.copy(), initialize_scales=True)
        mu, sd = features.standardize_regressor(vals)
        assert np.allclose(out['x'].values, (vals - mu) / sd, rtol=0, atol=1e-15)
        assert (mu, sd) == (m.extra_regressors['x']['mu'], m.extra_regressors['x']['std'])


# ---- the grouped-map pandas_udf wiring against a stub pyspark ------------------------------------

def _stub_pyspark(monkeypatch):
    """A stand-in for the three pyspark names the reference imports for its UDFs
    (prophet_modeler.py:8-9, prophet_scorer.py:8-9): type classes that remember their name,
    StructType / StructField, and a pandas_udf(schema, functionType) decorator factory that records
    what it was given.  No JVM, no Spark -- it pins names, types, order and the function type."""
    import sys
    import types as pytypes
    calls = []

    class _T(object):
        def __init__(self):
            self.name = type(self).__name__

        def __eq__(self, other):
            return type(self) is type(other)

    tnames = ['BinaryType', 'FloatType', 'IntegerType', 'TimestampType', 'DoubleType', 'LongType', 'StringType']
    tmod = pytypes.ModuleType('pyspark.sql.types')
    for nm in tnames:
        setattr(tmod, nm, type(nm, (_T,), {}))

    class StructField(object):
        def __init__(self, name, dataType, nullable=True):
            self.name, self.dataType, self.nullable = name, dataType, nullable

    class StructType(object):
        def __init__(self, fields):
            self.fields = list(fields)
    tmod.StructField, tmod.StructType = StructField, StructType

    class PandasUDFType(object):
        SCALAR, GROUPED_MAP, GROUPED_AGG = 200, 201, 202

    def pandas_udf(schema, functionType):
        def deco(fn):
            calls.append((schema, functionType, fn))

            def wrapped(pdf):
                return fn(pdf)
            wrapped.returnType, wrapped.evalType, wrapped.func = schema, functionType, fn
            return wrapped
        return deco
    fmod = pytypes.ModuleType('pyspark.sql.functions')
    fmod.pandas_udf, fmod.PandasUDFType = pandas_udf, PandasUDFType
    smod = pytypes.ModuleType('pyspark.sql')
    smod.functions, smod.types = fmod, tmod
    pmod = pytypes.ModuleType('pyspark')
    pmod.sql = smod
    for name, mod in (('pyspark', pmod), ('pyspark.sql', smod), ('pyspark.sql.functions', fmod), ('pyspark.sql.types', tmod)):
        monkeypatch.setitem(sys.modules, name, mod)
    return calls, PandasUDFType


def _reference_schema(path, func):
    """[(field, SparkType)] of the output_schema literal inside `func` of a reference job file, read
    from the checkout when it is there (None on the GPU box)."""
    import re
    if not os.path.exists(path):
        return None
    src = open(path).read()
    body = src[src.index('def %s(' % func):]
    body = body[body.index('output_schema = StructType(['):]
    body = body[:body.index('])')]
    return re.findall(r"StructField\('(\w+)',\s*(\w+)\(\)", body)


def test_as_pandas_udf_registers_the_references_two_schemas(monkeypatch):
    """as_pandas_udf (jobs/prophet_modeler.py) must hand pyspark exactly what the reference's two
    decorators do (prophet_modeler.py:32-41, prophet_scorer.py:27-35): GROUPED_MAP, and the output
    schemas field for field -- names, Spark types, order, nullable.  pyspark is not installed here, so
    the wiring runs against a stub that records the call; where the reference checkout is present the
    expected schemas are read from its source instead of from this test."""
    from time_series_spark_amd.jobs import prophet_modeler as pm, prophet_scorer as ps
    calls, PandasUDFType = _stub_pyspark(monkeypatch)
    want_model = [('series_id', 'IntegerType'), ('dim_id', 'IntegerType'), ('floor', 'FloatType'),
                  ('cap', 'FloatType'), ('model', 'BinaryType')]                         # prophet_modeler.py:32-38
    want_fc = [('series_id', 'IntegerType'), ('dim_id', 'IntegerType'), ('ds', 'TimestampType'),
               ('yhat', 'IntegerType')]                                                  # prophet_scorer.py:27-32
    ref_m = _reference_schema('/root/reference/src/jobs/prophet_modeler.py', 'model_time_series')
    ref_f = _reference_schema('/root/reference/src/jobs/prophet_scorer.py', 'forecast_time_series')
    if ref_m is not None:
        assert ref_m == want_model and ref_f == want_fc
    fn_m = lambda pdf: pdf      # noqa: E731
    udf_m = pm.as_pandas_udf(fn_m)
    udf_f = pm.as_pandas_udf(fn_m, columns=ps.FORECAST_COLUMNS)
    assert len(calls) == 2
    for (schema, ftype, fn), want in zip(calls, (want_model, want_fc)):
        assert ftype == PandasUDFType.GROUPED_MAP and fn is fn_m
        assert [(f.name, f.dataType.name) for f in schema.fields] == want
        assert all(f.nullable is True for f in schema.fields)
    # the wrapped object still is the `pdf -> pdf` function
    df = pd.DataFrame({'a': [1]})
    assert udf_m(df) is df and udf_f(df) is df
    # and the frames the two job functions return carry exactly those columns, in that order, with
    # dtypes Arrow converts to the declared Spark types without a cast error
    assert pm.MODEL_OUTPUT_COLUMNS == [c for c, _ in want_model]
    assert ps.FORECAST_COLUMNS == [c for c, _ in want_fc]


def test_native_csv_reader_permissive_mode(tmp_path):
    """spark.read.csv(schema=MODEL_INPUT_SCHEMA) runs in Spark's default mode PERMISSIVE
    (prophet_modeler.py:109-114): a line that does not convert is a row of nulls (EVERY column, dim_id
    included), not an error.  The native reader offers that as layout '...?' / mode='PERMISSIVE': the record
    is dropped and counted -- it must never come out under a made-up key (round-3 advice: it came out as
    (series_id, dim_id = 0), which either joined series 0 or formed a one-row group whose fit raised) --,
    the good rows are untouched; FAILFAST (default) still names file and line."""
    root = tmp_path / 'in' / 'series_id=7'
    root.mkdir(parents=True)
    lines = ['1,2019-01-01 00:00:00,10', '1,not-a-date,11', '1,2019-01-03 00:00:00,twelve',
             '1,2019-01-04 00:00:00', 'x,2019-01-05 00:00:00,14', '1,2019-01-06 00:00:00,15', '1,2019-01-07 00:00:00,']
    (root / 'part-0.csv').write_text('\n'.join(lines) + '\n')
    f = [str(root / 'part-0.csv')]
    with pytest.raises(ValueError, match=r'part-0.csv line 2 '):
        pm.read_model_input(f, str(tmp_path / 'in'))
    stats = {}
    sid, did, ds_ns, y = pm.read_model_input(f, str(tmp_path / 'in'), mode='PERMISSIVE', stats=stats)
    assert stats == {'malformed': 4} and len(y) == 3 and (sid == 7).all() and (did == 1).all()
    assert list(y[:2]) == [10.0, 15.0] and np.isnan(y[2])      # (the last line: an empty quantity is a plain null, a valid record)
    assert list(ds_ns[:2].astype('datetime64[ns]').astype(str)) == ['2019-01-01T00:00:00.000000000', '2019-01-06T00:00:00.000000000']
    # through the packer: only the rows fbprophet would keep, and no key that is not in the file
    p = pk.pack_rows(sid, did, ds_ns, y, key_dtypes=(np.int32, np.int32))
    assert p.N == 1 and list(p.y) == [10.0, 15.0] and len(p.dropped_keys) == 0
    with pytest.raises(ValueError, match='mode must be'):
        pm.read_model_input(f, str(tmp_path / 'in'), mode='DROPMALFORMED')


def test_holiday_windows_and_names_follow_fbprophet():
    """fbprophet's make_holiday_features parses lower_window and upper_window in one try (either one
    unusable -> both 0) and refuses names that contain the column-name separator or are reserved."""
    from time_series_spark_amd import features
    h = features.normalize_holidays([{'holiday': 'a', 'ds': ['2019-01-01'], 'lower_window': float('nan'), 'upper_window': 2},
                                     {'holiday': 'b', 'ds': ['2019-02-01'], 'lower_window': -1, 'upper_window': 1}])
    assert (h[0]['lower_window'], h[0]['upper_window']) == (0, 0) and (h[1]['lower_window'], h[1]['upper_window']) == (-1, 1)
    names, _, _ = features.holiday_columns(h)
    assert names == ['a_delim_+0', 'b_delim_+0', 'b_delim_+1', 'b_delim_-1']
    with pytest.raises(ValueError, match='_delim_'):
        features.normalize_holidays([{'holiday': 'x_delim_y', 'ds': ['2019-01-01']}])
    with pytest.raises(ValueError, match='reserved'):
        features.normalize_holidays([{'holiday': 'weekly', 'ds': ['2019-01-01']}])


def test_model_frames_with_iteration_counts_concatenate():
    """Round-4 advice: the frame model_panel returns carries its iteration counts in DataFrame.attrs (scheduling hints
    for persist_models' side file); pandas compares attrs with == when frames are concatenated, and a bare ndarray of
    more than one element raised there.  The counts are one comparable value now."""
    from time_series_spark_amd.jobs import prophet_modeler as pm
    frames = []
    for k in range(2):
        f = pd.DataFrame({'series_id': np.arange(3, dtype=np.int32) + 10 * k, 'dim_id': np.zeros(3, dtype=np.int32)})
        f.attrs['tsf_cost'] = pm._CostVector(np.array([5, 6, 7]) + k)
        frames.append(f)
    both = pd.concat(frames, ignore_index=True)            # (used to raise: truth value of an array is ambiguous)
    assert len(both) == 6
    same = pd.concat([frames[0], frames[0].copy()], ignore_index=True)
    assert len(same) == 6
    assert np.array_equal(np.asarray(frames[1].attrs['tsf_cost'], dtype=np.int64), [6, 7, 8]) and len(frames[1].attrs['tsf_cost']) == 3


def test_no_stream_ordered_allocation_in_product_paths():
    """profiles/r05_mallocasync/README.md: on ROCm 7.2.0 a gigabyte-sized hipMallocAsync block freed with hipFreeAsync
    behind the launches that use it is released while they still run (default pool release threshold; standalone
    reproducer tools/probes/mallocasync_probe.hip) -- the cause of round 3's intermittently wrong Newton fits.  The
    library therefore takes every scratch buffer from cached hipMalloc blocks of the context; the only stream-ordered
    allocation left is the debug path behind TSF_OPT_DEBUG_ASYNC_SCRATCH in launch_newton_batch."""
    import glob
    import re
    root = os.path.join(helpers.ROOT, 'time_series_spark_amd', 'csrc')
    hits = []
    for f in sorted(glob.glob(os.path.join(root, '*'))):
        if not f.endswith(('.h', '.hip', '.inc', '.cpp')):
            continue
        src = open(f).read()
        code = re.sub(r'//[^\n]*', '', src)                 # comments may talk about it
        for m in re.finditer(r'hip(Malloc|Free)Async|hipMallocFromPoolAsync', code):
            hits.append((os.path.basename(f), code[:m.start()].count('\n') + 1))
    files = {h[0] for h in hits}
    assert files <= {'tsf_inst_quad.hip'}, hits
    q = open(os.path.join(root, 'tsf_inst_quad.hip')).read()
    # ... and there only under the debug switch
    assert 'dbg_async & 1' in q and q.count('hipMallocAsync(') == 1 and q.count('hipFreeAsync(') == 1


# ---- round 6: the jobs as pipelines, blobs and panel flags written by the library ---------------------------------

def _blob_reference(spec_dict, theta, y_scale, grid, last_ds_ns, status, n_iter):
    """numpy statement of the blob layout (panel.py, version 2): what tsf_model_blobs must write."""
    theta = np.atleast_2d(np.asarray(theta, dtype=np.float64))
    N, nth = theta.shape
    ntc = int(grid['S'].max()) if len(grid) else 0
    rec = np.zeros(N, dtype=pk._rec_dtype(nth, ntc))
    rec['y_scale'] = y_scale
    for f in ('start_ns', 't_scale_ns', 'T', 'S', 'i1', 'NT'):
        rec[f] = grid[f]
    rec['last_ds_ns'], rec['status'], rec['n_iter'] = last_ds_ns, status, n_iter
    rec['n_theta'], rec['n_tchange'] = nth, ntc
    rec['theta'] = theta
    rec['t_change'] = grid['t_change'][:, :ntc]
    body, L, pre = rec.tobytes(), rec.dtype.itemsize, pk._prefix(spec_dict)
    return [pre + body[i * L:(i + 1) * L] for i in range(N)]


def test_model_blobs_written_by_the_library_match_the_numpy_layout(built, tmp_path):
    rng = np.random.default_rng(3)
    spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=[helpers.WEEKLY])
    for N, n_grids in ((1, 1), (5, 1), (5, 5), (3000, 1), (3000, 3000)):
        grid = np.zeros(n_grids, dtype=_lib.GRID_DTYPE)
        grid['S'] = rng.integers(1, 26, n_grids)
        grid['T'], grid['NT'], grid['i1'] = 730, 12, 729
        grid['start_ns'] = rng.integers(0, 2 ** 60, n_grids)
        grid['t_scale_ns'] = rng.integers(1, 2 ** 50, n_grids)
        grid['t_change'] = rng.random((n_grids, _lib.MAX_S + 4))
        theta = rng.standard_normal((N, spec.theta_stride))
        args = (spec.to_dict(), theta, rng.random(N) + 1, grid, rng.integers(0, 2 ** 60, N),
                rng.integers(-3, 40, N).astype(np.int32), rng.integers(1, 9000, N).astype(np.int32))
        want = _blob_reference(*args)
        buf = pk.dump_models_buffer(*args)
        assert buf.dtype == np.uint8 and buf.shape == (N, len(want[0]))
        assert pk.dump_models(*args) == want and [bytes(r) for r in buf] == want
        # the buffer is an Arrow binary column as it stands, survives parquet, and loads without a Python object per series
        import pyarrow as pa
        import pyarrow.parquet as pq
        col = pk.model_column_arrow(buf)
        assert col.to_pylist() == want
        pq.write_table(pa.table({'model': col}), str(tmp_path / 'm.parquet'), row_group_size=max(1, N // 3))
        back = pq.read_table(str(tmp_path / 'm.parquet')).column('model')
        view = pk.model_column_buffer(back)
        assert view is not None and np.array_equal(view, buf)
        for source in (buf, want, view):
            (sd, pos, rec), = pk.load_models(source)
            assert sd == spec.to_dict() and np.array_equal(pos, np.arange(N))
            assert np.array_equal(rec['theta'], theta) and np.array_equal(rec['status'], args[5])
    # blobs of different lengths / a null: no buffer view, the list path takes over
    import pyarrow as pa
    assert pk.model_column_buffer(pa.array([b'ab', b'abc'])) is None
    assert pk.model_column_buffer(pa.array([b'ab', None])) is None


def test_packer_reports_calendar_infinity_and_integer_flags(built):
    day = np.int64(86400) * 10 ** 9
    sid = np.repeat(np.arange(4, dtype=np.int64), 5)
    did = np.ones(20, dtype=np.int64)
    ds = np.tile(np.arange(5, dtype=np.int64) * day, 4)
    y = np.arange(20, dtype=np.float64)
    p = pk.pack_rows(sid, did, ds, y)
    assert p.aligned and not p.has_inf and p.integral and p.y2d.shape == (4, 5)
    assert np.array_equal(p.ds_grid, ds[:5])
    ds2 = ds.copy()
    ds2[17] += 1                                       # one timestamp of the last series moved
    p = pk.pack_rows(sid, did, ds2, y)
    assert not p.aligned and p.y2d is None
    y2 = y.copy()
    y2[3] = 0.5
    y2[11] = np.inf
    p = pk.pack_rows(sid, did, ds, y2)
    assert p.aligned and p.has_inf and not p.integral
    y3 = y.copy()
    y3[4] = 2.0 ** 31                                  # an integer that does not fit int32
    assert not pk.pack_rows(sid, did, ds, y3).integral
    # the flags describe the PACKED rows: shuffled input, a null row dropped from one series -> lengths differ
    o = np.random.default_rng(0).permutation(20)
    p = pk.pack_rows(sid[o], did[o], ds[o], y[o])
    assert p.aligned and p.integral and np.array_equal(p.y2d, y.reshape(4, 5))
    y4 = y.copy()
    y4[7] = np.nan
    p = pk.pack_rows(sid, did, ds, y4)
    assert not p.aligned and list(p.lengths) == [5, 4, 5, 5]
    # several Python threads in the packer at once (the jobs' pipeline stages share the library's thread pool)
    import concurrent.futures
    big_sid = np.repeat(np.arange(3000, dtype=np.int64), 40)
    big_ds = np.tile(np.arange(40, dtype=np.int64) * day, 3000)
    big_y = np.arange(120000, dtype=np.float64)
    with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
        got = list(ex.map(lambda k: pk.pack_rows(big_sid, np.full(120000, k, np.int64), big_ds, big_y + k), range(12)))
    for k, g in enumerate(got):
        assert g.N == 3000 and g.aligned and np.array_equal(g.y2d, (big_y + k).reshape(3000, 40))
        assert (g.keys['dim_id'] == k).all() and np.array_equal(g.stats[2], (big_y + k).reshape(3000, 40).max(axis=1))


def test_pipeline_keeps_order_and_hands_the_first_error_to_the_caller():
    from time_series_spark_amd import pipeline
    import time as _t
    seen = []

    def slow(x):
        _t.sleep(0.002 * (x % 3))
        return x * 10

    out = pipeline.run_pipeline(iter(range(17)), [slow, lambda v: (seen.append(v), v + 1)[1]], depth=2)
    assert out == [x * 10 + 1 for x in range(17)] and seen == [x * 10 for x in range(17)]
    assert pipeline.run_pipeline(iter(()), [slow]) == []

    def bad(x):
        if x == 5:
            raise KeyError('chunk 5')
        return x

    with pytest.raises(KeyError):
        pipeline.run_pipeline(iter(range(100)), [bad, slow])

    def source():
        yield 1
        raise OSError('cannot list')

    with pytest.raises(OSError):
        pipeline.run_pipeline(source(), [slow])


def test_clear_directory_moves_the_old_run_aside_and_deletes_it(tmp_path):
    from time_series_spark_amd import pipeline
    d = tmp_path / 'models'
    d.mkdir()
    (d / 'part-00000.parquet').write_bytes(b'x' * 1000)
    wait = pipeline.clear_directory(str(d))
    assert d.is_dir() and list(d.iterdir()) == []
    (d / 'new').write_text('1')
    wait()
    assert sorted(p.name for p in tmp_path.iterdir()) == ['models'] and (d / 'new').exists()
    assert pipeline.clear_directory(str(tmp_path / 'fresh'))() is None and (tmp_path / 'fresh').is_dir()


def _fake_gpu(monkeypatch):
    """Stand-ins for the GPU calls whose results depend on the data, so that 'the same models' means something."""
    def fit(spec, N, grid_n, ymean, T):
        g = np.zeros(grid_n, dtype=_lib.GRID_DTYPE)
        g['S'], g['T'], g['NT'] = 3, T, (T + 63) // 64
        g['t_scale_ns'] = 1
        th = np.zeros((N, spec.theta_stride))
        th[:, 0] = ymean
        return fc.FitResult(spec, th, np.asarray(ymean) + 1.0, np.zeros(N), np.full(N, 31, np.int32),
                            (np.asarray(ymean) % 7 + 1).astype(np.int32), np.ones(N, np.int32), g)

    def fake_aligned(spec, ds_ns, y, floor=None, cap=None, extra=None, ctx=None, devices=None, cost_hints=None):
        return fit(spec, len(y), 1, np.asarray(y, dtype=np.float64).mean(axis=1), y.shape[1])

    def fake_ragged(spec, offsets, ds_ns, y, floor=None, cap=None, extra=None, ctx=None, devices=None, cost_hints=None):
        m = np.add.reduceat(np.asarray(y, dtype=np.float64), offsets[:-1]) / np.diff(offsets)
        return fit(spec, len(offsets) - 1, len(offsets) - 1, m, int(np.diff(offsets).max()))

    def fake_predict(spec, theta, y_scale, grid, fut, floor=None, cap=None, extra_future=None, want_int=False, ctx=None,
                     devices=None):
        H = np.shape(fut)[-1]
        yh = theta[:, :1] + np.arange(H)[None, :] + 0.5
        return (yh, np.maximum(np.trunc(yh), 0).astype(np.int32)) if want_int else yh

    monkeypatch.setattr(fc, 'fit_aligned', fake_aligned)
    monkeypatch.setattr(fc, 'fit_ragged', fake_ragged)
    monkeypatch.setattr(fc, 'predict', fake_predict)


def _write_hive_input(root, n_series, T, two_dims=()):
    day = np.datetime64('2020-01-01T00:00:00', 's')
    for s in range(n_series):
        d = root / ('series_id=%d' % (s * 3 + 1))
        d.mkdir(parents=True)
        rows = []
        for dim in ((1, 2) if s in two_dims else (1,)):
            rows += ['%d,%s,%d' % (dim, str(day + np.timedelta64(i, 'D')).replace('T', ' '), 10 + (s * 7 + i * dim) % 23)
                     for i in range(T - (s % 2))]
        (d / 'part-00000.csv').write_text('\n'.join(rows) + '\n')


def test_jobs_run_as_pipelines_over_chunks_with_the_same_results(built, tmp_path, monkeypatch, capsys):
    """ProphetModeler.model over ranges of partition directories (io.chunks) and ProphetScorer.score over the row groups
    of the model parts write what the whole-input runs write: same models (series by series), same forecast rows."""
    _fake_gpu(monkeypatch)
    _write_hive_input(tmp_path / 'in', 11, 120, two_dims=(2, 5))
    base = {'model': {'floor': 0, 'cap_multiplier': 1.1, 'prophet': {'growth': 'linear', 'seasonality_mode': 'additive'}}}
    out = {}
    for tag, chunks in (('whole', 1), ('chunked', 4), ('auto', 'auto')):
        cfg = dict(base, io={'input': str(tmp_path / 'in'), 'models': str(tmp_path / ('m_' + tag)), 'chunks': chunks})
        frame = pm.ProphetModeler.model(None, cfg)
        assert pm.ProphetModeler.model(None, cfg, return_frame=False) is None          # (second run: overwrites, uses the hints)
        parts = sorted(f for f in os.listdir(cfg['io']['models']) if f.endswith('.parquet'))
        assert len(parts) == (4 if tag == 'chunked' else 1)
        assert [f for f in os.listdir(tmp_path) if '.old-' in f] == []                # nothing of the overwritten run left
        back = pd.read_parquet(cfg['io']['models']).sort_values(['series_id', 'dim_id']).reset_index(drop=True)
        assert list(back.columns) == pm.MODEL_OUTPUT_COLUMNS and len(back) == 13
        assert (back.dtypes[['series_id', 'dim_id', 'floor', 'cap']].astype(str) == ['int32', 'int32', 'float32', 'float32']).all()
        f2 = frame.sort_values(['series_id', 'dim_id']).reset_index(drop=True)
        assert list(f2['model']) == list(back['model']) and np.array_equal(f2['cap'].astype('float32'), back['cap'])
        side = pm.previous_run_cost(cfg['io']['models'], sidecar_only=True)
        assert side is not None and len(side) == 13
        blobs = pm.previous_run_cost(back)
        assert np.array_equal(side.sort_values(['series_id', 'dim_id'])['cost'].to_numpy(), blobs['cost'].to_numpy())
        out[tag] = back
    assert out['whole'].equals(out['chunked']) and out['whole'].equals(out['auto'])
    # the scorer: one pass over the row groups of every part; the CSV parts together = the single-file sink
    monkeypatch.setattr(pm, 'MODEL_ROW_GROUP', 4)
    monkeypatch.setattr(ps, 'SINK_PART_ROWS', 25)
    cfg = dict(base, io={'input': str(tmp_path / 'in'), 'models': str(tmp_path / 'm_rg'), 'chunks': 2})
    pm.ProphetModeler.model(None, cfg, return_frame=False)
    import pyarrow.parquet as pq
    assert [pq.ParquetFile(os.path.join(cfg['io']['models'], f)).num_row_groups
            for f in sorted(os.listdir(cfg['io']['models'])) if f.endswith('.parquet')] == [2, 2]
    scfg = {'io': {'models': cfg['io']['models'], 'forecasts': str(tmp_path / 'fc')},
            'forecast': {'periods': 10, 'frequency': 'D'}}
    assert ps.ProphetScorer.score(None, scfg) is None
    parts = sorted(f for f in os.listdir(scfg['io']['forecasts']) if f.endswith('.csv'))
    assert len(parts) >= 4
    got = pd.concat([pd.read_csv(os.path.join(scfg['io']['forecasts'], f)) for f in parts], ignore_index=True)
    sc = ps.ProphetScorer(dict(scfg, io=dict(scfg['io'], forecasts=str(tmp_path / 'fc_one'))))
    created = got['created_timestamp'].iloc[0]
    sc.write_converted(ps.forecast_panel(scfg)(pd.read_parquet(cfg['io']['models'])), created_timestamp=created)
    want = pd.read_csv(os.path.join(str(tmp_path / 'fc_one'), 'part-00000.csv'))
    key = ['series_id', 'dim_id', 'forecast_timestamp']
    assert list(got.columns) == ps.CONVERTED_COLUMNS and len(got) == 130
    assert got.sort_values(key).reset_index(drop=True).equals(want.sort_values(key).reset_index(drop=True))
    # a partition directory below another one: the same series could sit in two ranges -> the tree is read whole
    nested = tmp_path / 'in' / 'series_id=1' / 'series_id=99'
    nested.mkdir()
    (nested / 'x.csv').write_text('1,2020-01-01 00:00:00,5\n1,2020-01-02 00:00:00,6\n')
    cfg = dict(base, io={'input': str(tmp_path / 'in'), 'models': str(tmp_path / 'm_nested'), 'chunks': 3})
    frame = pm.ProphetModeler.model(None, cfg)
    assert len(frame) == 14 and len([f for f in os.listdir(cfg['io']['models']) if f.endswith('.parquet')]) == 1
    # a plain file among the root's children: not a Hive-only layout, read whole as well
    (tmp_path / 'in' / 'loose.csv').write_text('8,1,2020-01-01 00:00:00,5\n8,1,2020-01-02 00:00:00,6\n')
    frame = pm.ProphetModeler.model(None, cfg)
    assert len(frame) == 15 and len([f for f in os.listdir(cfg['io']['models']) if f.endswith('.parquet')]) == 1
    capsys.readouterr()


def test_model_batch_frame_and_table_agree_with_dropped_and_retried_series(built, monkeypatch):
    """fbprophet's rule inside fit_packed leaves pieces from several fit calls (L-BFGS, the Newton retry of its failures,
    Newton for short series); ModelBatch turns them into the reference's frame and into an Arrow table: same rows."""
    def result(spec, N, grid_n, status, mark):
        g = np.zeros(grid_n, dtype=_lib.GRID_DTYPE)
        g['S'] = 3
        th = np.full((N, spec.theta_stride), float(mark))
        return fc.FitResult(spec, th, np.ones(N), np.zeros(N), np.asarray(status, np.int32), np.full(N, mark, np.int32),
                            np.ones(N, np.int32), g)

    def fake_aligned(spec, ds_ns, y, floor=None, cap=None, extra=None, ctx=None, devices=None):
        algo = spec.lbfgs.get('algorithm')
        st = np.full(len(y), 31 if algo == _lib.ALGO_LBFGS else 60)
        if algo == _lib.ALGO_LBFGS:
            st[0] = -1                                       # first series: line search failed -> retried by Newton
        return result(spec, len(y), 1, st, 1 if algo == _lib.ALGO_LBFGS else 2)

    def fake_ragged(spec, offsets, ds_ns, y, floor=None, cap=None, extra=None, ctx=None, devices=None):
        algo = spec.lbfgs.get('algorithm')
        n = len(offsets) - 1
        st = np.full(n, 60 if algo else 31)
        if algo == _lib.ALGO_NEWTON and tuple(np.diff(offsets)) == (99,):
            st[0] = _lib.ST_NEWTON_FAIL                      # the 99-row series fails under Newton: dropped
        return result(spec, n, n, st, 3 if algo else 4)

    monkeypatch.setattr(fc, 'fit_aligned', fake_aligned)
    monkeypatch.setattr(fc, 'fit_ragged', fake_ragged)
    day = np.datetime64('2020-01-01', 'ns') + np.arange(150).astype('timedelta64[D]')
    rows = []
    for sid, T in enumerate([150, 150, 150, 60, 60, 99]):
        rows += [(sid, 1, d, 5 + (i * (sid + 1)) % 7) for i, d in enumerate(day[:T])]
    df = pd.DataFrame(rows, columns=['series_id', 'dim_id', 'ds', 'y'])
    cfg = {'model': {'floor': 0, 'cap_multiplier': 1.1, 'prophet': {'growth': 'linear', 'seasonality_mode': 'additive',
                                                                     'min_aligned_group': 2}}}
    batch = pm._model_packed(cfg, pk.pack_long_frame(df), len(df), 0.0)
    assert isinstance(batch, pm.ModelBatch) and len(batch) == 5 and list(batch.keep) == [0, 1, 2, 3, 4]
    frame, table = batch.to_frame(), batch.to_table()
    assert list(frame['series_id']) == [0, 1, 2, 3, 4] and table.column('model').to_pylist() == list(frame['model'])
    assert table.schema.names == pm.MODEL_OUTPUT_COLUMNS and str(table.schema.field('cap').type) == 'float'
    marks = [pk.load_model(b)['theta'][0] for b in frame['model']]
    assert marks == [3.0, 1.0, 1.0, 2.0, 2.0]               # retried by Newton (ragged call), L-BFGS, L-BFGS, Newton (short), Newton (short)
    assert list(np.asarray(frame.attrs['tsf_cost'])) == [3, 1, 1, 2, 2]


def test_packer_takes_the_reference_schemas_own_types_in_place(built):
    """tsf_pack_rows_typed: int32 series_id / dim_id / y and datetime64 ds (MODEL_INPUT_SCHEMA, prophet_modeler.py:12-17)
    go to the packer as they are; a frame already in packed order is used in place (no int64 / float64 copies of its
    columns), any other order gives the same panel as the int64 / float64 route, and infinity / NaT raise as fbprophet does."""
    rng = np.random.default_rng(4)
    n_s, T = 37, 50
    sid = np.repeat(np.arange(n_s, dtype=np.int32) * 3 + 1, T)
    did = np.repeat((np.arange(n_s, dtype=np.int32) % 2) + 1, T)
    ds = np.tile(np.datetime64('2020-01-01', 'ns') + np.arange(T).astype('timedelta64[D]'), n_s)
    y = rng.integers(1, 10 ** 6, n_s * T).astype(np.int32)
    df = pd.DataFrame({'series_id': sid, 'dim_id': did, 'ds': ds, 'y': y})
    p = pk.pack_long_frame(df)
    assert p.N == n_s and p.aligned and p.integral and p.y.dtype == np.int32 and p.y2d.dtype == np.int32
    assert np.shares_memory(p.y, df['y'].to_numpy()) and np.shares_memory(p.ds_ns, df['ds'].to_numpy().view(np.int64))
    assert str(p.keys['series_id'].dtype) == 'int32' and list(p.keys['series_id'][:3]) == [1, 4, 7]
    rs, rd, ro, rds, ry = helpers.pack_reference(sid.astype(np.int64), did.astype(np.int64), ds.view(np.int64), y.astype(np.float64))
    assert np.array_equal(p.offsets, ro) and np.array_equal(p.ds_ns, rds) and np.array_equal(p.y, ry)
    assert np.array_equal(p.stats[2], y.reshape(n_s, T).max(axis=1))
    for ydt in (np.int32, np.float32, np.float64):
        o = rng.permutation(n_s * T)
        q = pk.pack_long_frame(df.iloc[o].assign(y=lambda f: f['y'].astype(ydt)).reset_index(drop=True))
        assert q.y.dtype == np.float64 and np.array_equal(q.offsets, ro) and np.array_equal(q.ds_ns, rds) and np.array_equal(q.y, ry)
        assert q.aligned and q.integral and np.array_equal(q.keys['series_id'].to_numpy(), rs)
    bad = df.assign(y=df['y'].astype(np.float64))
    bad.loc[17, 'y'] = np.inf
    with pytest.raises(ValueError, match='infinity'):
        pk.pack_long_frame(bad)
    bad = df.copy()
    bad.loc[5, 'ds'] = pd.NaT
    with pytest.raises(ValueError, match='NaN in column ds'):
        pk.pack_long_frame(bad)
    # float32 y with a null: dropped as fbprophet drops it, the rest unchanged
    f32 = df.assign(y=df['y'].astype(np.float32))
    f32.loc[3, 'y'] = np.nan
    q = pk.pack_long_frame(f32)
    assert q.lengths[0] == T - 1 and not q.aligned and q.y.dtype == np.float64
