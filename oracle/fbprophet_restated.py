"""CPU restatement of fbprophet==0.5 (model side) -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (mageky/time-series-spark) delegates every number on its
hot path to two un-vendored third-party packages, ``fbprophet==0.5`` and
``pystan==2.19.1.1`` (/root/reference/environment.yml:12-13).  Neither can be installed
here (no network) and the reference's own tests assert no numeric value
(/root/reference/tests/unit/prophet_modeler_test.py:65-75,
/root/reference/tests/unit/prophet_scorer_test.py:97-114).  This file therefore restates
the *published* fbprophet 0.5 algorithm (forecaster.py + stan/unix/prophet.stan) from
knowledge of that package; every function says which upstream routine it follows.  It is
anchored on the reference's call sites:

    Prophet(growth='logistic', seasonality_mode='multiplicative')   prophet_modeler.py:65
    model.fit(pdf)                                                   prophet_modeler.py:66
    model.make_future_dataframe(periods, freq, include_history=False) prophet_scorer.py:64-66
    model.predict(future_df)                                         prophet_scorer.py:70

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product (time_series_spark_amd/) never does.

What is here
  * ``ProphetOracle``  -- pandas-in / pandas-out class with fbprophet's method names
    (fit, make_future_dataframe, predict, add_seasonality, add_regressor).
  * ``stan_log_prob`` / ``stan_neg_log_prob_grad`` -- the prophet.stan log-posterior in its
    literal dense-A form (numpy) and its analytic gradient.  The fast C restatement in
    ``stan_lbfgs.c`` is checked against this in tests/test_oracle.py.
The optimiser (Stan's L-BFGS) lives in ``stan_lbfgs.c``; ``ProphetOracle.fit`` calls it via
``oracle_lib``.
"""
from collections import OrderedDict, defaultdict
from datetime import timedelta

import numpy as np
import pandas as pd


# --------------------------------------------------------------------------------------
# Feature construction (fbprophet/forecaster.py)
# --------------------------------------------------------------------------------------

def fourier_series(dates, period, series_order):
    """fbprophet.Prophet.fourier_series: days since epoch, columns
    [sin(2*pi*1*t/p), cos(2*pi*1*t/p), sin(2*pi*2*t/p), ...]."""
    ns = np.asarray(pd.DatetimeIndex(dates).asi8, dtype=np.int64)
    # (dates - datetime(1970,1,1)).dt.total_seconds() / (3600*24.)
    t = (ns.astype(np.float64) / 1e9) / (3600 * 24.)
    return np.column_stack([
        fun((2.0 * (i + 1) * np.pi * t / period))
        for i in range(series_order)
        for fun in (np.sin, np.cos)
    ])


def piecewise_linear(t, deltas, k, m, changepoint_ts):
    """fbprophet.Prophet.piecewise_linear."""
    gammas = -changepoint_ts * deltas
    k_t = k * np.ones_like(t)
    m_t = m * np.ones_like(t)
    for s, t_s in enumerate(changepoint_ts):
        indx = t >= t_s
        k_t[indx] += deltas[s]
        m_t[indx] += gammas[s]
    return k_t * t + m_t


def piecewise_logistic(t, cap, deltas, k, m, changepoint_ts):
    """fbprophet.Prophet.piecewise_logistic."""
    k_cum = np.concatenate((np.atleast_1d(k), np.cumsum(deltas) + k))
    gammas = np.zeros(len(changepoint_ts))
    for i, t_s in enumerate(changepoint_ts):
        gammas[i] = ((t_s - m - np.sum(gammas)) * (1 - k_cum[i] / k_cum[i + 1]))
    k_t = k * np.ones_like(t)
    m_t = m * np.ones_like(t)
    for s, t_s in enumerate(changepoint_ts):
        indx = t >= t_s
        k_t[indx] += deltas[s]
        m_t[indx] += gammas[s]
    return cap / (1 + np.exp(-k_t * (t - m_t)))


# --------------------------------------------------------------------------------------
# prophet.stan, literal (dense A) form.  Parameter vector layout used everywhere in this
# repo (oracle, C-ABI, HIP kernels):
#       theta = [k, m, log(sigma_obs), delta[0..S-1], beta[0..K-1]]
# --------------------------------------------------------------------------------------

def unpack_theta(theta, S, K):
    k = theta[0]
    m = theta[1]
    log_sigma = theta[2]
    delta = theta[3:3 + S]
    beta = theta[3 + S:3 + S + K]
    return k, m, log_sigma, delta, beta


def stan_trend(dat, k, m, delta):
    """prophet.stan functions linear_trend / logistic_trend (dense A)."""
    A = dat['A']
    t = dat['t']
    tc = dat['t_change']
    if dat['trend_indicator'] == 0:
        return (k + A @ delta) * t + (m + A @ (-tc * delta))
    S = dat['S']
    k_s = np.concatenate(([k], k + np.cumsum(delta)))
    gamma = np.zeros(S)
    m_pr = m
    for i in range(S):
        gamma[i] = (tc[i] - m_pr) * (1 - k_s[i] / k_s[i + 1])
        m_pr = m_pr + gamma[i]
    z = (k + A @ delta) * (t - (m + A @ gamma))
    return dat['cap'] / (1.0 + np.exp(-z))


def stan_log_prob(dat, theta):
    """log-posterior of prophet.stan's model block, `~` statements with constants dropped,
    no Jacobian term (Stan's optimizing() uses jacobian=false)."""
    S, K = dat['S'], dat['K']
    k, m, log_sigma, delta, beta = unpack_theta(theta, S, K)
    sigma = np.exp(log_sigma)
    trend = stan_trend(dat, k, m, delta)
    X = dat['X']
    mu = trend * (1 + X @ (beta * dat['s_m'])) + X @ (beta * dat['s_a'])
    r = dat['y'] - mu
    lp = 0.0
    lp += -0.5 * k * k / 25.0                       # k ~ normal(0, 5)
    lp += -0.5 * m * m / 25.0                       # m ~ normal(0, 5)
    lp += -np.sum(np.abs(delta)) / dat['tau']       # delta ~ double_exponential(0, tau)
    lp += -0.5 * sigma * sigma / 0.25               # sigma_obs ~ normal(0, 0.5)
    lp += -0.5 * np.sum((beta / dat['sigmas']) ** 2)  # beta ~ normal(0, sigmas)
    lp += -dat['T'] * log_sigma - 0.5 * np.sum(r * r) / (sigma * sigma)  # y ~ normal(mu, sigma)
    return lp


def stan_neg_log_prob_grad(dat, theta):
    """f = -log_prob and its analytic gradient w.r.t. the unconstrained vector theta
    (what Stan's ModelAdaptor hands to the BFGS minimiser).  d|x|/dx := sign(x), 0 at 0
    (Stan's fabs autodiff)."""
    S, K, T = dat['S'], dat['K'], dat['T']
    k, m, log_sigma, delta, beta = unpack_theta(theta, S, K)
    sigma = np.exp(log_sigma)
    A, t, tc, X = dat['A'], dat['t'], dat['t_change'], dat['X']
    s_a, s_m = dat['s_a'], dat['s_m']
    logistic = dat['trend_indicator'] == 1
    if not logistic:
        trend = (k + A @ delta) * t + (m + A @ (-tc * delta))
    else:
        k_s = np.concatenate(([k], k + np.cumsum(delta)))
        gamma = np.zeros(S)
        mpr = np.zeros(S + 1)
        mpr[0] = m
        for i in range(S):
            gamma[i] = (tc[i] - mpr[i]) * (1 - k_s[i] / k_s[i + 1])
            mpr[i + 1] = mpr[i] + gamma[i]
        Kt = k + A @ delta
        Mt = m + A @ gamma
        z = Kt * (t - Mt)
        sg = 1.0 / (1.0 + np.exp(-z))
        trend = dat['cap'] * sg
    Xm = X @ (beta * s_m)
    Xa = X @ (beta * s_a)
    mu = trend * (1 + Xm) + Xa
    r = dat['y'] - mu
    sse = np.sum(r * r)
    inv_s2 = 1.0 / (sigma * sigma)
    f = (0.5 * k * k / 25.0 + 0.5 * m * m / 25.0 + np.sum(np.abs(delta)) / dat['tau']
         + 2.0 * sigma * sigma + 0.5 * np.sum((beta / dat['sigmas']) ** 2)
         + T * log_sigma + 0.5 * sse * inv_s2)
    g = np.zeros_like(theta)
    dmu = -r * inv_s2                      # df/dmu_t
    # beta
    g_beta = (X.T @ dmu) * s_a + (X.T @ (dmu * trend)) * s_m + beta / dat['sigmas'] ** 2
    dtrend = dmu * (1 + Xm)
    if not logistic:
        gk = np.sum(dtrend * t)
        gm = np.sum(dtrend)
        g_delta = A.T @ (dtrend * t) - tc * (A.T @ dtrend)
    else:
        dz = dtrend * dat['cap'] * sg * (1 - sg)
        dKt = dz * (t - Mt)
        dMt = -dz * Kt
        # K_c, M_c per segment c = number of changepoints <= t
        cidx = A.sum(axis=1).astype(int)
        dKc = np.bincount(cidx, weights=dKt, minlength=S + 1)
        dMc = np.bincount(cidx, weights=dMt, minlength=S + 1)
        # reverse through mpr recurrence: mpr[c+1] = mpr[c] + (tc[c]-mpr[c]) * rho_c
        ks_bar = dKc.copy()
        a_bar = np.zeros(S + 1)
        a_bar[S] = dMc[S]
        for c in range(S - 1, -1, -1):
            rho = 1 - k_s[c] / k_s[c + 1]
            rho_bar = a_bar[c + 1] * (tc[c] - mpr[c])
            a_bar[c] = dMc[c] + a_bar[c + 1] * (1 - rho)
            ks_bar[c] += rho_bar * (-1.0 / k_s[c + 1])
            ks_bar[c + 1] += rho_bar * (k_s[c] / (k_s[c + 1] ** 2))
        gk = np.sum(ks_bar)
        gm = a_bar[0]
        # delta_j feeds k_s[c] for all c > j
        suffix = np.cumsum(ks_bar[::-1])[::-1]
        g_delta = suffix[1:].copy()
    g[0] = gk + k / 25.0
    g[1] = gm + m / 25.0
    g[2] = T - sse * inv_s2 + 4.0 * sigma * sigma
    g[3:3 + S] = g_delta + np.sign(delta) / dat['tau']
    g[3 + S:] = g_beta
    return f, g


# --------------------------------------------------------------------------------------
# The Prophet class (fit / predict plumbing)
# --------------------------------------------------------------------------------------

class ProphetOracle(object):
    """Restatement of fbprophet.Prophet (v0.5) restricted to MAP fitting and point
    prediction.  Not offered: MCMC, uncertainty intervals (the reference discards them,
    /root/reference/src/jobs/prophet_scorer.py:86), built-in country holidays (the
    ``holidays`` pip package is absent; pass an explicit holidays frame)."""

    def __init__(self, growth='linear', changepoints=None, n_changepoints=25,
                 changepoint_range=0.8, yearly_seasonality='auto',
                 weekly_seasonality='auto', daily_seasonality='auto', holidays=None,
                 seasonality_mode='additive', seasonality_prior_scale=10.0,
                 holidays_prior_scale=10.0, changepoint_prior_scale=0.05,
                 lbfgs_options=None):
        self.growth = growth
        self.changepoints = pd.to_datetime(changepoints) if changepoints is not None else None
        if self.changepoints is not None:
            self.n_changepoints = len(self.changepoints)
            self.specified_changepoints = True
        else:
            self.n_changepoints = n_changepoints
            self.specified_changepoints = False
        self.changepoint_range = changepoint_range
        self.yearly_seasonality = yearly_seasonality
        self.weekly_seasonality = weekly_seasonality
        self.daily_seasonality = daily_seasonality
        self.holidays = holidays
        self.seasonality_mode = seasonality_mode
        self.seasonality_prior_scale = float(seasonality_prior_scale)
        self.changepoint_prior_scale = float(changepoint_prior_scale)
        self.holidays_prior_scale = float(holidays_prior_scale)
        self.lbfgs_options = dict(lbfgs_options or {})
        self.start = None
        self.y_scale = None
        self.logistic_floor = False
        self.t_scale = None
        self.changepoints_t = None
        self.seasonalities = OrderedDict({})
        self.extra_regressors = OrderedDict({})
        self.params = {}
        self.history = None
        self.train_component_cols = None
        self.train_holiday_names = None
        self.fit_info = {}
        if self.growth not in ('linear', 'logistic'):
            raise ValueError("Parameter 'growth' should be 'linear' or 'logistic'.")
        if (self.changepoint_range < 0) or (self.changepoint_range > 1):
            raise ValueError("Parameter 'changepoint_range' must be in [0, 1]")
        if self.seasonality_mode not in ['additive', 'multiplicative']:
            raise ValueError("seasonality_mode must be 'additive' or 'multiplicative'")

    # ---- Prophet.setup_dataframe -----------------------------------------------------
    def setup_dataframe(self, df, initialize_scales=False):
        if 'y' in df:
            df['y'] = pd.to_numeric(df['y'])
            if np.isinf(df['y'].values).any():
                raise ValueError('Found infinity in column y.')
        df['ds'] = pd.to_datetime(df['ds'])
        if df['ds'].isnull().any():
            raise ValueError('Found NaN in column ds.')
        for name in self.extra_regressors:
            if name not in df:
                raise ValueError('Regressor "{}" missing from dataframe'.format(name))
            df[name] = pd.to_numeric(df[name])
            if df[name].isnull().any():
                raise ValueError('Found NaN in column ' + name)
        df = df.sort_values('ds', kind='mergesort')
        df = df.reset_index(drop=True)
        self.initialize_scales(initialize_scales, df)
        if self.logistic_floor:
            if 'floor' not in df:
                raise ValueError("Expected column 'floor'.")
        else:
            df['floor'] = 0
        if self.growth == 'logistic':
            if 'cap' not in df:
                raise ValueError("Capacities must be supplied for logistic growth in column 'cap'")
            if (df['cap'] <= df['floor']).any():
                raise ValueError('cap must be greater than floor (which defaults to 0).')
            df['cap_scaled'] = (df['cap'] - df['floor']) / self.y_scale
        df['t'] = (df['ds'] - self.start) / self.t_scale
        if 'y' in df:
            df['y_scaled'] = (df['y'] - df['floor']) / self.y_scale
        for name, props in self.extra_regressors.items():
            df[name] = ((df[name] - props['mu']) / props['std'])
        return df

    # ---- Prophet.initialize_scales ---------------------------------------------------
    def initialize_scales(self, initialize_scales, df):
        if not initialize_scales:
            return
        if self.growth == 'logistic' and 'floor' in df:
            self.logistic_floor = True
            floor = df['floor']
        else:
            floor = 0.
        self.y_scale = (df['y'] - floor).abs().max()
        if self.y_scale == 0:
            self.y_scale = 1
        self.start = df['ds'].min()
        self.t_scale = df['ds'].max() - self.start
        for name, props in self.extra_regressors.items():
            standardize = props['standardize']
            n_vals = len(df[name].unique())
            if n_vals < 2:
                standardize = False
            if standardize == 'auto':
                if set(df[name].unique()) == set([1, 0]):
                    standardize = False
                else:
                    standardize = True
            if standardize:
                mu = df[name].mean()
                std = df[name].std()
                self.extra_regressors[name]['mu'] = mu
                self.extra_regressors[name]['std'] = std

    # ---- Prophet.set_changepoints ----------------------------------------------------
    def set_changepoints(self):
        if self.changepoints is not None:
            if len(self.changepoints) > 0:
                too_low = min(self.changepoints) < self.history['ds'].min()
                too_high = max(self.changepoints) > self.history['ds'].max()
                if too_low or too_high:
                    raise ValueError('Changepoints must fall within training data.')
        else:
            hist_size = int(np.floor(self.history.shape[0] * self.changepoint_range))
            if self.n_changepoints + 1 > hist_size:
                self.n_changepoints = hist_size - 1
            if self.n_changepoints > 0:
                cp_indexes = (
                    np.linspace(0, hist_size - 1, self.n_changepoints + 1)
                    .round().astype(int)
                )
                self.changepoints = (self.history.iloc[cp_indexes]['ds'].tail(-1))
            else:
                self.changepoints = pd.Series(pd.to_datetime([]), name='ds')
        if len(self.changepoints) > 0:
            self.changepoints_t = np.sort(np.array(
                (pd.DatetimeIndex(self.changepoints) - self.start) / self.t_scale,
                dtype=np.float64))
        else:
            self.changepoints_t = np.array([0.])  # dummy changepoint

    # ---- seasonality / holiday / regressor features ----------------------------------
    def add_seasonality(self, name, period, fourier_order, prior_scale=None, mode=None):
        if self.history is not None:
            raise Exception("Seasonality must be added prior to model fitting.")
        ps = self.seasonality_prior_scale if prior_scale is None else float(prior_scale)
        if ps <= 0:
            raise ValueError('Prior scale must be > 0')
        if fourier_order <= 0:
            raise ValueError('Fourier Order must be > 0')
        mode = self.seasonality_mode if mode is None else mode
        if mode not in ['additive', 'multiplicative']:
            raise ValueError("mode must be 'additive' or 'multiplicative'")
        self.seasonalities[name] = {'period': period, 'fourier_order': fourier_order,
                                    'prior_scale': ps, 'mode': mode}
        return self

    def add_regressor(self, name, prior_scale=None, standardize='auto', mode=None):
        if self.history is not None:
            raise Exception("Regressors must be added prior to model fitting.")
        if prior_scale is None:
            prior_scale = float(self.holidays_prior_scale)
        if mode is None:
            mode = self.seasonality_mode
        if prior_scale <= 0:
            raise ValueError('Prior scale must be > 0')
        self.extra_regressors[name] = {'prior_scale': prior_scale, 'standardize': standardize,
                                       'mu': 0., 'std': 1., 'mode': mode}
        return self

    def make_holiday_features(self, dates, holidays):
        """Prophet.make_holiday_features: one indicator column per (holiday, window
        offset), columns sorted by name '<holiday>_delim_<+|-><offset>'."""
        expanded_holidays = defaultdict(lambda: np.zeros(dates.shape[0]))
        prior_scales = {}
        row_dates = pd.DatetimeIndex(dates).normalize()
        for _ix, row in holidays.iterrows():
            dt = pd.Timestamp(row.ds).normalize()
            try:
                lw = int(row.get('lower_window', 0))
                uw = int(row.get('upper_window', 0))
            except ValueError:
                lw = 0
                uw = 0
            ps = float(row.get('prior_scale', self.holidays_prior_scale))
            if np.isnan(ps):
                ps = float(self.holidays_prior_scale)
            if row.holiday in prior_scales and prior_scales[row.holiday] != ps:
                raise ValueError('Holiday {} does not have consistent prior scale '
                                 'specification.'.format(row.holiday))
            if ps <= 0:
                raise ValueError('Prior scale must be > 0')
            prior_scales[row.holiday] = ps
            for offset in range(lw, uw + 1):
                occurrence = dt + timedelta(days=offset)
                key = '{}_delim_{}{}'.format(row.holiday, '+' if offset >= 0 else '-',
                                             abs(offset))
                col = expanded_holidays[key]
                col[np.asarray(row_dates == occurrence)] = 1.
        holiday_features = pd.DataFrame(expanded_holidays)
        holiday_features = holiday_features[sorted(holiday_features.columns.tolist())]
        prior_scale_list = [prior_scales[h.split('_delim_')[0]]
                            for h in holiday_features.columns]
        holiday_names = list(prior_scales.keys())
        if self.train_holiday_names is None:
            self.train_holiday_names = pd.Series(holiday_names)
        return holiday_features, prior_scale_list, holiday_names

    def make_all_seasonality_features(self, df):
        seasonal_features = []
        prior_scales = []
        modes = {'additive': [], 'multiplicative': []}
        for name, props in self.seasonalities.items():
            feats = fourier_series(df['ds'], props['period'], props['fourier_order'])
            cols = ['{}_delim_{}'.format(name, i + 1) for i in range(feats.shape[1])]
            seasonal_features.append(pd.DataFrame(feats, columns=cols))
            prior_scales.extend([props['prior_scale']] * feats.shape[1])
            modes[props['mode']].append(name)
        if self.holidays is not None and len(self.holidays) > 0:
            feats, holiday_priors, holiday_names = self.make_holiday_features(
                df['ds'], self.holidays)
            seasonal_features.append(feats)
            prior_scales.extend(holiday_priors)
            modes[self.seasonality_mode].extend(holiday_names)
        for name, props in self.extra_regressors.items():
            seasonal_features.append(pd.DataFrame(df[name]).reset_index(drop=True))
            prior_scales.append(props['prior_scale'])
            modes[props['mode']].append(name)
        if len(seasonal_features) == 0:
            seasonal_features.append(pd.DataFrame({'zeros': np.zeros(df.shape[0])}))
            prior_scales.append(1.)
        seasonal_features = pd.concat(seasonal_features, axis=1)
        # regressor_column_matrix reduced to what the Stan data needs: per-column
        # additive / multiplicative 0-1 masks.
        s_a = np.zeros(seasonal_features.shape[1])
        s_m = np.zeros(seasonal_features.shape[1])
        for j, col in enumerate(seasonal_features.columns):
            comp = col.split('_delim_')[0]
            if comp in modes['additive']:
                s_a[j] = 1.
            if comp in modes['multiplicative']:
                s_m[j] = 1.
        return seasonal_features, prior_scales, s_a, s_m

    # ---- Prophet.set_auto_seasonalities ----------------------------------------------
    def parse_seasonality_args(self, name, arg, auto_disable, default_order):
        if isinstance(arg, str) and arg == 'auto':
            fourier_order = 0
            if name in self.seasonalities:
                pass
            elif auto_disable:
                pass
            else:
                fourier_order = default_order
        elif arg is True:
            fourier_order = default_order
        elif arg is False:
            fourier_order = 0
        else:
            fourier_order = int(arg)
        return fourier_order

    def set_auto_seasonalities(self):
        first = self.history['ds'].min()
        last = self.history['ds'].max()
        dt = self.history['ds'].diff()
        min_dt = dt.iloc[dt.values.nonzero()[0]].min()
        yearly_disable = last - first < pd.Timedelta(days=730)
        fo = self.parse_seasonality_args('yearly', self.yearly_seasonality, yearly_disable, 10)
        if fo > 0:
            self.seasonalities['yearly'] = {'period': 365.25, 'fourier_order': fo,
                                            'prior_scale': self.seasonality_prior_scale,
                                            'mode': self.seasonality_mode}
        weekly_disable = ((last - first < pd.Timedelta(weeks=2)) or
                          (min_dt >= pd.Timedelta(weeks=1)))
        fo = self.parse_seasonality_args('weekly', self.weekly_seasonality, weekly_disable, 3)
        if fo > 0:
            self.seasonalities['weekly'] = {'period': 7, 'fourier_order': fo,
                                            'prior_scale': self.seasonality_prior_scale,
                                            'mode': self.seasonality_mode}
        daily_disable = ((last - first < pd.Timedelta(days=2)) or
                         (min_dt >= pd.Timedelta(days=1)))
        fo = self.parse_seasonality_args('daily', self.daily_seasonality, daily_disable, 4)
        if fo > 0:
            self.seasonalities['daily'] = {'period': 1, 'fourier_order': fo,
                                           'prior_scale': self.seasonality_prior_scale,
                                           'mode': self.seasonality_mode}

    # ---- growth inits ----------------------------------------------------------------
    @staticmethod
    def linear_growth_init(df):
        i0, i1 = df['ds'].idxmin(), df['ds'].idxmax()
        T = df['t'].iloc[i1] - df['t'].iloc[i0]
        k = (df['y_scaled'].iloc[i1] - df['y_scaled'].iloc[i0]) / T
        m = df['y_scaled'].iloc[i0] - k * df['t'].iloc[i0]
        return (k, m)

    @staticmethod
    def logistic_growth_init(df):
        i0, i1 = df['ds'].idxmin(), df['ds'].idxmax()
        T = df['t'].iloc[i1] - df['t'].iloc[i0]
        C0 = df['cap_scaled'].iloc[i0]
        C1 = df['cap_scaled'].iloc[i1]
        y0 = max(0.01 * C0, min(0.99 * C0, df['y_scaled'].iloc[i0]))
        y1 = max(0.01 * C1, min(0.99 * C1, df['y_scaled'].iloc[i1]))
        r0 = C0 / y0
        r1 = C1 / y1
        if abs(r0 - r1) <= 0.01:
            r0 = 1.05 * r0
        L0 = np.log(r0 - 1)
        L1 = np.log(r1 - 1)
        m = L0 * T / (L0 - L1)
        k = (L0 - L1) / T
        return (k, m)

    # ---- Prophet.fit -----------------------------------------------------------------
    def stan_data(self, df):
        """Build the dict fbprophet hands to StanModel.optimizing (plus dense A)."""
        history = df[df['y'].notnull()].copy()
        if history.shape[0] < 2:
            raise ValueError('Dataframe has less than 2 non-NaN rows.')
        self.history_dates = pd.to_datetime(df['ds']).sort_values()
        history = self.setup_dataframe(history, initialize_scales=True)
        self.history = history
        self.set_auto_seasonalities()
        seasonal_features, prior_scales, s_a, s_m = self.make_all_seasonality_features(history)
        self.train_component_cols = (s_a, s_m)
        self.set_changepoints()
        t = history['t'].values.astype(np.float64)
        tc = self.changepoints_t
        A = (t[:, None] >= tc[None, :]).astype(np.float64)
        dat = {
            'T': history.shape[0],
            'K': seasonal_features.shape[1],
            'S': len(tc),
            'y': history['y_scaled'].values.astype(np.float64),
            't': t,
            't_change': tc,
            'A': A,
            'X': seasonal_features.values.astype(np.float64),
            'sigmas': np.asarray(prior_scales, dtype=np.float64),
            'tau': self.changepoint_prior_scale,
            'trend_indicator': int(self.growth == 'logistic'),
            's_a': s_a,
            's_m': s_m,
        }
        if self.growth == 'linear':
            dat['cap'] = np.zeros(history.shape[0])
            kinit = self.linear_growth_init(history)
        else:
            dat['cap'] = history['cap_scaled'].values.astype(np.float64)
            kinit = self.logistic_growth_init(history)
        theta0 = np.zeros(3 + dat['S'] + dat['K'])
        theta0[0] = kinit[0]
        theta0[1] = kinit[1]
        theta0[2] = 0.0  # sigma_obs = 1
        return dat, theta0

    def fit(self, df, optimizer=None, **kwargs):
        if self.history is not None:
            raise Exception('Prophet object can only be fit once. Instantiate a new object.')
        if ('ds' not in df) or ('y' not in df):
            raise ValueError("Dataframe must have columns 'ds' and 'y' with the dates and "
                             "values respectively.")
        dat, theta0 = self.stan_data(df)
        history = self.history
        S, K = dat['S'], dat['K']
        if (history['y'].min() == history['y'].max()) and self.growth == 'linear':
            # Nothing to fit.
            theta = theta0.copy()
            theta[2] = np.log(1e-9)
            self.fit_info = {'status': 'constant', 'n_iter': 0, 'n_eval': 0}
        else:
            if optimizer is None:
                from oracle import oracle_lib
                optimizer = oracle_lib.stan_lbfgs
            theta, info = optimizer(dat, theta0, **self.lbfgs_options)
            self.fit_info = info
            if info.get('error'):
                # Stan raises RuntimeError on line-search failure at the initial point; the
                # reference turns that into "series dropped" (prophet_modeler.py:81-85).
                raise RuntimeError(info['error'])
        k, m, log_sigma, delta, beta = unpack_theta(theta, S, K)
        self.params = {'k': np.array([k]), 'm': np.array([m]),
                       'sigma_obs': np.array([np.exp(log_sigma)]),
                       'delta': np.array(delta).reshape(1, -1),
                       'beta': np.array(beta).reshape(1, -1)}
        self.theta = theta
        # If no changepoints were requested, replace delta with 0s
        if len(self.changepoints) == 0:
            self.params['k'] = self.params['k'] + self.params['delta'].reshape(-1)
            self.params['delta'] = np.zeros(self.params['delta'].shape).reshape((-1, 1))
        return self

    # ---- Prophet.make_future_dataframe -----------------------------------------------
    def make_future_dataframe(self, periods, freq='D', include_history=True):
        if self.history_dates is None:
            raise Exception('Model must be fit before this can be used.')
        last_date = self.history_dates.max()
        dates = pd.date_range(start=last_date, periods=periods + 1, freq=freq)
        dates = dates[dates > last_date]
        dates = dates[:periods]
        if include_history:
            dates = np.concatenate((np.array(self.history_dates), dates))
        return pd.DataFrame({'ds': dates})

    # ---- Prophet.predict (point forecast only) ---------------------------------------
    def predict_trend(self, df):
        k = np.nanmean(self.params['k'])
        m = np.nanmean(self.params['m'])
        deltas = np.nanmean(self.params['delta'], axis=0)
        t = np.array(df['t'], dtype=np.float64)
        if self.growth == 'linear':
            trend = piecewise_linear(t, deltas, k, m, self.changepoints_t)
        else:
            cap = np.array(df['cap_scaled'], dtype=np.float64)
            trend = piecewise_logistic(t, cap, deltas, k, m, self.changepoints_t)
        return trend * self.y_scale + df['floor'].values

    def predict(self, df=None):
        if df is None:
            df = self.history.copy()
        else:
            if df.shape[0] == 0:
                raise ValueError('Dataframe has no rows.')
            df = self.setup_dataframe(df.copy())
        trend = self.predict_trend(df)
        seasonal_features, _, s_a, s_m = self.make_all_seasonality_features(df)
        X = seasonal_features.values
        beta = self.params['beta'][0]
        additive = (X @ (beta * s_a)) * self.y_scale
        multiplicative = X @ (beta * s_m)
        out = pd.DataFrame({'ds': df['ds'].values, 'trend': trend,
                            'additive_terms': additive,
                            'multiplicative_terms': multiplicative})
        out['yhat'] = out['trend'] * (1 + out['multiplicative_terms']) + out['additive_terms']
        return out

    # ---- Prophet.predict_uncertainty (literal: numpy's global generator, as fbprophet) ---------
    def sample_predictive_trend(self, df):
        k = self.params['k'][0]
        m = self.params['m'][0]
        deltas = self.params['delta'][0]
        t = np.array(df['t'])
        T = t.max()
        if T > 1:
            S = len(self.changepoints_t)
            n_changes = np.random.poisson(S * (T - 1))
        else:
            n_changes = 0
        if n_changes > 0:
            changepoint_ts_new = 1 + np.random.rand(n_changes) * (T - 1)
            changepoint_ts_new.sort()
        else:
            changepoint_ts_new = []
        lambda_ = np.mean(np.abs(deltas)) + 1e-8
        deltas_new = np.random.laplace(0, lambda_, n_changes)
        changepoint_ts = np.concatenate((self.changepoints_t, changepoint_ts_new))
        deltas = np.concatenate((deltas, deltas_new))
        if self.growth == 'linear':
            trend = piecewise_linear(t, deltas, k, m, changepoint_ts)
        else:
            trend = piecewise_logistic(t, np.array(df['cap_scaled']), deltas, k, m, changepoint_ts)
        return trend * self.y_scale + df['floor'].values

    def predict_uncertainty(self, df, uncertainty_samples=1000, interval_width=0.8):
        """yhat_lower / yhat_upper as fbprophet computes them (sample_posterior_predictive with MAP
        parameters, sample_model, np.nanpercentile)."""
        df = self.setup_dataframe(df.copy())
        seasonal_features, _, s_a, s_m = self.make_all_seasonality_features(df)
        X = seasonal_features.values
        beta = self.params['beta'][0]
        Xb_a = (X @ (beta * s_a)) * self.y_scale
        Xb_m = X @ (beta * s_m)
        sigma = self.params['sigma_obs'][0]
        sims = np.empty((df.shape[0], uncertainty_samples))
        for i in range(uncertainty_samples):
            trend = self.sample_predictive_trend(df)
            noise = np.random.normal(0, sigma, df.shape[0]) * self.y_scale
            sims[:, i] = trend * (1 + Xb_m) + Xb_a + noise
        lower_p = 100 * (1.0 - interval_width) / 2
        upper_p = 100 * (1.0 + interval_width) / 2
        return np.nanpercentile(sims, lower_p, axis=1), np.nanpercentile(sims, upper_p, axis=1)

