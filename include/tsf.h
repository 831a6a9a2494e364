/*
 * tsf.h -- C ABI of libtsf_amd.so: batched per-series Prophet-model MAP fit + predict on
 * MI355X (gfx950).  Plain pointers and sizes, no C++/torch types, no exceptions.
 *
 * What each entry point replaces in the reference (mageky/time-series-spark):
 *
 *   tsf_fit_aligned / tsf_fit_ragged (+ _dev)
 *       `Prophet(growth=..., seasonality_mode=...)` + `model.fit(pdf)` executed once per
 *       (series_id, dim_id) group inside the grouped-map pandas_udf
 *       /root/reference/src/jobs/prophet_modeler.py:56-66 (floor :56-57, cap :59-60,
 *       constructor :65, fit :66), i.e. fbprophet 0.5 setup_dataframe / set_changepoints /
 *       make_all_seasonality_features / *_growth_init and pystan 2.19.1.1
 *       StanModel.optimizing(algorithm='LBFGS' | 'Newton', see tsf_spec.algorithm) on
 *       prophet.stan -- for a whole panel of series in one call.
 *   tsf_predict (+ _dev)
 *       `model.predict(future_df)` + the int cast + floor clamp of
 *       /root/reference/src/jobs/prophet_scorer.py:64-84 (future frame :64-68, predict :70,
 *       astype(int) :73, clamp :76-84).  Only `yhat` is produced: the reference keeps
 *       nothing else (:86).
 *   (tests and measurements -- per-evaluation hooks, route switches, kernel timers -- are NOT here: include/tsf_dev.h)
 *
 * Conventions
 *   - Every function returns 0 on success, <0 on API misuse / HIP failure
 *     (tsf_last_error(ctx) gives the text).  Per-series outcomes go to status[] (TSF_ST_*),
 *     mirroring the reference's "RuntimeError -> series dropped" (prophet_modeler.py:81-85)
 *     vs "ValueError propagates" split: the Python layer decides what to raise.
 *   - All buffers are caller-allocated and caller-owned.  `_dev` variants take DEVICE
 *     pointers on the context's GPU plus a hipStream_t (as void*, NULL = default stream) and
 *     are asynchronous; the plain variants take HOST pointers, copy in, run, copy out, sync.
 *   - A tsf_ctx is bound to one GPU and is not re-entrant; use one per GPU / host thread.
 *   - Timestamps are int64 nanoseconds since the Unix epoch (pandas datetime64[ns]), sorted
 *     ascending within a series, rows with NaN y already removed (fbprophet
 *     setup_dataframe does both on the host as well).
 *   - theta layout per series, stride tsf_theta_stride(spec):
 *       [k, m, log(sigma_obs), delta[n_changepoints], beta[K]]
 *     beta in design-column order: seasonalities in spec order, each
 *     [sin1, cos1, sin2, cos2, ...], then the extra (holiday / regressor) columns.
 *     When a series is too short for n_changepoints, S < n_changepoints and the unused
 *     deltas are 0.
 */
#ifndef TSF_H
#define TSF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSF_MAX_SEAS 8
#define TSF_MAX_EXTRA 64
#define TSF_MAX_S 60          /* max changepoints */
#define TSF_MAX_K 64          /* max design columns */
#define TSF_MAX_P 128         /* 3 + S + K */
#define TSF_MAX_T 1048576     /* rows per series (a series is one wavefront's work: 64 chunks of
                                 ceil(T/64) rows); every fit entry point rejects longer input */

enum { TSF_GROWTH_LINEAR = 0, TSF_GROWTH_LOGISTIC = 1 };
enum { TSF_MODE_ADDITIVE = 0, TSF_MODE_MULTIPLICATIVE = 1 };
enum { TSF_Y_F64 = 0, TSF_Y_F32 = 1, TSF_Y_I32 = 2 };
enum { TSF_EVAL_AUTO = 0, TSF_EVAL_RESIDUAL = 1, TSF_EVAL_QUADRATIC = 2 };
/* Which of Stan's optimisers runs the MAP fit.  fbprophet 0.5 chooses
 * `'Newton' if T < 100 else 'LBFGS'` (TSF_ALGO_AUTO: decided per CALL from the longest series of
 * the call -- callers split panels at 100 rows).  Default TSF_ALGO_LBFGS.
 * Newton kernels: one parameter per lane for models of up to 64 parameters (3 + n_changepoints + K) of one
 * column mode -- quadratic-form evaluations, several series per wave, for linear / additive models of up to 28
 * design columns --; two parameters per lane (round 4, one wave and ~140 KB of LDS per series: slow, meant for the
 * handful of series fbprophet retries after a failed L-BFGS fit) up to TSF_MAX_P = 128 and for mixed additive /
 * multiplicative columns.  Every model the library fits has one. */
enum { TSF_ALGO_LBFGS = 0, TSF_ALGO_NEWTON = 1, TSF_ALGO_AUTO = 2 };
enum { TSF_RK_AUTO = 0, TSF_RK_WAVE = 1, TSF_RK_MFMA = 2, TSF_RK_COOP = 3 };
/* What a fit converges to.  TSF_CONVERGE_STAN (default): Stan's optimiser under Stan's termination tests -- what
 * `Prophet.fit` returns (prophet_modeler.py:66): the point where the tests fire, a median 1e-3 .. 1e-2 away from the
 * optimum in forecast, because L-BFGS stalls on the kinks of the Laplace prior on the changepoints.
 * TSF_CONVERGE_MAP: from that point on to the maximum a posteriori estimate of prophet.stan's model itself (an
 * orthant-wise active-set L-BFGS on the same log-posterior, tsf_map_kernels.h): forecasts that are a property of the
 * model, not of a floating-point trajectory.  status then is TSF_ST_MAP_*; n_iter / n_eval count both phases.
 * For linear growth with additive seasonality (<= 64 parameters) the estimate is computed directly, without the Stan-rule
 * fit before it (tsf_map_quad.h: sigma in closed form and an L1-regularised quadratic programme by an active-set method,
 * in turn): n_iter then counts those rounds, n_eval the Cholesky solves + 1, and the call is faster than a Stan-rule fit. */
enum { TSF_CONVERGE_STAN = 0, TSF_CONVERGE_MAP = 1 };
#define TSF_NEWTON_BELOW_T 100

/* per-series status: >= 0 are Stan's optimiser termination codes */
enum {
    TSF_ST_CONTINUE = 0,       /* never returned */
    TSF_ST_ABSX = 10, TSF_ST_ABSF = 20, TSF_ST_RELF = 21, TSF_ST_ABSGRAD = 30,
    TSF_ST_RELGRAD = 31, TSF_ST_MAXIT = 40,
    TSF_ST_CONSTANT = 50,      /* constant y, linear growth: fbprophet skips optimisation */
    TSF_ST_NEWTON_CONVERGED = 60, /* Newton: |lp - last lp| < 1e-8 */
    /* tsf_spec.converge = TSF_CONVERGE_MAP: how the continuation to the maximum a posteriori estimate ended */
    TSF_ST_MAP_KKT = 70,       /* KKT residual (largest one-sided derivative that still descends) <= map_tol */
    TSF_ST_MAP_FTOL = 71,      /* 20 iterations together gained < 1e-13 |f|: the function value has converged */
    TSF_ST_MAP_MAXIT = 72,     /* map_max_iter iterations (the direct solver: 120 rounds or 3 000 solves) */
    TSF_ST_MAP_LS = 73,        /* no lower point along the steepest one-sided descent direction either (rounding level) */
    TSF_ST_NEWTON_FAIL = -4,   /* Newton: log_prob threw inside the finite-difference Hessian */
    TSF_ST_LSFAIL = -1,        /* line search failed (pystan raises RuntimeError) */
    TSF_ST_INIT_NONFINITE = -2,/* log_prob non-finite at the initial point (RuntimeError) */
    TSF_ST_EVAL_LIMIT = -3,    /* > 64*max_iter+1024 evaluations: the line search never settled
                                  (guard; treated like LSFAIL) */
    TSF_ST_TOO_FEW = -10,      /* < 2 rows (fbprophet raises ValueError) */
    TSF_ST_CAP = -11           /* cap <= floor (fbprophet raises ValueError) */
};

/* Model + optimiser settings shared by every series of a call.  Defaults = fbprophet 0.5
 * Prophet.__init__ and stan::services::optimize::lbfgs as pystan 2.19 drives it. */
typedef struct {
    int32_t growth;                         /* TSF_GROWTH_* */
    int32_t n_changepoints;                 /* 25 */
    double changepoint_range;               /* 0.8 */
    double changepoint_prior_scale;         /* tau = 0.05 */
    int32_t n_seas;                         /* Fourier seasonalities */
    int32_t n_extra;                        /* explicit design columns (holidays, regressors) */
    double seas_period[TSF_MAX_SEAS];       /* days: 365.25, 7, 1 */
    double seas_prior_scale[TSF_MAX_SEAS];  /* 10 */
    int32_t seas_order[TSF_MAX_SEAS];       /* 10, 3, 4 */
    int32_t seas_mode[TSF_MAX_SEAS];        /* TSF_MODE_* */
    double extra_prior_scale[TSF_MAX_EXTRA];
    int32_t extra_mode[TSF_MAX_EXTRA];
    int32_t max_iter;                       /* 10000 */
    int32_t history;                        /* 5 */
    double init_alpha;                      /* 1e-3 */
    double tol_obj;                         /* 1e-12 */
    double tol_rel_obj;                     /* 1e4  (x DBL_EPSILON) */
    double tol_grad;                        /* 1e-8 */
    double tol_rel_grad;                    /* 1e7  (x DBL_EPSILON) */
    double tol_param;                       /* 1e-8 */
    /* How the normal likelihood's data term is evaluated (same function, different rounding):
     * TSF_EVAL_RESIDUAL sums residuals over the T rows at every evaluation; TSF_EVAL_QUADRATIC
     * uses SSE(theta) = s0 - 2 c.D + D.(Z^T Z) D around a re-centred reference point -- only
     * possible where the mean is linear in (k, m, delta, beta): linear growth, every column
     * additive (and history == 5).  TSF_EVAL_AUTO picks QUADRATIC where possible; the choice
     * depends on the MODEL only, never on the shape or composition of the panel. */
    int32_t eval_form;                      /* TSF_EVAL_AUTO */
    int32_t recenter_every;                 /* 128: re-centre at least every n accepted iterations */
    double recenter_ratio;                  /* 1.0: ... and when |Z D|^2 > ratio * s0 */
    int32_t algorithm;                      /* TSF_ALGO_LBFGS */
    /* Which kernel runs a RESIDUAL-form L-BFGS fit (same arithmetic, same bits): TSF_RK_WAVE one
     * wavefront per series; TSF_RK_MFMA 16 series per workgroup evaluated together on the matrix
     * cores (aligned panels, one parameter per lane, <= 28 changepoints); TSF_RK_AUTO = WAVE with
     * the cooperative tail: the series still running when the launch has handed out its last series
     * are suspended and finished by one WORKGROUP each (8 waves sharing every evaluation; series of
     * at most 4096 rows) -- and models with more than 64 parameters (two per lane) run on workgroups from
     * their first evaluation, the workgroup kernel being the faster one for them; TSF_RK_COOP = every
     * series on a workgroup from its first evaluation (lowest latency for panels smaller than the GPU). */
    int32_t residual_kernel;                /* TSF_RK_AUTO */
    /* test / tuning hook of the cooperative tail: >= 0 suspends a fit once it has used that many
     * evaluations instead of at the tail of the launch; -1 = the default rule.  Results do not depend
     * on where a fit is suspended. */
    int32_t coop_after;                     /* -1 */
    int32_t converge;                       /* TSF_CONVERGE_STAN */
    int32_t map_max_iter;                   /* 10000: iterations of the continuation (converge = MAP) */
    double map_tol;                         /* 1e-7: its KKT tolerance */
} tsf_spec;

/* What setup derives from one timestamp vector ("grid").  One per call for aligned panels,
 * one per series for ragged panels. */
typedef struct {
    int64_t start_ns;                       /* min ds */
    int64_t t_scale_ns;                     /* max ds - min ds */
    int32_t T;                              /* rows */
    int32_t S;                              /* changepoints actually used */
    int32_t i1;                             /* first row holding max ds */
    int32_t NT;                             /* chunk length ceil(T/64) */
    double t_change[TSF_MAX_S + 4];         /* scaled changepoint times */
} tsf_grid_info;

/* Fit outputs; every pointer caller-allocated (host or device to match the call). */
typedef struct {
    double *theta;          /* [N][tsf_theta_stride(spec)] */
    double *y_scale;        /* [N] */
    double *fval;           /* [N] -log posterior at the returned theta */
    int32_t *status;        /* [N] TSF_ST_* */
    int32_t *n_iter;        /* [N] L-BFGS / Newton iterations */
    int32_t *n_eval;        /* [N] log_prob+gradient evaluations */
    tsf_grid_info *grid;    /* [1] aligned, [N] ragged */
} tsf_fit_out;

typedef struct tsf_ctx tsf_ctx;

int tsf_create(int device_id, tsf_ctx **out);
void tsf_destroy(tsf_ctx *ctx);
const char *tsf_last_error(const tsf_ctx *ctx);
int tsf_device_count(void);

void tsf_spec_default(tsf_spec *spec);
int tsf_spec_size(void);                      /* sizeof(tsf_spec), for binding self-checks */
int tsf_grid_info_size(void);
int tsf_spec_K(const tsf_spec *spec);         /* design columns */
int tsf_theta_stride(const tsf_spec *spec);   /* 3 + n_changepoints + K */

/* ---- fit ------------------------------------------------------------------------------
 * aligned: every series observed on the same T timestamps.  y is [N][T] row-major of
 * y_dtype.  floor / cap: [N] or NULL (NULL = 0; floor is ignored for linear growth exactly
 * as fbprophet ignores the column).  extra: [n_extra][T] or NULL. */
int tsf_fit_aligned(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int32_t T,
                    const int64_t *ds, const void *y, int32_t y_dtype, const double *floor,
                    const double *cap, const double *extra, tsf_fit_out *out);
int tsf_fit_aligned_dev(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int32_t T,
                        const int64_t *ds, const void *y, int32_t y_dtype,
                        const double *floor, const double *cap, const double *extra,
                        tsf_fit_out *out, void *stream);

/* ragged: series n owns rows [offsets[n], offsets[n+1]) of ds / y / extra
 * (extra: [n_extra][total_rows]). */
int tsf_fit_ragged(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, const int64_t *offsets,
                   const int64_t *ds, const void *y, int32_t y_dtype, const double *floor,
                   const double *cap, const double *extra, tsf_fit_out *out);
int tsf_fit_ragged_dev(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, const int64_t *offsets,
                       int64_t total_rows, int32_t max_T, const int64_t *ds, const void *y,
                       int32_t y_dtype, const double *floor, const double *cap,
                       const double *extra, tsf_fit_out *out, void *stream);

/* ---- predict --------------------------------------------------------------------------
 * yhat[n][h] = trend*(1+multiplicative)+additive in original units (float64).  If
 * yhat_int != NULL also the reference's post-step: (int) truncation toward zero, then
 * values below floor[n] replaced by floor[n] (prophet_scorer.py:73-84).
 * n_grids = 1 (aligned fit) or N.  ds_future: [H] if shared_future else [N][H].
 * extra_future: [n_extra][H] (shared) or [N][n_extra][H]; NULL if n_extra == 0.
 * floor/cap: the values the caller puts in the future frame (prophet_scorer.py:67-68). */
int tsf_predict(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int32_t H, const double *theta,
                const double *y_scale, const tsf_grid_info *grid, int32_t n_grids,
                const int64_t *ds_future, int32_t shared_future, const double *floor,
                const double *cap, const double *extra_future, double *yhat,
                int32_t *yhat_int);
int tsf_predict_dev(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int32_t H,
                    const double *theta, const double *y_scale, const tsf_grid_info *grid,
                    int32_t n_grids, const int64_t *ds_future, int32_t shared_future,
                    const double *floor, const double *cap, const double *extra_future,
                    double *yhat, int32_t *yhat_int, void *stream);

/* ---- uncertainty intervals ----------------------------------------------------------------
 * yhat_lower / yhat_upper of fbprophet's Prophet.predict_uncertainty -- computed by
 * model.predict(future_df) at /root/reference/src/jobs/prophet_scorer.py:70 and dropped at :86:
 * n_samples simulated futures per series (new trend changepoints ~ Poisson(S (T - 1)) on [1, T] with
 * Laplace(0, mean|delta| + 1e-8) slope changes, observation noise N(0, sigma_obs)), then the
 * (1 -+ interval_width) / 2 percentiles per future row.  fbprophet uses numpy's unseeded global
 * generator; here the draws come from a counter-based generator keyed by (seed, series_key[n],
 * sample, stream), so results are reproducible and independent of how series are batched
 * (series_key NULL: the index of the series in this call).  Also returns the point forecast.
 * Arguments as tsf_predict; n_samples in [2, 4096] (fbprophet: 1000), interval_width in (0, 1)
 * (fbprophet: 0.8). */
int tsf_predict_intervals(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int32_t H,
                          const double *theta, const double *y_scale, const tsf_grid_info *grid,
                          int32_t n_grids, const int64_t *ds_future, int32_t shared_future,
                          const double *floor, const double *cap, const double *extra_future,
                          const int64_t *series_key, int32_t n_samples, double interval_width,
                          uint64_t seed, double *yhat, double *yhat_lower, double *yhat_upper);
int tsf_predict_intervals_dev(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int32_t H,
                              const double *theta, const double *y_scale, const tsf_grid_info *grid,
                              int32_t n_grids, const int64_t *ds_future, int32_t shared_future,
                              const double *floor, const double *cap, const double *extra_future,
                              const int64_t *series_key, int32_t n_samples, double interval_width,
                              uint64_t seed, double *yhat, double *yhat_lower, double *yhat_upper,
                              void *stream);

/* ---- scheduling hints ---------------------------------------------------------------------
 * A launch ends with its longest fits (cfg2: 1 582 evaluations against a mean of 454), and nothing
 * cheap about a series predicts how long its fit takes -- except an earlier fit of the same
 * series: a job that re-fits its panel regularly (the reference's modeler is such a job) can hand
 * the evaluation counts of the previous run (tsf_fit_out.n_eval) to the next one.
 * tsf_set_cost_hints: cost[i] = expected relative cost of series i of the NEXT fit call on this
 * context (any fit entry point): consumed by it if it has exactly n series, discarded otherwise.  The work queue of that call hands the
 * series out in order of decreasing cost (ties: by index).  Results do not depend on the order
 * (every series is fitted by itself); only the launch time does: the BASELINE cfg2 panel with the
 * counts of its own previous fit takes 7.1-7.6 ms instead of 9.4.
 * cost: HOST pointer, copied by the call; NULL or n == 0 clears.  The hints are used once.
 * Reference interface replaced: none (Spark's scheduler knows nothing about a group's cost). */
int tsf_set_cost_hints(tsf_ctx *ctx, const int32_t *cost, int64_t n);

/* ---- host-side panel packing (no device work, no tsf_ctx) ---------------------------------
 * Regroups a long table (one row per observation) into the contiguous per-series runs
 * tsf_fit_ragged takes.  Replaces the row movement of
 *   df.groupby('series_id','dim_id').apply(...)   /root/reference/src/jobs/prophet_modeler.py:139-141
 * (Spark shuffle + one Arrow->pandas frame per group) and fbprophet's per-group
 * `history = df[df['y'].notnull()]`, sort by ds (Prophet.fit / setup_dataframe).
 * Order contract: series ascending by (series_id, dim_id); inside a series ascending ds with
 * ties in input order; rows whose y is NaN dropped; series left with no row dropped.
 *
 *   tsf_pack_rows   builds the plan.  The four input arrays are only read and must stay alive
 *                   until tsf_pack_fetch returns.  n_threads <= 0: one per core (max 32).
 *                   *identity = 1 when the input already is in packed order with no NaN
 *                   (then ds/y need not be copied at all).
 *   tsf_pack_fetch  fills caller-allocated outputs (any may be NULL): keys [n_series],
 *                   offsets [n_series+1], ds_out / y_out [n_rows], and per series
 *                   span = last ds - first ds, min_dt = smallest positive spacing (-1 if none),
 *                   y_max -- the inputs of fbprophet's set_auto_seasonalities and of
 *                   cap = max(y) * cap_multiplier (prophet_modeler.py:59-60).
 *   tsf_pack_flags  what tsf_pack_fetch's pass over the packed rows saw (valid after it): *aligned = 1 when every
 *                   series is observed on the first series' timestamp vector (one shared calendar: the panel can go
 *                   to tsf_fit_aligned as [n_series][T] without another look at ds); *has_inf = 1 when a y is
 *                   infinite (fbprophet raises "Found infinity in column y."); *integral = 1 when every y is an
 *                   integer that fits int32 -- the quantity column of the reference's schema
 *                   (prophet_modeler.py:16), which may then cross to the device as TSF_Y_I32; *has_nat = 1 when a ds
 *                   is INT64_MIN, pandas' NaT (fbprophet raises "Found NaN in column ds.").
 *   tsf_pack_rows_typed  the same plan for columns in the caller's own types: keys of key_bytes = 4 (the reference's
 *                   int32 series_id / dim_id, prophet_modeler.py:13-14) or 8, y of y_dtype TSF_Y_* (int32 = the
 *                   reference's quantity: no NaN possible).  A table already in packed order is then used in place, in
 *                   those types (tsf_pack_fetch writes ds_out / y_out -- always int64 / float64 -- only when asked).
 * Returns 0, -1 bad arguments, -2 out of memory, -3 other failure. */
typedef struct tsf_pack tsf_pack;
int tsf_pack_rows(int64_t n, const int64_t *series_id, const int64_t *dim_id, const int64_t *ds,
                  const double *y, int32_t n_threads, tsf_pack **out, int64_t *n_rows,
                  int64_t *n_series, int32_t *identity);
int tsf_pack_fetch(tsf_pack *p, int64_t *key_series_id, int64_t *key_dim_id, int64_t *offsets,
                   int64_t *ds_out, double *y_out, int64_t *span, int64_t *min_dt, double *y_max);
int tsf_pack_rows_typed(int64_t n, const void *series_id, const void *dim_id, int32_t key_bytes, const int64_t *ds,
                        const void *y, int32_t y_dtype, int32_t n_threads, tsf_pack **out, int64_t *n_rows,
                        int64_t *n_series, int32_t *identity);
int tsf_pack_flags(const tsf_pack *p, int32_t *aligned, int32_t *has_inf, int32_t *integral, int32_t *has_nat);
void tsf_pack_free(tsf_pack *p);

/* ---- model blobs (host side) ------------------------------------------------------------------
 * The `model` column of the fit UDF's output -- pickle.dumps(model) per series in the reference
 * (/root/reference/src/jobs/prophet_modeler.py:72-75) -- for a whole fitted batch at once: blob n is written at
 * out + n * stride, stride = prefix_len + 64 + 8 * (n_theta + n_tchange): the prefix (magic, version, the constructor
 * arguments: shared by the batch, built by the caller), then the little-endian record
 *   f64 y_scale | i64 start_ns, t_scale_ns, last_ds_ns | i32 T, S, i1, NT, status, n_iter, n_theta, n_tchange |
 *   f64 theta[n_theta] | f64 t_change[n_tchange]
 * theta [N][n_theta], y_scale / status / n_iter [N] and grid [n_grids] (1 or N) are a fit's outputs (tsf_fit_out);
 * last_ds [N] is the last history date of each series, null-y rows included (where make_future_dataframe starts,
 * prophet_scorer.py:64-66).  One contiguous buffer with equal strides is an Arrow binary column as it stands: the model
 * parquet (prophet_modeler.py:118-125) is written from it without a Python object per series.
 * Returns 0, -1 bad arguments, -2 out of memory. */
int tsf_model_blobs(int64_t N, const void *prefix, int32_t prefix_len, int32_t n_theta, const double *theta,
                    const double *y_scale, const tsf_grid_info *grid, int32_t n_grids, const int64_t *last_ds,
                    const int32_t *status, const int32_t *n_iter, int32_t n_tchange, void *out, int32_t n_threads);

/* ---- model-input reader (host side, no device work, no tsf_ctx) ---------------------------
 * Replaces ProphetModeler.read_input_dataframe's
 *   spark.read.csv(path, header=False, schema=MODEL_INPUT_SCHEMA)
 *   /root/reference/src/jobs/prophet_modeler.py:102-116 (schema :12-17)
 * for header-less CSV files, parsed in parallel straight into the columns tsf_pack_rows takes.
 *   layout      one letter per column of the FILES: 's' series_id, 'd' dim_id, 't' start_time,
 *               'q' quantity, 'x' ignored.  Hive-partitioned input (series_id=751/...csv) has
 *               layout "dtq" and the partition value in series_id[file].
 *   start_time  yyyy-MM-dd[( |T)HH:mm[:ss[.fffffffff]]][Z], taken as naive wall time
 *   quantity    integer (the reference's schema) or decimal; an empty field is a null and
 *               comes out as NaN (the packer drops it as fbprophet drops y.isnull() rows)
 *   rows come out in file order, files in the order given.
 *   A trailing '?' in layout ("dtq?") = Spark's default CSV mode PERMISSIVE: a line that does not
 *   match the schema (a field that does not convert, too few fields) is not an error.  Spark 2.4 turns
 *   it into a row of nulls (every column, dim_id included); a null y is dropped by the fit, and a null
 *   key cannot be expressed in the int64 columns here, so the record is DROPPED by the reader and
 *   counted (tsf_csv_malformed) -- never emitted under a made-up key.  Without the '?' the first such
 *   line fails the read (FAILFAST).
 * Returns 0; TSF_CSV_E_OPEN / TSF_CSV_E_PARSE with *err_file (index into paths) and *err_line
 * (1-based) set; -1 bad arguments, -2 out of memory, -3 other failure. */
enum { TSF_CSV_E_OPEN = -10, TSF_CSV_E_PARSE = -11, TSF_CSV_E_CODEC = -12 };
typedef struct tsf_csv tsf_csv;
int tsf_csv_read(int32_t n_files, const char *const *paths, const int64_t *series_id,
                 const char *layout, int32_t n_threads, tsf_csv **out, int64_t *n_rows,
                 int32_t *err_file, int64_t *err_line);
int tsf_csv_fetch(tsf_csv *t, int64_t *series_id, int64_t *dim_id, int64_t *ds, double *y);
/* The table's own columns ([n_rows] each, valid until tsf_csv_free): a caller that can adopt foreign memory
 * (numpy can) saves the copy of tsf_csv_fetch. */
int tsf_csv_columns(tsf_csv *t, const int64_t **series_id, const int64_t **dim_id, const int64_t **ds,
                    const double **y);
int64_t tsf_csv_malformed(const tsf_csv *t);     /* records dropped in permissive mode */
void tsf_csv_free(tsf_csv *t);

/* ---- input discovery (host side) -----------------------------------------------------------
 * What spark.read.csv(path) lists under a directory (prophet_modeler.py:109-114): every regular file below
 * `root` whose name, and every directory's name on the way, does not start with '_' or '.' (_SUCCESS, .crc,
 * _temporary); a `series_id=<int>` directory supplies series_id for the files below it (Hive partition
 * discovery).  A root that is a regular file is that file alone.  The files come out partitioned ones first,
 * by partition value, then by path -- so that the reader's rows arrive grouped --: *n_partitioned of the
 * *n_files.  tsf_csv_dir_paths / _series_id can be handed to tsf_csv_read as they are (layout "dtq" for the
 * first n_partitioned, "sdtq" for the rest).
 * Returns 0, or with the handle still valid (free it) TSF_CSV_E_OPEN (a directory could not be read),
 * TSF_CSV_E_PARSE (a `series_id=` directory whose value is not an integer) or TSF_CSV_E_CODEC (a part in a
 * codec the reader does not inflate: .bz2 .snappy .lz4 .zst .xz; .gz and .deflate are read) with the offending
 * path in tsf_csv_dir_error_path; -1 bad arguments, -2 out of memory. */
typedef struct tsf_csv_dir tsf_csv_dir;
int tsf_csv_discover(const char *root, int32_t n_threads, tsf_csv_dir **out, int32_t *n_files,
                     int32_t *n_partitioned);
/* The walk and the read in one pass (round 4): tsf_csv_discover_load lists like tsf_csv_discover and the thread that
 * lists a directory reads its files straight away (no second pass over 10 000 paths: the 16 ms walk disappears under
 * the reads); tsf_csv_read_loaded(d, first, count, ...) is tsf_csv_read over files [first, first + count) of the
 * sorted list with the bytes already in memory (a range can be handed over once; err_file is relative to `first`).
 * Files in a codec the reader refuses are listed but not loaded (TSF_CSV_E_CODEC from the discover call). */
int tsf_csv_discover_load(const char *root, int32_t n_threads, tsf_csv_dir **out, int32_t *n_files,
                          int32_t *n_partitioned);
int tsf_csv_read_loaded(tsf_csv_dir *d, int32_t first, int32_t count, const char *layout, int32_t n_threads,
                        tsf_csv **out, int64_t *n_rows, int32_t *err_file, int64_t *err_line);
/* The input directory in chunks (round 6): tsf_csv_root_open lists the CHILDREN of `root` once (hidden names skipped;
 * `series_id=<int>` directories first, by value, then the rest by name); tsf_csv_root_load is tsf_csv_discover_load over
 * the subtrees of children [first, first + count) -- its handle goes to tsf_csv_read_loaded / tsf_csv_dir_* as usual.
 * A job reads, fits and persists such ranges as a pipeline (files of chunk k + 1 are read while chunk k is on the GPU).
 * *hive_only = 1 when every child is a `series_id=<int>` directory and no value occurs twice -- the layout the reference
 * reads (spark.read.csv over `series_id=751/...`, prophet_modeler.py:109-114): ranges of children then hold disjoint
 * series, so fitting range by range fits every series once, on all of its rows.  *nested = 1 from a load that met a
 * `series_id=` directory below another one with a different value (the same series could then sit in two ranges): read
 * the tree whole instead.  Returns as tsf_csv_discover; tsf_csv_root_open also TSF_CSV_E_OPEN if root cannot be listed. */
typedef struct tsf_csv_root tsf_csv_root;
int tsf_csv_root_open(const char *root, tsf_csv_root **out, int32_t *n_children, int32_t *hive_only);
int tsf_csv_root_load(tsf_csv_root *r, int32_t first, int32_t count, int32_t n_threads, tsf_csv_dir **out,
                      int32_t *n_files, int32_t *n_partitioned, int32_t *nested);
void tsf_csv_root_free(tsf_csv_root *r);
const char *const *tsf_csv_dir_paths(const tsf_csv_dir *d);
const int64_t *tsf_csv_dir_series_id(const tsf_csv_dir *d);
const char *tsf_csv_dir_error_path(const tsf_csv_dir *d);
void tsf_csv_dir_free(tsf_csv_dir *d);

/* ---- forecast sink (host side) -------------------------------------------------------------
 * ProphetScorer.convert_forecasts + write_forecasts in one pass
 * (/root/reference/src/jobs/prophet_scorer.py:130-150): a CSV with header
 *   created_timestamp,series_id,dim_id,forecast_date,forecast_timestamp,forecast_quantity
 * forecast_date = the date of ds as %Y-%m-%d (:107-108); forecast_timestamp in Spark 2.4's default
 * CSV timestampFormat yyyy-MM-dd'T'HH:mm:ss.SSSXXX with the wall time taken as UTC
 * (2002-12-28T22:00:00.000Z).  ds: int64 ns since the epoch.  Overwrites `path`.
 * Returns 0, TSF_CSV_E_OPEN if the file cannot be written, -1 bad arguments, -2 out of memory. */
int tsf_csv_write_forecasts(const char *path, const char *created_timestamp, int64_t n,
                            const int64_t *series_id, const int64_t *dim_id, const int64_t *ds,
                            const int64_t *quantity, int32_t n_threads);
/* the same sink for int32 id / quantity columns (what the scorer's forecast frame holds: no widening copies) */
int tsf_csv_write_forecasts_i32(const char *path, const char *created_timestamp, int64_t n,
                                const int32_t *series_id, const int32_t *dim_id, const int64_t *ds,
                                const int32_t *quantity, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif /* TSF_H */
