/*
 * oracle/prophet_canon.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; never shipped, never the
 * thing measured).  PARITY UNPINNED w.r.t. real fbprophet/pystan output: see
 * oracle/fbprophet_restated.py.
 *
 * The complete per-series path the reference runs at
 *   /root/reference/src/jobs/prophet_modeler.py:56-66   (floor/cap, Prophet(...).fit)
 *   /root/reference/src/jobs/prophet_scorer.py:64-70    (make_future_dataframe, predict)
 * restated in plain C in ONE FIXED ARITHMETIC ORDER ("canonical W64 arithmetic"):
 *
 *   - every sum over time is taken as 64 contiguous chunk partials (chunk length
 *     NT = ceil(T/64), accumulated with fma from the LAST element of the chunk to the
 *     first), combined by an xor-butterfly (offsets 1,2,4,8,16,32 for scalars; 32,16,1,2,4,8
 *     for the per-column sums X^T r) or, for the trend gradient, a 16-wide Hillis-Steele suffix
 *     scan plus a carry over the four groups of 16;
 *   - every dot product over parameters is 64 partials (parameter p lives in slot p%64,
 *     second term fma'd for p>=64) combined by the same xor-butterfly;
 *   - exp/log/sin/cos come from det_math.h; everything else is IEEE double + - * / sqrt
 *     and fma, compiled with -ffp-contract=off.
 *
 * The HIP kernels in time_series_spark_amd/csrc/ execute exactly this operation sequence
 * (one wavefront = the 64 chunks/slots), so that product and oracle agree bit-for-bit and the
 * north-star 1e-4 forecast tolerance is meaningful despite the optimiser being chaotic.
 * Per-evaluation agreement of this canonical form with the literal dense-A Stan form
 * (fbprophet_restated.stan_neg_log_prob_grad) is asserted in tests/test_oracle.py.
 *
 * Upstream routines followed (fbprophet 0.5 forecaster.py / prophet.stan / stan 2.19
 * optimization/*.hpp; UPSTREAM-RECALL, not in /root/reference):
 *   cn_prepare      setup_dataframe, initialize_scales, set_changepoints,
 *                   make_all_seasonality_features (fourier_series), *_growth_init
 *   cn_eval         prophet.stan model block (-log_prob and gradient)
 *   cn_lbfgs        BFGSMinimizer<LBFGSUpdate>::step, WolfeLineSearch, WolfLSZoom, CubicInterp
 *   cn_predict      predict_trend (piecewise_linear / piecewise_logistic),
 *                   predict_seasonal_components, yhat = trend*(1+mult)+add
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "det_math.h"

#define CN_W 64
#define CN_MAX_S 64
#define CN_MAX_SEAS 8
#define CN_MAX_EXTRA 64
#define CN_MAX_P 128
#define CN_MAX_HIST 8

typedef struct {
    int32_t growth;                 /* 0 linear, 1 logistic */
    int32_t n_changepoints;         /* 25 */
    double changepoint_range;       /* 0.8 */
    double tau;                     /* changepoint_prior_scale 0.05 */
    int32_t n_seas;
    int32_t n_extra;
    double seas_period[CN_MAX_SEAS];
    double seas_prior[CN_MAX_SEAS];
    int32_t seas_order[CN_MAX_SEAS];
    int32_t seas_mode[CN_MAX_SEAS]; /* 0 additive, 1 multiplicative */
    double extra_prior[CN_MAX_EXTRA];
    int32_t extra_mode[CN_MAX_EXTRA];
    int32_t max_iter;               /* 10000 */
    int32_t history;                /* 5 */
    double init_alpha;              /* 1e-3 */
    double tol_obj, tol_rel_obj, tol_grad, tol_rel_grad, tol_param;
    int32_t eval_mode;              /* 0 residual form always; 1 quadratic (Gram) form where the
                                       model is linear in (k, m, delta, beta): linear growth,
                                       all columns additive */
    int32_t recenter_every;         /* quadratic form: re-centre at least every n accepted iters */
    double recenter_ratio;          /* ... and whenever |Z(theta-ref)|^2 > ratio * SSE(ref) */
} cn_spec;

typedef struct {
    int32_t status;   /* Stan TERM_* code, or <0 error */
    int32_t n_iter;
    int32_t n_eval;
    int32_t S;
    int32_t K;
    int32_t pad_;
    double f;
    double y_scale;
    double floor_;
    double cap_scaled;
    int64_t start_ns;
    int64_t t_scale_ns;
} cn_fitinfo;

enum { TERM_SUCCESS = 0, TERM_ABSX = 10, TERM_ABSF = 20, TERM_RELF = 21, TERM_ABSGRAD = 30,
       TERM_RELGRAD = 31, TERM_MAXIT = 40, TERM_LSFAIL = -1, CN_INIT_NONFINITE = -2,
       CN_EVAL_LIMIT = -3,   /* more than 64*max_iter+1024 evaluations: the line search never settled */
       CN_CONSTANT = 50, CN_ERR_TOO_FEW = -10, CN_ERR_CAP = -11, CN_ERR_SIZE = -12 };

typedef struct {
    int T, NT, S, K, Ka, P, growth;
    int S_out;               /* changepoints the caller sees; S = 1 > S_out = 0 when the fit runs on
                                fbprophet's dummy changepoint (set_changepoints: `changepoints_t =
                                np.array([0.])` when there are none) */
    double *t, *y, *X;       /* X: [T][K] internal column order (additive first) */
    int *cidx;
    int perm[CN_MAX_P];      /* internal column -> original column */
    double prior[CN_MAX_P];  /* prior scale per internal column */
    double t_change[CN_MAX_S];
    double cap, floor_, y_scale, tau;
    int64_t start_ns, tscale_ns;
    double k0, m0;
    int constant_y;
    int n_eval;
    int eval_limit;          /* evaluation budget of the whole fit (guard, see cn_lbfgs) */
    /* raw data-term pieces of the last residual-form evaluation: SSE and Z^T r */
    double last_sse, last_ztr[CN_MAX_P];
    /* quadratic (Gram) form state */
    int gram, since_rc, n_resid;
    double *M;               /* [P][P] Gram matrix Z^T Z (row/column 2 = log sigma: zero) */
    double ref[CN_MAX_P], cvec[CN_MAX_P], s0, last_q2;
    /* optional cross-check of every quadratic-form evaluation against the residual form */
    int check;
    double chk_f, chk_g;     /* max |df| / max(1,|f|),  max |dg|_2 / |g|_2 */
} cn_series;

/* ---- canonical reductions ----------------------------------------------------------- */

static double bfly_offsets(double v[CN_W], const int *offs, int noff)
{
    double n[CN_W];
    for (int o = 0; o < noff; ++o) {
        const int off = offs[o];
        for (int i = 0; i < CN_W; ++i) n[i] = v[i] + v[i ^ off];
        memcpy(v, n, sizeof(n));
    }
    return v[0];
}

/* scalar reductions: offsets 1,2,4,8,16,32 */
static double bfly(double v[CN_W])
{
    static const int offs[6] = {1, 2, 4, 8, 16, 32};
    return bfly_offsets(v, offs, 6);
}

/* per-column sums: offsets 32,16,1,2,4,8 (the order of the register transpose network) */
static double bfly_cols(double v[CN_W])
{
    static const int offs[6] = {32, 16, 1, 2, 4, 8};
    return bfly_offsets(v, offs, 6);
}

/* inclusive suffix sum over the 64 chunks: Hillis-Steele (1,2,4,8) inside each group of 16
 * (out-of-group terms are an explicit + 0.0), then the totals of the later groups are added
 * as one carry: group 0 += T1 + (T2 + T3), group 1 += T2 + T3, group 2 += T3, group 3 += 0.0 */
static void suffix_scan(double v[CN_W])
{
    double n[CN_W];
    for (int off = 1; off < 16; off <<= 1) {
        for (int L = 0; L < CN_W; ++L) n[L] = v[L] + (((L & 15) + off < 16) ? v[L + off] : 0.0);
        memcpy(v, n, sizeof(n));
    }
    const double t1 = v[16], t2 = v[32], t3 = v[48];
    const double s2 = t2 + t3;
    const double s1 = t1 + s2;
    for (int L = 0; L < CN_W; ++L) {
        const int row = L >> 4;
        const double carry = (row == 0) ? s1 : (row == 1 ? s2 : (row == 2 ? t3 : 0.0));
        v[L] = v[L] + carry;
    }
}

/* inclusive prefix sum over the 64 lanes, the mirror image of suffix_scan: Hillis-Steele (1,2,4,8) inside each
 * group of 16 (out-of-group terms an explicit + 0.0), then the totals of the EARLIER groups as one carry:
 * group 1 += T0, group 2 += T0 + T1, group 3 += (T0 + T1) + T2, group 0 += 0.0 */
static void prefix_scan(double v[CN_W])
{
    double n[CN_W];
    for (int off = 1; off < 16; off <<= 1) {
        for (int L = 0; L < CN_W; ++L) n[L] = v[L] + (((L & 15) >= off) ? v[L - off] : 0.0);
        memcpy(v, n, sizeof(n));
    }
    const double t0 = v[15], t1 = v[31], t2 = v[47];
    const double s1 = t0 + t1;
    const double s2 = s1 + t2;
    for (int L = 0; L < CN_W; ++L) {
        const int row = L >> 4;
        const double carry = (row == 0) ? 0.0 : (row == 1 ? t0 : (row == 2 ? s1 : s2));
        v[L] = v[L] + carry;
    }
}

/* Scans of affine maps x -> a x + b over the 64 lanes (round 5: the changepoint recurrences of the logistic trend
 * and their adjoint as scans instead of S-step chains).  compose(later, earlier) = later o earlier:
 * (a2, b2) o (a1, b1) = (a2 a1, fma(a2, b1, b2)); out-of-group operands are the identity (1, 0).
 * affine_prefix_scan: lane L ends with f_L o f_(L-1) o ... o f_0 -- Hillis-Steele (1,2,4,8) inside each group of 16
 * (later = own, earlier = lane - off), then the composed maps of the earlier groups as one carry, applied as the
 * EARLIER map: group 1: T0; group 2: T1 o T0; group 3: T2 o (T1 o T0) (T_r = lane 16 r + 15 after the in-group stages).
 * affine_suffix_scan: lane L ends with f_L o f_(L+1) o ... o f_63 -- the mirror image (own = the LATER-applied...
 * careful: in f_L o f_(L+1) the map of lane L is applied LAST, so own is `later` again and lane + off is `earlier`);
 * carry of group 2: T3, group 1: T2 o T3, group 0: T1 o (T2 o T3) (T_r = lane 16 r after the in-group stages). */
static void affine_prefix_scan(double a[CN_W], double b[CN_W])
{
    double na[CN_W], nb[CN_W];
    for (int off = 1; off < 16; off <<= 1) {
        for (int L = 0; L < CN_W; ++L) {
            const int in = (L & 15) >= off;
            const double ea = in ? a[L - off] : 1.0, eb = in ? b[L - off] : 0.0;
            na[L] = a[L] * ea;
            nb[L] = fma(a[L], eb, b[L]);
        }
        memcpy(a, na, sizeof(na)); memcpy(b, nb, sizeof(nb));
    }
    const double a0 = a[15], b0 = b[15], a1 = a[31], b1 = b[31], a2 = a[47], b2 = b[47];
    const double c2a = a1 * a0, c2b = fma(a1, b0, b1);           /* T1 o T0 */
    const double c3a = a2 * c2a, c3b = fma(a2, c2b, b2);         /* T2 o (T1 o T0) */
    for (int L = 0; L < CN_W; ++L) {
        const int row = L >> 4;
        const double ea = (row == 0) ? 1.0 : (row == 1 ? a0 : (row == 2 ? c2a : c3a));
        const double eb = (row == 0) ? 0.0 : (row == 1 ? b0 : (row == 2 ? c2b : c3b));
        na[L] = a[L] * ea;
        nb[L] = fma(a[L], eb, b[L]);
    }
    memcpy(a, na, sizeof(na)); memcpy(b, nb, sizeof(nb));
}

static void affine_suffix_scan(double a[CN_W], double b[CN_W])
{
    double na[CN_W], nb[CN_W];
    for (int off = 1; off < 16; off <<= 1) {
        for (int L = 0; L < CN_W; ++L) {
            const int in = (L & 15) + off < 16;
            const double ea = in ? a[L + off] : 1.0, eb = in ? b[L + off] : 0.0;
            na[L] = a[L] * ea;
            nb[L] = fma(a[L], eb, b[L]);
        }
        memcpy(a, na, sizeof(na)); memcpy(b, nb, sizeof(nb));
    }
    const double a1 = a[16], b1 = b[16], a2 = a[32], b2 = b[32], a3 = a[48], b3 = b[48];
    const double c1a = a2 * a3, c1b = fma(a2, b3, b2);           /* T2 o T3 */
    const double c0a = a1 * c1a, c0b = fma(a1, c1b, b1);         /* T1 o (T2 o T3) */
    for (int L = 0; L < CN_W; ++L) {
        const int row = L >> 4;
        const double ea = (row == 3) ? 1.0 : (row == 2 ? a3 : (row == 1 ? c1a : c0a));
        const double eb = (row == 3) ? 0.0 : (row == 2 ? b3 : (row == 1 ? c1b : c0b));
        na[L] = a[L] * ea;
        nb[L] = fma(a[L], eb, b[L]);
    }
    memcpy(a, na, sizeof(na)); memcpy(b, nb, sizeof(nb));
}

/* dot over (zero padded) 128-vectors */
static double dotc(const double *a, const double *b)
{
    double part[CN_W];
    for (int l = 0; l < CN_W; ++l) part[l] = fma(a[l + CN_W], b[l + CN_W], a[l] * b[l]);
    return bfly(part);
}

/* ---- column bookkeeping -------------------------------------------------------------- */

static int spec_K(const cn_spec *sp)
{
    int K = sp->n_extra;
    for (int s = 0; s < sp->n_seas; ++s) K += 2 * sp->seas_order[s];
    return K;
}

static void free_series(cn_series *se)
{
    if (!se) return;
    free(se->t); free(se->y); free(se->X); free(se->cidx); free(se->M); free(se);
}

/* fbprophet fourier_series argument: 2.0*(i+1)*np.pi*t/period with
 * t = (1e-9 * ns) / 86400. (pandas 0.25 Timedelta.total_seconds = 1e-9 * asi8). */
/* Canonical design values (round 5): only the FIRST harmonic of a seasonality goes through
 * det_sincos, at fbprophet's own argument 2.0*1*pi*t/period; harmonic h + 1 follows from the
 * three-term recurrence u[h+1] = 2 cos(theta) u[h] - u[h-1] (sin and cos both satisfy it;
 * u[0] = 0 resp. 1), one fma per value.  A design row is then a function of two numbers per
 * seasonality, which is what the residual-form kernel keeps per row and expands in registers
 * (tsf_fit_kernels.h, eval_fg HARM) instead of streaming 2*order columns.  Against sin / cos of
 * fbprophet's argument of harmonic h the values differ by <= ~h times the rounding of the base
 * argument (<= 3e-11 for 15 years of daily seasonality; tests/test_oracle.py pins <= 1e-9). */
static void fourier_row(const cn_spec *sp, int64_t ns, double *row /* original order */)
{
    const double tdays = (1e-9 * (double)ns) / 86400.0;
    int col = 0;
    for (int s = 0; s < sp->n_seas; ++s) {
        const double arg = (2.0 * 3.141592653589793 * tdays) / sp->seas_period[s];
        double s1, c1;
        det_sincos(arg, &s1, &c1);
        const double c2 = 2.0 * c1;
        double sp_ = 0.0, cp_ = 1.0, sc = s1, cc = c1;
        for (int h = 0; h < sp->seas_order[s]; ++h) {
            if (h > 0) {
                const double sn = fma(c2, sc, -sp_), cn = fma(c2, cc, -cp_);
                sp_ = sc; cp_ = cc; sc = sn; cc = cn;
            }
            row[col] = sc; row[col + 1] = cc;
            col += 2;
        }
    }
}

static int build_perm(const cn_spec *sp, int K, int *perm, double *prior, int *Ka_out)
{
    int mode[CN_MAX_P];
    double pr[CN_MAX_P];
    int col = 0;
    for (int s = 0; s < sp->n_seas; ++s)
        for (int h = 0; h < 2 * sp->seas_order[s]; ++h) { mode[col] = sp->seas_mode[s]; pr[col] = sp->seas_prior[s]; col++; }
    for (int e = 0; e < sp->n_extra; ++e) { mode[col] = sp->extra_mode[e]; pr[col] = sp->extra_prior[e]; col++; }
    int n = 0;
    for (int j = 0; j < K; ++j) if (mode[j] == 0) { perm[n] = j; prior[n] = pr[j]; n++; }
    *Ka_out = n;
    for (int j = 0; j < K; ++j) if (mode[j] != 0) { perm[n] = j; prior[n] = pr[j]; n++; }
    return 0;
}

/* ---- setup --------------------------------------------------------------------------- */

static cn_series *cn_prepare(const cn_spec *sp, int T, const int64_t *ds, const double *y,
                             double floor_in, double cap_in, const double *extra, int *err)
{
    *err = 0;
    if (T < 2) { *err = CN_ERR_TOO_FEW; return NULL; }
    const int K = spec_K(sp);
    cn_series *se = (cn_series *)calloc(1, sizeof(cn_series));
    se->T = T; se->NT = (T + CN_W - 1) / CN_W; se->growth = sp->growth; se->K = K; se->tau = sp->tau;
    se->t = (double *)calloc(T, sizeof(double));
    se->y = (double *)calloc(T, sizeof(double));
    se->cidx = (int *)calloc(T, sizeof(int));
    se->X = (double *)calloc((size_t)T * (K > 0 ? K : 1), sizeof(double));
    /* initialize_scales */
    const double floor_ = (sp->growth == 1) ? floor_in : 0.0;
    double ys = 0.0, ymin = y[0], ymax = y[0];
    for (int i = 0; i < T; ++i) {
        double a = fabs(y[i] - floor_);
        if (a > ys) ys = a;
        if (y[i] < ymin) ymin = y[i];
        if (y[i] > ymax) ymax = y[i];
    }
    if (ys == 0.0) ys = 1.0;
    se->y_scale = ys; se->floor_ = floor_;
    se->constant_y = (ymin == ymax) && sp->growth == 0;
    se->start_ns = ds[0]; se->tscale_ns = ds[T - 1] - ds[0];
    if (sp->growth == 1) {
        if (cap_in <= floor_) { *err = CN_ERR_CAP; free_series(se); return NULL; }
        se->cap = (cap_in - floor_) / ys;
    }
    const double tsc = (double)se->tscale_ns;
    for (int i = 0; i < T; ++i) {
        se->t[i] = (double)(ds[i] - se->start_ns) / tsc;
        se->y[i] = (y[i] - floor_) / ys;
    }
    /* set_changepoints */
    int hist = (int)floor((double)T * sp->changepoint_range);
    int S = sp->n_changepoints;
    if (S + 1 > hist) S = hist - 1;
    if (S < 0) S = 0;
    if (S > CN_MAX_S || 3 + S > CN_W - 1) { *err = CN_ERR_SIZE; free_series(se); return NULL; }     /* (the deltas live in lanes 3 .. 3 + S - 1) */
    se->S_out = S;
    if (S == 0) {
        /* fbprophet set_changepoints: no changepoints -> one dummy changepoint at t = 0.  The
         * Stan model is fitted with S = 1 (delta_1 under the Laplace prior, A[:, 0] = 1 for every
         * row); Prophet.fit then folds it away: k += delta, delta = 0 (to_original). */
        S = 1;
        se->t_change[0] = 0.0;
    }
    se->S = S;
    if (se->S_out > 0) {
        const double step = (double)(hist - 1) / (double)S;
        for (int j = 1; j <= S; ++j) {
            double v = (j == S) ? (double)(hist - 1) : (double)j * step;
            int idx = (int)rint(v);
            se->t_change[j - 1] = se->t[idx];
        }
    }
    for (int i = 0; i < T; ++i) {
        int c = 0;
        while (c < S && se->t[i] >= se->t_change[c]) ++c;
        se->cidx[i] = c;
    }
    se->P = 3 + S + K;
    if (se->P > CN_MAX_P || K > CN_MAX_P) { *err = CN_ERR_SIZE; free_series(se); return NULL; }
    /* design matrix, internal column order */
    build_perm(sp, K, se->perm, se->prior, &se->Ka);
    {
        double row[CN_MAX_P];
        const int nf = K - sp->n_extra;
        for (int i = 0; i < T; ++i) {
            fourier_row(sp, ds[i], row);
            for (int e = 0; e < sp->n_extra; ++e) row[nf + e] = extra[(size_t)e * T + i];
            for (int j = 0; j < K; ++j) se->X[(size_t)i * K + j] = row[se->perm[j]];
        }
    }
    /* growth init: first row with min ds, first row with max ds (idxmin / idxmax) */
    int i0 = 0, i1 = T - 1;
    while (i1 > 0 && ds[i1 - 1] == ds[T - 1]) --i1;
    const double Td = se->t[i1] - se->t[i0];
    if (sp->growth == 0) {
        se->k0 = (se->y[i1] - se->y[i0]) / Td;
        se->m0 = se->y[i0] - se->k0 * se->t[i0];
    } else {
        const double C0 = se->cap, C1 = se->cap;
        double y0 = fmax(0.01 * C0, fmin(0.99 * C0, se->y[i0]));
        double y1 = fmax(0.01 * C1, fmin(0.99 * C1, se->y[i1]));
        double r0 = C0 / y0, r1 = C1 / y1;
        if (fabs(r0 - r1) <= 0.01) r0 = 1.05 * r0;
        const double L0 = det_log(r0 - 1.0), L1 = det_log(r1 - 1.0);
        se->m0 = L0 * Td / (L0 - L1);
        se->k0 = (L0 - L1) / Td;
    }
    return se;
}

/* ---- -log_prob and gradient, canonical W64 arithmetic --------------------------------- */
/* th, g: zero padded 128-vectors, internal layout [k, m, log sigma, delta[S], beta_int[K]]. */

static int cn_eval(cn_series *se, const double *th, double *f_out, double *g)
{
    const int T = se->T, NT = se->NT, S = se->S, K = se->K, Ka = se->Ka;
    const double k = th[0], m = th[1], ls = th[2];
    const double *delta = th + 3, *beta = th + 3 + S;
    se->n_eval++;
    const double sigma = det_exp(ls);
    const double inv_s2 = 1.0 / (sigma * sigma);
    double ks[CN_MAX_S + 1], mc[CN_MAX_S + 1];
    ks[0] = k;
    for (int j = 0; j < S; ++j) ks[j + 1] = ks[j] + delta[j];
    mc[0] = m;
    if (se->growth == 0) {
        for (int j = 0; j < S; ++j) mc[j + 1] = mc[j] + ((-se->t_change[j]) * delta[j]);
    } else {
        /* Logistic growth, round 5: the two S-step chains as scans over the parameter lanes (delta_j lives in lane
         * 3 + j, as in theta): ks[j+1] = k + (inclusive prefix sum of delta)[3 + j]  (prophet.stan:
         * k + cumulative_sum(delta)); the offset recurrence m[j+1] = m[j] + (t_j - m[j]) (1 - ks[j] / ks[j+1]) is the
         * affine map m -> rho_j m + t_j (1 - rho_j), rho_j = ks[j] / ks[j+1], and m[j+1] = A m + B with (A, B) the
         * prefix composition of those maps. */
        double pd[CN_W], fa[CN_W], fb[CN_W];
        for (int L = 0; L < CN_W; ++L) pd[L] = (L >= 3 && L < 3 + S) ? delta[L - 3] : 0.0;
        prefix_scan(pd);
        for (int j = 0; j < S; ++j) ks[j + 1] = k + pd[3 + j];
        for (int L = 0; L < CN_W; ++L) {
            fa[L] = 1.0; fb[L] = 0.0;
            if (L >= 3 && L < 3 + S) {
                const int j = L - 3;
                const double rho = ks[j] / ks[j + 1];
                fa[L] = rho;
                fb[L] = se->t_change[j] * (1.0 - rho);
            }
        }
        affine_prefix_scan(fa, fb);
        for (int j = 0; j < S; ++j) mc[j + 1] = fma(fa[3 + j], m, fb[3 + j]);
    }
    double sseL[CN_W], tot1[CN_W], tot2[CN_W];
    double tp1[CN_MAX_S + 1], tp2[CN_MAX_S + 1];
    static __thread double accL[CN_W][CN_MAX_P];
    for (int j = 0; j < S; ++j) tp1[j] = tp2[j] = 0.0;
    for (int L = 0; L < CN_W; ++L) {
        const int lo = L * NT, hi = (lo + NT < T) ? lo + NT : T;
        double sse = 0.0, rt1 = 0.0, rt2 = 0.0;
        double *acc = accL[L];
        for (int j = 0; j < K; ++j) acc[j] = 0.0;
        for (int i = hi - 1; i >= lo; --i) {
            const int c = se->cidx[i];
            const double *x = se->X + (size_t)i * K;
            const double ti = se->t[i];
            double xa = 0.0, xm = 0.0;
            for (int j = 0; j < Ka; ++j) xa = fma(x[j], beta[j], xa);
            for (int j = Ka; j < K; ++j) xm = fma(x[j], beta[j], xm);
            double gtr, q = 0.0;
            if (se->growth == 0) {
                gtr = fma(ks[c], ti, mc[c]);
            } else {
                const double z = ks[c] * (ti - mc[c]);
                const double e = det_exp(-z);
                const double sg = 1.0 / (1.0 + e);
                gtr = se->cap * sg;
                q = gtr * (1.0 - sg);
            }
            const double opm = 1.0 + xm;
            const double mu = fma(gtr, opm, xa);
            const double r = se->y[i] - mu;
            sse = fma(r, r, sse);
            for (int j = 0; j < Ka; ++j) acc[j] = fma(x[j], r, acc[j]);
            if (Ka < K) {
                const double rg = r * gtr;
                for (int j = Ka; j < K; ++j) acc[j] = fma(x[j], rg, acc[j]);
            }
            double v = r * opm;
            if (se->growth == 1) v = v * q;
            rt1 = fma(v, ti, rt1);
            rt2 = rt2 + v;
            const int cprev = (i > 0) ? se->cidx[i - 1] : 0;
            for (int j = cprev; j < c; ++j) { tp1[j] = rt1; tp2[j] = rt2; }
        }
        sseL[L] = sse; tot1[L] = rt1; tot2[L] = rt2;
    }
    const double sse = bfly(sseL);
    suffix_scan(tot1);
    suffix_scan(tot2);
    const double TA = tot1[0], TB = tot2[0];
    double SA[CN_MAX_S + 1], SB[CN_MAX_S + 1];
    for (int j = 0; j < S; ++j) {
        /* chunk holding the first point with t >= t_change[j] */
        int fj = 0;
        while (fj < T && se->cidx[fj] <= j) ++fj;
        const int Lj = fj / NT;
        const double e1 = (Lj + 1 < CN_W) ? tot1[Lj + 1] : 0.0;
        const double e2 = (Lj + 1 < CN_W) ? tot2[Lj + 1] : 0.0;
        SA[j] = tp1[j] + e1;
        SB[j] = tp2[j] + e2;
    }
    /* per-column sums over the 64 chunk partials */
    double ACC[CN_MAX_P];
    for (int j = 0; j < K; ++j) {
        double col[CN_W];
        for (int L = 0; L < CN_W; ++L) col[L] = accL[L][j];
        ACC[j] = bfly_cols(col);
    }
    /* priors */
    double pa[CN_W], pb[CN_W];
    for (int l = 0; l < CN_W; ++l) { pa[l] = 0.0; pb[l] = 0.0; }
    for (int p = 0; p < se->P; ++p) {
        const int l = p % CN_W;
        if (p >= 3 && p < 3 + S) pa[l] = pa[l] + fabs(th[p]);
        if (p >= 3 + S) { const double qq = th[p] / se->prior[p - 3 - S]; pb[l] = fma(qq, qq, pb[l]); }
    }
    const double sabs = bfly(pa), sb = bfly(pb);
    const double s2 = sigma * sigma;
    double f = ((0.5 * k) * k) / 25.0 + ((0.5 * m) * m) / 25.0;
    f = f + sabs / se->tau;
    f = f + 2.0 * s2;
    f = f + 0.5 * sb;
    f = f + (double)T * ls;
    f = f + (0.5 * sse) * inv_s2;

    for (int p = 0; p < CN_MAX_P; ++p) g[p] = 0.0;
    const double nis = -inv_s2;
    double gk, gm;
    double *gd = g + 3;
    if (se->growth == 0) {
        for (int j = 0; j < S; ++j) {
            const double zr = SA[j] - se->t_change[j] * SB[j];
            se->last_ztr[3 + j] = zr;
            gd[j] = nis * zr;
        }
        se->last_ztr[0] = TA; se->last_ztr[1] = TB; se->last_ztr[2] = 0.0;
        for (int j = 0; j < K; ++j) se->last_ztr[3 + S + j] = ACC[j];
        se->last_sse = sse;
        gk = nis * TA;
        gm = nis * TB;
    } else {
        double D1[CN_MAX_S + 1], D2[CN_MAX_S + 1];
        for (int c = 0; c <= S; ++c) {
            const double hiA = (c == 0) ? TA : SA[c - 1], hiB = (c == 0) ? TB : SB[c - 1];
            const double loA = (c == S) ? 0.0 : SA[c], loB = (c == S) ? 0.0 : SB[c];
            const double A = hiA - loA, B = hiB - loB;
            D1[c] = A - mc[c] * B;
            D2[c] = -(ks[c] * B);
        }
        /* Reverse sweep through the offset recurrence, round 5: abar[c] = D2[c] + rho_c abar[c+1] (abar[S] = D2[S])
         * as a suffix composition of the maps x -> rho_c x + D2[c] over the lanes c = 0 .. S (lane S: the constant map
         * x -> D2[S]); rho_bar[c] = abar[c+1] (t_c - mc[c]); the two adjustments of D1 lane-parallel (own first, then
         * the one from the segment before); the slope gradient's running sums as one suffix scan. */
        double fa[CN_W], fb[CN_W], rb[CN_MAX_S + 1], ab[CN_W];
        for (int L = 0; L < CN_W; ++L) {
            fa[L] = 1.0; fb[L] = 0.0;
            if (L < S) { fa[L] = ks[L] / ks[L + 1]; fb[L] = D2[L]; }
            else if (L == S) { fa[L] = 0.0; fb[L] = D2[S]; }
        }
        affine_suffix_scan(fa, fb);                       /* abar[c] = fb[c] */
        for (int c = 0; c < S; ++c) rb[c] = fb[c + 1] * (se->t_change[c] - mc[c]);
        for (int L = 0; L < CN_W; ++L) ab[L] = 0.0;
        for (int c = 0; c <= S; ++c) {
            double d = D1[c];
            if (c < S) d = d + rb[c] * (-1.0 / ks[c + 1]);
            if (c >= 1) d = d + rb[c - 1] * ((ks[c - 1] / ks[c]) / ks[c]);
            ab[c] = d;
        }
        suffix_scan(ab);
        for (int c = S; c >= 1; --c) gd[c - 1] = nis * ab[c];
        gk = nis * ab[0];
        gm = nis * fb[0];
    }
    g[0] = gk + k / 25.0;
    g[1] = gm + m / 25.0;
    g[2] = ((double)T - sse * inv_s2) + 4.0 * s2;
    for (int j = 0; j < S; ++j) {
        const double sgn = (double)((delta[j] > 0.0) - (delta[j] < 0.0));
        gd[j] = gd[j] + sgn / se->tau;
    }
    double *gb = g + 3 + S;
    for (int j = 0; j < K; ++j) gb[j] = nis * ACC[j] + beta[j] / (se->prior[j] * se->prior[j]);
    *f_out = f;
    if (!isfinite(f)) return 2;
    for (int p = 0; p < se->P; ++p) if (!isfinite(g[p])) return 3;
    return 0;
}


/* ---- quadratic (Gram) form of the data term, linear growth + additive columns only --------
 *
 * With linear growth and only additive columns the mean is LINEAR in (k, m, delta, beta):
 * mu = Z theta, Z = [t, 1, (t - s_j)+ ..., X].  Around a reference point `ref` with residual
 * r_ref = y - Z ref, s0 = |r_ref|^2, c = Z^T r_ref (all three produced by ONE residual-form
 * evaluation, cn_eval):
 *     SSE(theta)   = s0 - 2 c.D + D.M D,     Z^T r(theta) = c - M D,     D = theta - ref,
 * with M = Z^T Z.  This is the same normal log-likelihood prophet.stan evaluates (algebra, not
 * an approximation); the rounding differs from the residual form, and stays at the level of
 * the residual form's as long as |Z D|^2 is not large against s0 -- hence the re-centring rule
 * in cn_lbfgs (a residual-form evaluation at the accepted iterate becomes the new reference).
 *
 * Canonical order: M's column q is "Z^T r" of the residual machinery with r := column q of Z
 * (cn_ztr below: same chunk partials, scans and butterflies as cn_eval); the mat-vec
 * (M D)[p] is four interleaved fma chains over q (q mod 4), combined (a0 + a1) + (a2 + a3).
 */

/* Z^T r and r.r for a given weight vector r[T], in cn_eval's exact operation order */
static void cn_ztr(const cn_series *se, const double *r_in, double *ztr, double *sse_out)
{
    const int T = se->T, NT = se->NT, S = se->S, K = se->K;
    double sseL[CN_W], tot1[CN_W], tot2[CN_W];
    double tp1[CN_MAX_S + 1], tp2[CN_MAX_S + 1];
    static __thread double accL[CN_W][CN_MAX_P];
    for (int j = 0; j < S; ++j) tp1[j] = tp2[j] = 0.0;
    for (int L = 0; L < CN_W; ++L) {
        const int lo = L * NT, hi = (lo + NT < T) ? lo + NT : T;
        double sse = 0.0, rt1 = 0.0, rt2 = 0.0;
        double *acc = accL[L];
        for (int j = 0; j < K; ++j) acc[j] = 0.0;
        for (int i = hi - 1; i >= lo; --i) {
            const int c = se->cidx[i];
            const double *x = se->X + (size_t)i * K;
            const double r = r_in[i];
            sse = fma(r, r, sse);
            for (int j = 0; j < K; ++j) acc[j] = fma(x[j], r, acc[j]);
            rt1 = fma(r, se->t[i], rt1);
            rt2 = rt2 + r;
            const int cprev = (i > 0) ? se->cidx[i - 1] : 0;
            for (int j = cprev; j < c; ++j) { tp1[j] = rt1; tp2[j] = rt2; }
        }
        sseL[L] = sse; tot1[L] = rt1; tot2[L] = rt2;
    }
    *sse_out = bfly(sseL);
    suffix_scan(tot1);
    suffix_scan(tot2);
    ztr[0] = tot1[0]; ztr[1] = tot2[0]; ztr[2] = 0.0;
    for (int j = 0; j < S; ++j) {
        int fj = 0;
        while (fj < T && se->cidx[fj] <= j) ++fj;
        const int Lj = fj / NT;
        const double e1 = (Lj + 1 < CN_W) ? tot1[Lj + 1] : 0.0;
        const double e2 = (Lj + 1 < CN_W) ? tot2[Lj + 1] : 0.0;
        const double SA = tp1[j] + e1, SB = tp2[j] + e2;
        ztr[3 + j] = SA - se->t_change[j] * SB;
    }
    for (int j = 0; j < K; ++j) {
        double col[CN_W];
        for (int L = 0; L < CN_W; ++L) col[L] = accL[L][j];
        ztr[3 + S + j] = bfly_cols(col);
    }
}

/* column p of Z at row i */
static double cn_zcol(const cn_series *se, int p, int i)
{
    if (p == 0) return se->t[i];
    if (p == 1) return 1.0;
    if (p == 2) return 0.0;
    if (p < 3 + se->S) return (se->cidx[i] > p - 3) ? se->t[i] - se->t_change[p - 3] : 0.0;
    return se->X[(size_t)i * se->K + (p - 3 - se->S)];
}

static void cn_build_gram(cn_series *se)
{
    const int P = se->P, T = se->T;
    se->M = (double *)calloc((size_t)P * P, sizeof(double));
    double *z = (double *)calloc(T, sizeof(double));
    double col[CN_MAX_P], dummy;
    for (int q = 0; q < P; ++q) {
        if (q == 2) continue;
        for (int i = 0; i < T; ++i) z[i] = cn_zcol(se, q, i);
        cn_ztr(se, z, col, &dummy);
        for (int p = 0; p < P; ++p) se->M[(size_t)p * P + q] = col[p];     /* M[p][q] */
    }
    free(z);
}

static void cn_set_ref(cn_series *se, const double *th)
{
    for (int p = 0; p < CN_MAX_P; ++p) { se->ref[p] = 0.0; se->cvec[p] = 0.0; }
    for (int p = 0; p < se->P; ++p) { se->ref[p] = th[p]; se->cvec[p] = se->last_ztr[p]; }
    se->ref[2] = 0.0; se->cvec[2] = 0.0;
    se->s0 = se->last_sse;
    se->since_rc = 0;
    se->n_resid++;
}

/* f and gradient from the data-term pieces (SSE, Z^T r), quadratic-path form: the prior
 * terms use reciprocals computed once per series (1/25, 1/tau, 1/prior, 1/prior^2) and one
 * uniform per-parameter gradient formula
 *     g_p = fma(theta_p, lc_p, nis * ztr_p) + sgn(theta_p) * sc_p        (p != 2)
 * (lc_p = 1/25 for k,m; 0 for delta; 1/prior^2 for beta;  sc_p = 1/tau for delta, else 0). */
static int cn_assemble_q(const cn_series *se, const double *th, double sse, const double *ztr,
                         double *f_out, double *g)
{
    const int T = se->T, S = se->S, P = se->P;
    const double k = th[0], m = th[1], ls = th[2];
    const double C25 = 1.0 / 25.0, inv_tau = 1.0 / se->tau;
    const double sigma = det_exp(ls);
    const double s2 = sigma * sigma;
    const double inv_s2 = 1.0 / s2;
    double pa[CN_W], pb[CN_W];
    for (int l = 0; l < CN_W; ++l) { pa[l] = 0.0; pb[l] = 0.0; }
    for (int p = 0; p < P; ++p) {
        const int l = p % CN_W;
        if (p >= 3 && p < 3 + S) pa[l] = pa[l] + fabs(th[p]);
        if (p >= 3 + S) { const double qq = th[p] * (1.0 / se->prior[p - 3 - S]); pb[l] = fma(qq, qq, pb[l]); }
    }
    const double sabs = bfly(pa), sb = bfly(pb);
    double f = ((0.5 * k) * k) * C25 + ((0.5 * m) * m) * C25;
    f = f + sabs * inv_tau;
    f = f + 2.0 * s2;
    f = f + 0.5 * sb;
    f = f + (double)T * ls;
    f = f + (0.5 * sse) * inv_s2;
    for (int p = 0; p < CN_MAX_P; ++p) g[p] = 0.0;
    const double nis = -inv_s2;
    for (int p = 0; p < P; ++p) {
        if (p == 2) { g[2] = ((double)T - sse * inv_s2) + 4.0 * s2; continue; }
        double lc, sc;
        if (p < 2) { lc = C25; sc = 0.0; }
        else if (p < 3 + S) { lc = 0.0; sc = inv_tau; }
        else { const double pr = se->prior[p - 3 - S]; lc = 1.0 / (pr * pr); sc = 0.0; }
        const double sgn = (double)((th[p] > 0.0) - (th[p] < 0.0));
        g[p] = fma(th[p], lc, nis * ztr[p]) + sgn * sc;
    }
    *f_out = f;
    if (!isfinite(f)) return 2;
    for (int p = 0; p < P; ++p) if (!isfinite(g[p])) return 3;
    return 0;
}

/* residual-form evaluation on the quadratic path (initial point and re-centring): r = y - mu
 * with mu = (ks[c] t + mc[c]) + x.beta exactly as cn_eval computes it for linear growth /
 * additive columns, then Z^T r by cn_ztr, then cn_assemble_q.  Leaves SSE and Z^T r in
 * se->last_sse / se->last_ztr for cn_set_ref. */
static int cn_resid_q(cn_series *se, const double *th, double *f_out, double *g)
{
    const int T = se->T, S = se->S, K = se->K;
    const double *delta = th + 3, *beta = th + 3 + S;
    se->n_eval++;
    double ks[CN_MAX_S + 1], mc[CN_MAX_S + 1];
    ks[0] = th[0]; mc[0] = th[1];
    for (int j = 0; j < S; ++j) {
        ks[j + 1] = ks[j] + delta[j];
        mc[j + 1] = mc[j] + ((-se->t_change[j]) * delta[j]);
    }
    double *r = (double *)malloc(sizeof(double) * T);
    for (int i = 0; i < T; ++i) {
        const double *x = se->X + (size_t)i * K;
        double xa = 0.0;
        for (int j = 0; j < K; ++j) xa = fma(x[j], beta[j], xa);
        const double gtr = fma(ks[se->cidx[i]], se->t[i], mc[se->cidx[i]]);
        r[i] = se->y[i] - (gtr + xa);
    }
    cn_ztr(se, r, se->last_ztr, &se->last_sse);
    free(r);
    return cn_assemble_q(se, th, se->last_sse, se->last_ztr, f_out, g);
}

static int cn_eval_gram(cn_series *se, const double *th, double *f_out, double *g)
{
    const int P = se->P;
    se->n_eval++;
    double D[CN_MAX_P], v[CN_MAX_P], ztr[CN_MAX_P];
    for (int p = 0; p < CN_MAX_P; ++p) { D[p] = 0.0; v[p] = 0.0; ztr[p] = 0.0; }
    for (int p = 0; p < P; ++p) D[p] = (p == 2) ? 0.0 : th[p] - se->ref[p];
    for (int p = 0; p < P; ++p) {
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        const double *row = se->M + (size_t)p * P;
        for (int q = 0; q < P; ++q) a[q & 3] = fma(row[q], D[q], a[q & 3]);
        v[p] = (a[0] + a[1]) + (a[2] + a[3]);
    }
    const double q2 = dotc(D, v);
    const double cd = dotc(se->cvec, D);
    const double sse = fma(-2.0, cd, se->s0) + q2;
    se->last_q2 = q2;
    for (int p = 0; p < P; ++p) ztr[p] = se->cvec[p] - v[p];
    return cn_assemble_q(se, th, sse, ztr, f_out, g);
}

static int cn_eval_any(cn_series *se, const double *th, double *f_out, double *g)
{
    if (!se->gram) return cn_eval(se, th, f_out, g);
    const int rc = cn_eval_gram(se, th, f_out, g);
    if (se->check && rc == 0) {
        double fr, gr[CN_MAX_P], dn = 0.0, gn = 0.0;
        const int ne = se->n_eval;
        if (cn_eval(se, th, &fr, gr) == 0) {
            for (int p = 0; p < se->P; ++p) { dn += (g[p] - gr[p]) * (g[p] - gr[p]); gn += gr[p] * gr[p]; }
            const double ef = fabs(*f_out - fr) / fmax(1.0, fabs(fr)), eg = sqrt(dn) / sqrt(gn);
            if (ef > se->chk_f) se->chk_f = ef;
            if (eg > se->chk_g) se->chk_g = eg;
        }
        se->n_eval = ne;
    }
    return rc;
}

/* ---- Stan L-BFGS ---------------------------------------------------------------------- */

static double cubic_interp6(double df0, double x1, double f1, double df1, double loX, double hiX)
{
    const double c3 = (-12.0 * f1 + 6.0 * x1 * (df0 + df1)) / (x1 * x1 * x1);
    const double c2 = -(4.0 * df0 + 2.0 * df1) / x1 + 6.0 * f1 / (x1 * x1);
    const double c1 = df0;
    const double t_s = sqrt(c2 * c2 - 2.0 * c1 * c3);
    const double s1 = -(c2 + t_s) / c3;
    const double s2 = -(c2 - t_s) / c3;
    double tmpF, minF, minX;
    minF = loX * (loX * (loX * c3 / 3.0 + c2) / 2.0 + c1);
    minX = loX;
    tmpF = hiX * (hiX * (hiX * c3 / 3.0 + c2) / 2.0 + c1);
    if (tmpF < minF) { minF = tmpF; minX = hiX; }
    if (loX < s1 && s1 < hiX) {
        tmpF = s1 * (s1 * (s1 * c3 / 3.0 + c2) / 2.0 + c1);
        if (tmpF < minF) { minF = tmpF; minX = s1; }
    }
    if (loX < s2 && s2 < hiX) {
        tmpF = s2 * (s2 * (s2 * c3 / 3.0 + c2) / 2.0 + c1);
        if (tmpF < minF) { minF = tmpF; minX = s2; }
    }
    return minX;
}

static void axpy_to(double *out, const double *x, double a, const double *p)
{
    for (int i = 0; i < CN_MAX_P; ++i) out[i] = fma(a, p[i], x[i]);
}

/* Line search written as ONE loop with ONE evaluation site (the HIP kernel has the same
 * shape); the control flow is WolfeLineSearch followed by WolfLSZoom. */
static int line_search(cn_series *se, double *alpha_io, double *x1, double *f1_out, double *g1,
                       const double *p, const double *x0, double f0, const double *g0)
{
    const double c1 = 1e-4, c2 = 0.9, minAlpha = 1e-12, min_range = 1e-16;
    const int maxLSIts = 20, maxLSRestarts = 10;
    const double dfp = dotc(g0, p);
    const double c1dfp = c1 * dfp, c2dfp = c2 * dfp;
    double alpha = *alpha_io;
    double alpha0 = minAlpha, prevF = f0, prevDFp = dfp;
    int nits = 0, lsRestarts = 0;
    int zoom = 0, itNum = 0;
    double alo = 0, aloF = 0, aloDFp = 0, ahi = 0, ahiF = 0, ahiDFp = 0;
    int rc = 0;
    for (;;) {
        if (!zoom) {
            if (nits >= maxLSIts) { rc = 1; break; }
        } else {
            itNum++;
            if (fabs(alo - ahi) < min_range) { rc = 1; break; }
            if (itNum % 5 == 0) {
                alpha = 0.5 * (alo + ahi);
            } else {
                const double d1 = aloDFp + ahiDFp - 3.0 * (aloF - ahiF) / (alo - ahi);
                double d2 = sqrt(d1 * d1 - aloDFp * ahiDFp);
                if (ahi < alo) d2 = -d2;
                alpha = ahi - (ahi - alo) * (ahiDFp + d2 - d1) / (ahiDFp - aloDFp + 2.0 * d2);
                const double lo = fmin(alo, ahi), hi = fmax(alo, ahi), w = fabs(alo - ahi);
                if (!isfinite(alpha) || alpha < lo + 0.01 * w || alpha > hi - 0.01 * w)
                    alpha = 0.5 * (alo + ahi);
            }
        }
        double f1, newDFp;
        int bad = 0;
        for (;;) {   /* evaluation with Stan's non-finite handling */
            if (se->n_eval >= se->eval_limit) return 2;      /* guard: budget exhausted */
            axpy_to(x1, x0, alpha, p);
            const int ret = cn_eval_any(se, x1, &f1, g1);
            if (ret == 0) break;
            if (!zoom) {
                if (lsRestarts >= maxLSRestarts) { bad = 1; break; }
                alpha = 0.5 * (alpha0 + alpha);
                lsRestarts++;
            } else {
                alpha = 0.5 * (alpha + fmin(alo, ahi));
                if (fabs(fmin(alo, ahi) - alpha) < min_range) { bad = 1; break; }
            }
        }
        if (bad) { rc = 1; break; }
        newDFp = dotc(g1, p);
        if (!zoom) {
            lsRestarts = 0;
            if (f1 > f0 + alpha * c1dfp || (f1 >= prevF && nits > 0)) {
                zoom = 1; alo = alpha0; aloF = prevF; aloDFp = prevDFp;
                ahi = alpha; ahiF = f1; ahiDFp = newDFp;
                continue;
            }
            if (fabs(newDFp) <= -c2dfp) { rc = 0; *f1_out = f1; break; }
            if (newDFp >= 0) {
                zoom = 1; alo = alpha; aloF = f1; aloDFp = newDFp;
                ahi = alpha0; ahiF = prevF; ahiDFp = prevDFp;
                continue;
            }
            alpha0 = alpha; prevF = f1; prevDFp = newDFp;
            alpha *= 10.0;
            nits++;
        } else {
            if (f1 > (f0 + alpha * c1dfp) || f1 >= aloF) {
                ahi = alpha; ahiF = f1; ahiDFp = newDFp;
            } else {
                if (fabs(newDFp) <= -c2dfp) { rc = 0; *f1_out = f1; break; }
                if (newDFp * (ahi - alo) >= 0) { ahi = alo; ahiF = aloF; ahiDFp = aloDFp; }
                alo = alpha; aloF = f1; aloDFp = newDFp;
            }
        }
    }
    *alpha_io = alpha;
    return rc;
}

static int cn_lbfgs(cn_series *se, const cn_spec *o, const double *theta0, double *theta_out,
                    cn_fitinfo *res)
{
    const int NP = CN_MAX_P;
    const int H = o->history > CN_MAX_HIST ? CN_MAX_HIST : o->history;
    const double eps = DBL_EPSILON;
    const double minAlpha = 1e-12;
    double xk[CN_MAX_P], gk[CN_MAX_P], pk[CN_MAX_P], xk_1[CN_MAX_P], gk_1[CN_MAX_P],
           pk_1[CN_MAX_P], sk[CN_MAX_P], yk[CN_MAX_P];
    double Sb[CN_MAX_HIST][CN_MAX_P], Yb[CN_MAX_HIST][CN_MAX_P];
    double rho[CN_MAX_HIST], alphas[CN_MAX_HIST];
    int hist_len = 0, hist_head = 0;
    double gammak = 1.0, fk, fk_1 = 0.0, alpha = o->init_alpha;
    int itNum = 0, ret = 0;
    memset(pk_1, 0, sizeof(pk_1)); memset(gk_1, 0, sizeof(gk_1)); memset(xk_1, 0, sizeof(xk_1));
    memcpy(xk, theta0, sizeof(xk));
    se->n_eval = 0;
    /* Guard against a line search that never settles (a non-finite bracket makes every exit test
     * of the zoom phase false): the whole fit may spend at most 64*max_iter+1024 evaluations.
     * Reported as CN_EVAL_LIMIT (< 0, i.e. "optimiser failed": the reference drops the series). */
    se->eval_limit = 64 * o->max_iter + 1024;
    if (se->gram ? cn_resid_q(se, xk, &fk, gk) : cn_eval(se, xk, &fk, gk)) {
        memcpy(theta_out, theta0, sizeof(xk));
        res->status = CN_INIT_NONFINITE; res->n_iter = 0; res->n_eval = se->n_eval; res->f = fk;
        return 0;
    }
    if (se->gram) cn_set_ref(se, xk);
    for (int i = 0; i < NP; ++i) pk[i] = -gk[i];
    while (ret == 0) {
        int resetB;
        itNum++;
        resetB = (itNum == 1) ? 1 : 0;
        for (;;) {
            if (resetB) for (int i = 0; i < NP; ++i) pk[i] = -gk[i];
            if (itNum > 1 && resetB != 2) {
                const double ci = cubic_interp6(dotc(gk_1, pk_1), alpha, fk - fk_1, dotc(gk, pk),
                                                minAlpha, 1.0);
                alpha = fmin(1.0, 1.01 * ci);
            } else {
                alpha = o->init_alpha;
            }
            const int rc = line_search(se, &alpha, xk_1, &fk_1, gk_1, pk, xk, fk, gk);
            if (rc == 2) { ret = CN_EVAL_LIMIT; goto done; }
            if (rc) {
                if (resetB) { ret = TERM_LSFAIL; goto done; }
                resetB = 2;
                continue;
            }
            break;
        }
        /* swap: k becomes the most recent iterate */
        { double tf = fk; fk = fk_1; fk_1 = tf; }
        for (int i = 0; i < NP; ++i) {
            double tx = xk[i]; xk[i] = xk_1[i]; xk_1[i] = tx;
            double tg = gk[i]; gk[i] = gk_1[i]; gk_1[i] = tg;
            double tp = pk[i]; pk[i] = pk_1[i]; pk_1[i] = tp;
        }
        if (se->gram) {
            /* re-centring rule: the accepted iterate is re-evaluated in residual form and
             * becomes the reference when the quadratic term has grown against s0, or after
             * recenter_every accepted iterations */
            se->since_rc++;
            if (se->last_q2 > o->recenter_ratio * se->s0 || se->since_rc >= o->recenter_every) {
                double fr, gr[CN_MAX_P];
                if (cn_resid_q(se, xk, &fr, gr) == 0) {
                    fk = fr; memcpy(gk, gr, sizeof(gr)); cn_set_ref(se, xk);
                }
            }
        }
        for (int i = 0; i < NP; ++i) { sk[i] = xk[i] - xk_1[i]; yk[i] = gk[i] - gk_1[i]; }
        const double gradNorm = sqrt(dotc(gk, gk));
        const double stepNorm = sqrt(dotc(sk, sk));
        const double skyk = dotc(yk, sk);
        const double ykyk = dotc(yk, yk);
        if (resetB) {
            const double B0fact = ykyk / skyk;
            hist_len = 0; hist_head = 0;
            for (int i = 0; i < NP; ++i) pk_1[i] = pk_1[i] / B0fact;
            alpha = alpha * B0fact;
        }
        gammak = skyk / ykyk;
        {
            int slot;
            if (hist_len < H) { slot = (hist_head + hist_len) % H; hist_len++; }
            else { slot = hist_head; hist_head = (hist_head + 1) % H; }
            rho[slot] = 1.0 / skyk;
            memcpy(Sb[slot], sk, sizeof(sk));
            memcpy(Yb[slot], yk, sizeof(yk));
        }
        for (int i = 0; i < NP; ++i) pk[i] = -gk[i];
        for (int h = hist_len - 1; h >= 0; --h) {
            const int slot = (hist_head + h) % H;
            const double a = rho[slot] * dotc(Sb[slot], pk);
            for (int i = 0; i < NP; ++i) pk[i] = fma(-a, Yb[slot][i], pk[i]);
            alphas[h] = a;
        }
        for (int i = 0; i < NP; ++i) pk[i] = pk[i] * gammak;
        for (int h = 0; h < hist_len; ++h) {
            const int slot = (hist_head + h) % H;
            const double b = rho[slot] * dotc(Yb[slot], pk);
            const double cc = alphas[h] - b;
            for (int i = 0; i < NP; ++i) pk[i] = fma(cc, Sb[slot][i], pk[i]);
        }
        const double dF = fabs(fk_1 - fk);
        const double fmaxv = fmax(fabs(fk_1), fmax(fabs(fk), 1.0));
        if (dF < o->tol_obj) ret = TERM_ABSF;
        else if (dF < o->tol_rel_obj * eps * fmaxv) ret = TERM_RELF;
        else if (gradNorm < o->tol_grad) ret = TERM_ABSGRAD;
        else if (-dotc(gk, pk) / fmax(fabs(fk), 1.0) < o->tol_rel_grad * eps) ret = TERM_RELGRAD;
        else if (stepNorm < o->tol_param) ret = TERM_ABSX;
        else if (itNum >= o->max_iter) ret = TERM_MAXIT;
        else ret = TERM_SUCCESS;
    }
done:
    memcpy(theta_out, xk, sizeof(xk));
    res->status = ret; res->n_iter = itNum; res->n_eval = se->n_eval; res->f = fk;
    return 0;
}

/* ---- Stan's Newton optimiser ------------------------------------------------------------
 * fbprophet 0.5 calls optimizing(algorithm='Newton') when T < 100 and as the retry after an
 * L-BFGS RuntimeError (forecaster.py fit).  Restated from stan 2.19
 * (UPSTREAM-RECALL, not in /root/reference):
 *   model/grad_hess_log_prob.hpp   Hessian by finite differences of the gradient: epsilon 1e-3,
 *       perturbations {-2e,-e,+e,+2e} of one coordinate at a time, coefficients
 *       {1/12,-2/3,2/3,-1/12}, increment = half_epsilon * coefficient * gradient added to
 *       H[d][dd] and H[dd][d]  (half_epsilon = 0.5 * epsilon MULTIPLIES in the recalled source,
 *       where the finite-difference formula divides: the Hessian comes out 1e-6 of its value,
 *       the Newton step 1e6 too long and the step-halving loop absorbs it -- kept as recalled,
 *       see tests/dev/newton_vs_lbfgs.py for what the other spelling changes);
 *   optimization/newton.hpp        make_negative_definite_and_solve (eigen-decomposition,
 *       eigenvalues replaced by -|lambda|), newton_step (step halving from 1 until
 *       f1 >= f0, give up below 1e-50);
 *   services/optimize/newton.hpp   iterate until |lp - lastlp| < 1e-8 or num_iterations.
 * NOT pinned against Stan output (parity unpinned); it exists so that a GPU Newton has a
 * checker.  Canonical order (lane = parameter p < 64):
 *   A[d][p]   = fma chain over the 4 perturbations of coordinate d, H[a][b] = A[a][b] + A[b][a];
 *   eigen-decomposition by Householder tridiagonalisation + implicit QL (cn_tridiag_ql; the
 *   round-robin Jacobi of round 1, cn_jacobi, is kept as an independent cross-check);
 *   proj[j]   = fma chain over i of V[i][j] * g[i];  proj[j] = -proj[j] / |lambda_j|;
 *   step[i]   = fma chain over j of V[i][j] * proj[j];  new[i] = th[i] - size * step[i]. */

/* Newton up to 128 parameters (round 4: the two-parameters-per-lane kernel newton_kernel2, so that fbprophet's
 * retry-with-Newton and its T < 100 rule exist for models of more than 64 parameters and for mixed additive /
 * multiplicative columns too).  Sums over more than 64 entries follow the rule of dotc: entry j in slot j % 64, the
 * j >= 64 term fma'd onto the j - 64 term, then the 64-slot butterfly.  The round-robin Jacobi cross-check keeps
 * its 64 (CN_JACOBI_MAX_P). */
#define CN_NEWTON_MAX_P 128
#define CN_JACOBI_MAX_P 64
enum { TERM_NEWTON_CONVERGED = 60, CN_NEWTON_FAIL = -4, CN_NEWTON_TOO_WIDE = -13 };

/* Symmetric eigen-decomposition, parallel-order Jacobi.  Every round applies n/2 rotations on
 * disjoint index pairs at once (angles all taken from the matrix before the round), pairs from
 * the round-robin tournament: in round r index m-1 meets r and every other i meets
 * (2r - i) mod (m-1), m = n rounded up to even (an odd n plays against a dummy = no rotation).
 * A: [n][n] row-major (destroyed), V: eigenvectors in columns, lam: eigenvalues (unsorted). */
static int cn_jacobi(int n, double *A, double *V, double *lam)
{
    const int m = n + (n & 1);
    double B[CN_JACOBI_MAX_P * CN_JACOBI_MAX_P], W[CN_JACOBI_MAX_P * CN_JACOBI_MAX_P];
    double c[CN_JACOBI_MAX_P], kap[CN_JACOBI_MAX_P];
    int par[CN_JACOBI_MAX_P];
    if (n > CN_JACOBI_MAX_P) return -1;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    int sweep = 0;
    for (; sweep < 30; ++sweep) {
        /* relative off-diagonal weight, columns summed first, then the 64-slot butterfly */
        double offp[CN_W], diap[CN_W];
        for (int j = 0; j < CN_W; ++j) {
            double so = 0.0, sd = 0.0;
            if (j < n)
                for (int i = 0; i < n; ++i) {
                    const double a = A[i * n + j];
                    if (i == j) sd = a * a; else so = fma(a, a, so);
                }
            offp[j] = so; diap[j] = sd;
        }
        const double off2 = bfly(offp), dia2 = bfly(diap);
        if (off2 <= 1e-26 * dia2) break;
        for (int r = 0; r < m - 1; ++r) {
            for (int i = 0; i < n; ++i) {
                int q;
                if (i == m - 1) q = r;
                else if (i == r) q = m - 1;
                else { q = (2 * r - i) % (m - 1); if (q < 0) q += m - 1; }
                par[i] = q;
                if (q >= n) { c[i] = 1.0; kap[i] = 0.0; continue; }
                const int lo = i < q ? i : q, hi = i < q ? q : i;
                const double apq = A[lo * n + hi];
                if (apq == 0.0) { c[i] = 1.0; kap[i] = 0.0; continue; }
                const double tau = (A[hi * n + hi] - A[lo * n + lo]) / (2.0 * apq);
                const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double cc = 1.0 / sqrt(1.0 + t * t);
                const double ss = t * cc;
                c[i] = cc;
                kap[i] = (i == lo) ? -ss : ss;
            }
            /* B = A J, W = V J (column i mixes with column par[i]); A = J^T B */
            for (int rr = 0; rr < n; ++rr)
                for (int i = 0; i < n; ++i) {
                    const int q = par[i];
                    const double aq = q < n ? A[rr * n + q] : 0.0, vq = q < n ? V[rr * n + q] : 0.0;
                    B[rr * n + i] = fma(aq, kap[i], A[rr * n + i] * c[i]);
                    W[rr * n + i] = fma(vq, kap[i], V[rr * n + i] * c[i]);
                }
            for (int i = 0; i < n; ++i) {
                const int q = par[i];
                for (int j = 0; j < n; ++j) {
                    const double bq = q < n ? B[q * n + j] : 0.0;
                    A[i * n + j] = fma(kap[i], bq, c[i] * B[i * n + j]);
                }
            }
            memcpy(V, W, sizeof(double) * (size_t)(n * n));
        }
    }
    for (int i = 0; i < n; ++i) lam[i] = A[i * n + i];
    return sweep;
}

/* Symmetric eigen-decomposition by Householder tridiagonalisation + implicit QL with shifts (the
 * method behind Eigen's SelfAdjointEigenSolver that Stan's newton_step calls, restated in the
 * classical tred2 / tql2 form) in an operation order that one wavefront can follow: every "for all
 * j" below is one lane per j, sums over lanes are the 64-slot butterfly, sums over k inside a lane
 * are sequential fma chains, scalars are computed identically by every lane.
 * A: [n][n] row-major, full symmetric (destroyed); V: eigenvectors in columns; lam: eigenvalues
 * (in the order the QL iteration leaves them).  Returns the number of QL iterations. */
static double cn_pythag(double a, double b)
{
    const double absa = fabs(a), absb = fabs(b);
    if (absa > absb) { const double r = absb / absa; return absa * sqrt(1.0 + r * r); }
    if (absb == 0.0) return 0.0;
    { const double r = absa / absb; return absb * sqrt(1.0 + r * r); }
}

static int cn_tridiag_ql(int n, double *A, double *V, double *lam)
{
    double d[CN_NEWTON_MAX_P], e[CN_NEWTON_MAX_P], hh[CN_NEWTON_MAX_P];
    double u[CN_NEWTON_MAX_P], pv[CN_NEWTON_MAX_P], qv[CN_NEWTON_MAX_P], part[CN_W];
    for (int i = 0; i < n; ++i) { e[i] = 0.0; hh[i] = 0.0; }
    /* ---- Householder: zero A[i][0..i-2] for i = n-1 .. 2; reflector u kept in row i, u.u/2 in hh[i] */
    for (int i = n - 1; i >= 2; --i) {
        const int l = i - 1;
        for (int j = 0; j < CN_W; ++j) part[j] = (j < l) ? A[i * n + j] * A[i * n + j] : 0.0;
        for (int j = CN_W; j < l; ++j) part[j - CN_W] = fma(A[i * n + j], A[i * n + j], part[j - CN_W]);
        const double sigma = bfly(part);
        const double alpha = A[i * n + l];
        if (sigma == 0.0) { e[i] = alpha; hh[i] = 0.0; continue; }
        const double mu = sqrt(sigma + alpha * alpha);
        const double beta = (alpha >= 0.0) ? -mu : mu;
        for (int k = 0; k <= l; ++k) u[k] = A[i * n + k];
        u[l] = alpha - beta;
        const double H = 0.5 * (sigma + u[l] * u[l]);
        for (int j = 0; j <= l; ++j) {                      /* p = A u / H */
            double a = 0.0;
            for (int k = 0; k <= l; ++k) a = fma(A[j * n + k], u[k], a);
            pv[j] = a / H;
        }
        for (int j = 0; j < CN_W; ++j) part[j] = (j <= l) ? u[j] * pv[j] : 0.0;
        for (int j = CN_W; j <= l; ++j) part[j - CN_W] = fma(u[j], pv[j], part[j - CN_W]);
        const double K = bfly(part) / (2.0 * H);
        for (int j = 0; j <= l; ++j) qv[j] = pv[j] - K * u[j];
        for (int j = 0; j <= l; ++j)                        /* A <- A - u q^T - q u^T */
            for (int k = 0; k <= l; ++k)
                A[j * n + k] = fma(-qv[j], u[k], fma(-u[j], qv[k], A[j * n + k]));
        for (int k = 0; k <= l; ++k) A[i * n + k] = u[k];
        e[i] = beta; hh[i] = H;
    }
    if (n > 1) e[1] = A[1 * n + 0];
    for (int i = 0; i < n; ++i) d[i] = A[i * n + i];
    /* ---- Q = H_{n-1} ... H_2: start from the identity, apply H_2, H_3, ... from the left */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int i = 2; i < n; ++i) {
        if (hh[i] == 0.0) continue;
        const int l = i - 1;
        for (int c = 0; c <= l; ++c) {                      /* lane = column c */
            double w = 0.0;
            for (int k = 0; k <= l; ++k) w = fma(A[i * n + k], V[k * n + c], w);
            w = w / hh[i];
            for (int r = 0; r <= l; ++r) V[r * n + c] = fma(-A[i * n + r], w, V[r * n + c]);
        }
    }
    /* ---- implicit QL on (d, e), rotations applied to the columns of V (lane = row) */
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    int iters = 0;
    for (int l = 0; l < n; ++l) {
        for (int guard = 0; guard < 60; ++guard) {
            int m = l;
            for (; m < n - 1; ++m) {
                const double dd = fabs(d[m]) + fabs(d[m + 1]);
                if (fabs(e[m]) + dd == dd) break;
            }
            if (m == l) break;
            ++iters;
            double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
            double r = cn_pythag(g, 1.0);
            g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
            double s = 1.0, c = 1.0, p = 0.0;
            int i = m - 1, underflow = 0;
            for (; i >= l; --i) {
                double f = s * e[i];
                const double b = c * e[i];
                r = sqrt(fma(f, f, g * g));     /* entries of a Hessian: nowhere near over/underflow */
                e[i + 1] = r;
                if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; underflow = 1; break; }
                { const double ri = 1.0 / r; s = f * ri; c = g * ri; }
                g = d[i + 1] - p;
                r = (d[i] - g) * s + 2.0 * c * b;
                p = s * r;
                d[i + 1] = g + p;
                g = c * r - b;
                for (int k = 0; k < n; ++k) {               /* lane = row k */
                    f = V[k * n + i + 1];
                    V[k * n + i + 1] = fma(s, V[k * n + i], c * f);
                    V[k * n + i] = fma(c, V[k * n + i], -(s * f));
                }
            }
            if (underflow) continue;
            d[l] -= p; e[l] = g; e[m] = 0.0;
        }
    }
    for (int i = 0; i < n; ++i) lam[i] = d[i];
    return iters;
}

/* make_negative_definite_and_solve where the finite-difference Hessian already IS negative definite (round 6; a third of
 * the iterations on BASELINE cfg5): then |H| = -H and the step V |Lambda|^-1 V^T g is the solution of (-H) step = g -- by
 * Cholesky, -H = L L^T, instead of an eigen-decomposition (69 % of a Newton fit on the GPU).  Same step up to rounding
 * (~1e-12 relative: Newton is not chaotic).  Canonical order, lane = row i:
 *   column j = 0 .. P-1:  s_i = -H[i][j] - sum_{k<j} L[i][k] L[j][k]   (i >= j; FOUR fma chains over k, chain k mod 4, the
 *                         first one starting from -H[i][j], combined (a0 + a1) + (a2 + a3) -- one chain of 33 dependent
 *                         LDS-fed steps per column was half of the shortcut's cost on the GPU; H read from row j of the
 *                         symmetric matrix);  the pivot s_j must be > 0, else H is not negative definite: return 0
 *                         and the eigen route runs on the untouched H;  L[j][j] = sqrt(s_j), L[i][j] = s_i / L[j][j];
 *   forward  r = g:       z_j = r_j / L[j][j], then r_i = fma(-L[i][j], z_j, r_i) for i > j;
 *   backward r = z:       step_j = r_j / L[j][j] (j = P-1 .. 0), then r_i = fma(-L[j][i], step_j, r_i) for i < j.
 * Only where the kernels hold one parameter per lane -- P <= 64 and one column mode (cn_newton checks): the
 * two-parameters-per-lane kernel (wider or mixed-mode models, a handful of retried series) keeps the eigen route. */
static int cn_chol_neg_solve(int P, const double *H, const double *g, double *step)
{
    double L[CN_W * CN_W], s[CN_W], r[CN_W], z[CN_W];
    if (P > CN_W) return 0;
    for (int j = 0; j < P; ++j) {
        for (int i = j; i < P; ++i) {
            double a[4] = {-H[j * P + i], 0.0, 0.0, 0.0};
            for (int k = 0; k < j; ++k) a[k & 3] = fma(-L[i * CN_W + k], L[j * CN_W + k], a[k & 3]);
            s[i] = (a[0] + a[1]) + (a[2] + a[3]);
        }
        if (!(s[j] > 0.0)) return 0;
        const double ljj = sqrt(s[j]);
        L[j * CN_W + j] = ljj;
        for (int i = j + 1; i < P; ++i) L[i * CN_W + j] = s[i] / ljj;
    }
    for (int i = 0; i < P; ++i) { r[i] = g[i]; z[i] = 0.0; }
    for (int j = 0; j < P; ++j) {
        const double zj = r[j] / L[j * CN_W + j];
        z[j] = zj;
        for (int i = j + 1; i < P; ++i) r[i] = fma(-L[i * CN_W + j], zj, r[i]);
    }
    for (int i = 0; i < P; ++i) r[i] = z[i];
    for (int j = P - 1; j >= 0; --j) {
        const double sj = r[j] / L[j * CN_W + j];
        step[j] = sj;
        for (int i = 0; i < j; ++i) r[i] = fma(-L[j * CN_W + i], sj, r[i]);
    }
    return 1;
}

static int cn_newton(cn_series *se, const cn_spec *o, const double *theta0, double *theta_out,
                     cn_fitinfo *res)
{
    const int P = 3 + se->S + se->K;
    const double epsilon = 1e-3, half_epsilon = 0.5 * epsilon;
    const double pert[4] = {-2 * epsilon, -1 * epsilon, epsilon, 2 * epsilon};
    const double coef[4] = {1.0 / 12.0, -2.0 / 3.0, 2.0 / 3.0, -1.0 / 12.0};
    double th[CN_MAX_P], x[CN_MAX_P], g[CN_MAX_P], tg[CN_MAX_P], step[CN_MAX_P];
    double A[CN_NEWTON_MAX_P * CN_NEWTON_MAX_P], H[CN_NEWTON_MAX_P * CN_NEWTON_MAX_P];
    double V[CN_NEWTON_MAX_P * CN_NEWTON_MAX_P], lam[CN_NEWTON_MAX_P], proj[CN_NEWTON_MAX_P];
    double f, lp, lastlp;
    int it = 0, ret = TERM_MAXIT;
    memcpy(th, theta0, sizeof(th));
    se->n_eval = 0;
    if (P > CN_NEWTON_MAX_P) {
        memcpy(theta_out, theta0, sizeof(th));
        res->status = CN_NEWTON_TOO_WIDE; res->n_iter = 0; res->n_eval = 0; res->f = 0.0;
        return 0;
    }
    /* Evaluation form (cn_spec.eval_mode, as for L-BFGS): models that are linear in (k, m, delta,
     * beta) evaluate the accepted point of every Newton iteration in residual form -- which also
     * re-centres the quadratic form there -- and the 4 P finite-difference points and the halving
     * trials of that iteration in quadratic (Gram) form around it. */
#define CN_NEWTON_EVAL_RES(th_, f_, g_) (se->gram ? cn_resid_q(se, th_, f_, g_) : cn_eval(se, th_, f_, g_))
#define CN_NEWTON_EVAL(th_, f_, g_) (se->gram ? cn_eval_gram(se, th_, f_, g_) : cn_eval(se, th_, f_, g_))
    if (CN_NEWTON_EVAL_RES(th, &f, g)) {
        /* services/optimize/newton.hpp carries on with lp = -inf and the first
         * grad_hess_log_prob throws: pystan raises RuntimeError */
        memcpy(theta_out, theta0, sizeof(th));
        res->status = CN_INIT_NONFINITE; res->n_iter = 0; res->n_eval = se->n_eval; res->f = f;
        return 0;
    }
    lp = -f;
    for (int mI = 0; mI < o->max_iter; ++mI) {
        lastlp = lp;
        /* ---- newton_step: grad_hess_log_prob ---- */
        double f0;
        if (CN_NEWTON_EVAL_RES(th, &f, g)) { ret = CN_NEWTON_FAIL; break; }
        if (se->gram) cn_set_ref(se, th);
        f0 = -f;
        int bad = 0;
        for (int d = 0; d < P && !bad; ++d) {
            double acc[CN_NEWTON_MAX_P];
            for (int p = 0; p < P; ++p) acc[p] = 0.0;
            for (int i = 0; i < 4; ++i) {
                memcpy(x, th, sizeof(x));
                x[d] = th[d] + pert[i];
                double fp;
                if (CN_NEWTON_EVAL(x, &fp, tg)) { bad = 1; break; }   /* Stan: exception leaves newton_step */
                const double w = half_epsilon * coef[i];
                for (int p = 0; p < P; ++p) acc[p] = fma(w, -tg[p], acc[p]);
            }
            for (int p = 0; p < P; ++p) A[d * P + p] = acc[p];
        }
        if (bad) { ret = CN_NEWTON_FAIL; break; }
        for (int a = 0; a < P; ++a)
            for (int b = 0; b < P; ++b) H[a * P + b] = A[a * P + b] + A[b * P + a];
        /* ---- make_negative_definite_and_solve (gradient of lp = -g) ---- */
        memset(step, 0, sizeof(step));
        const int one_per_lane = P <= CN_W && (se->Ka == 0 || se->Ka == se->K);
        if (!(one_per_lane && cn_chol_neg_solve(P, H, g, step))) {
            cn_tridiag_ql(P, H, V, lam);
            for (int j = 0; j < P; ++j) {
                double a = 0.0;
                for (int i = 0; i < P; ++i) a = fma(V[i * P + j], -g[i], a);
                proj[j] = -a / fabs(lam[j]);
            }
            for (int i = 0; i < P; ++i) {
                double a = 0.0;
                for (int j = 0; j < P; ++j) a = fma(V[i * P + j], proj[j], a);
                step[i] = a;
            }
        }
        /* ---- step halving ----
         * Round 6, quadratic form: the ~30 trials of an iteration (the recalled epsilon / 2 factor makes the step 1e6 too
         * long) all lie on ONE line through the point the quadratic form was re-centred at a moment ago, x = th - size * step,
         * so D = x - ref = -size * step and
         *     SSE(size) = s0 - 2 c.D + D.(M D) = s0 + 2 size (c . step) + size^2 (step . M step):
         * two dot products and one mat-vec per ITERATION (sw, cs below), three scalars per TRIAL, where every trial used to
         * be a P x P mat-vec (19 % of a fit on the GPU).  The value is cn_assemble_q's on that SSE (gradient not formed:
         * the next iteration starts with a residual-form evaluation of the accepted point).  Same function, other rounding
         * than cn_eval_gram at the same x. */
        double sw = 0.0, cs = 0.0;
        if (se->gram) {
            double Dl[CN_MAX_P], vl[CN_MAX_P];
            for (int p = 0; p < CN_MAX_P; ++p) { Dl[p] = 0.0; vl[p] = 0.0; }
            for (int p = 0; p < P; ++p) Dl[p] = (p == 2) ? 0.0 : step[p];
            for (int p = 0; p < P; ++p) {
                double a[4] = {0.0, 0.0, 0.0, 0.0};
                const double *row = se->M + (size_t)p * P;
                for (int q = 0; q < P; ++q) a[q & 3] = fma(row[q], Dl[q], a[q & 3]);
                vl[p] = (a[0] + a[1]) + (a[2] + a[3]);
            }
            sw = dotc(Dl, vl);
            cs = dotc(se->cvec, Dl);
        }
        double size = 2.0, f1 = -1e100;
        int moved = 1;
        memcpy(x, th, sizeof(x));
        while (f1 < f0) {
            size *= 0.5;
            if (size < 1e-50) { moved = 0; break; }
            for (int i = 0; i < P; ++i) x[i] = th[i] - size * step[i];
            double fn;
            if (se->gram) {
                double zero[CN_MAX_P];
                for (int p = 0; p < CN_MAX_P; ++p) zero[p] = 0.0;
                se->n_eval++;
                const double q2 = (size * size) * sw;
                const double cd = -(size * cs);
                const double sse = fma(-2.0, cd, se->s0) + q2;
                f1 = cn_assemble_q(se, x, sse, zero, &fn, tg) ? -1e100 : -fn;
            } else {
                f1 = CN_NEWTON_EVAL(x, &fn, tg) ? -1e100 : -fn;
            }
        }
        it++;
        if (moved) { memcpy(th, x, sizeof(th)); lp = f1; }
        else lp = f0;
        /* the first comparison in Stan is against the initial lp computed WITH the constant
         * terms (log_prob<false,false>) and therefore never fires */
        if (mI > 0 && fabs(lp - lastlp) < 1e-8) { ret = TERM_NEWTON_CONVERGED; break; }
    }
    memcpy(theta_out, th, sizeof(th));
    res->status = ret; res->n_iter = it; res->n_eval = se->n_eval; res->f = -lp;
    return 0;
}

/* ---- exported entry points (theta in ORIGINAL column order) ----------------------------- */

/* Caller layout: [k, m, log sigma, delta[S_out], beta[K]] -- the dummy changepoint of a series
 * without changepoints (S = 1, S_out = 0) has no slot there. */
static void to_internal(const cn_series *se, const double *th_orig, double *th_int)
{
    memset(th_int, 0, sizeof(double) * CN_MAX_P);
    for (int p = 0; p < 3 + se->S_out; ++p) th_int[p] = th_orig[p];
    for (int j = 0; j < se->K; ++j) th_int[3 + se->S + j] = th_orig[3 + se->S_out + se->perm[j]];
}

/* fold: fitted parameters -- Prophet.fit's `if len(self.changepoints) == 0: k = k + delta;
 * delta = 0`; gradients are returned without their dummy-delta entry instead (fold = 0). */
static void to_original(const cn_series *se, const double *th_int, double *th_orig, int fold)
{
    for (int p = 0; p < 3 + se->S_out; ++p) th_orig[p] = th_int[p];
    if (fold && se->S_out == 0) th_orig[0] = th_int[0] + th_int[3];
    for (int j = 0; j < se->K; ++j) th_orig[3 + se->S_out + se->perm[j]] = th_int[3 + se->S + j];
}

static void fill_info(const cn_series *se, cn_fitinfo *info)
{
    info->S = se->S_out; info->K = se->K; info->y_scale = se->y_scale; info->floor_ = se->floor_;
    info->cap_scaled = se->cap; info->start_ns = se->start_ns; info->t_scale_ns = se->tscale_ns;
}

void cn_default_spec(cn_spec *sp)
{
    memset(sp, 0, sizeof(*sp));
    sp->growth = 0; sp->n_changepoints = 25; sp->changepoint_range = 0.8; sp->tau = 0.05;
    sp->max_iter = 10000; sp->history = 5; sp->init_alpha = 1e-3; sp->tol_obj = 1e-12;
    sp->tol_rel_obj = 1e4; sp->tol_grad = 1e-8; sp->tol_rel_grad = 1e7; sp->tol_param = 1e-8;
    sp->eval_mode = 0; sp->recenter_every = 128; sp->recenter_ratio = 1.0;
}

int cn_spec_size(void) { return (int)sizeof(cn_spec); }

/* Design matrix in original column order [T][K], scaled t [T], scaled y [T], t_change [S],
 * init (k, m).  Any output pointer may be NULL. */
int cn_design(const cn_spec *sp, int T, const int64_t *ds, const double *y, double floor_,
              double cap, const double *extra, double *X_out, double *t_out, double *y_out,
              double *tchange_out, double *init_out, cn_fitinfo *info)
{
    int err;
    cn_series *se = cn_prepare(sp, T, ds, y, floor_, cap, extra, &err);
    if (!se) { info->status = err; return err; }
    fill_info(se, info);
    info->status = 0;
    if (X_out)
        for (int i = 0; i < T; ++i)
            for (int j = 0; j < se->K; ++j)
                X_out[(size_t)i * se->K + se->perm[j]] = se->X[(size_t)i * se->K + j];
    if (t_out) memcpy(t_out, se->t, sizeof(double) * T);
    if (y_out) memcpy(y_out, se->y, sizeof(double) * T);
    if (tchange_out) memcpy(tchange_out, se->t_change, sizeof(double) * se->S_out);
    if (init_out) { init_out[0] = se->k0; init_out[1] = se->m0; }
    free_series(se);
    return 0;
}

/* f = -log_prob and gradient at theta (original order). */
int cn_eval_at(const cn_spec *sp, int T, const int64_t *ds, const double *y, double floor_,
               double cap, const double *extra, const double *theta, double *f_out,
               double *g_out)
{
    int err;
    cn_series *se = cn_prepare(sp, T, ds, y, floor_, cap, extra, &err);
    if (!se) return err;
    double th[CN_MAX_P], g[CN_MAX_P];
    to_internal(se, theta, th);
    const int rc = cn_eval(se, th, f_out, g);
    to_original(se, g, g_out, 0);
    free_series(se);
    return rc;
}

/* Quadratic-form evaluation (cn_eval_gram) at theta around the reference point theta_ref: a residual-form
 * pass at theta_ref (cn_resid_q) makes it the reference, then ONE evaluation of the quadratic form -- what
 * every trial point of a line search costs under eval_mode 1.  Counterpart of the product's
 * tsf_eval_quadratic (tests compare bits).  Returns 0, 1 (non-finite) or < 0 (model not linear/additive). */
int cn_eval_quadratic_at(const cn_spec *sp, int T, const int64_t *ds, const double *y, const double *extra,
                         const double *theta_ref, const double *theta, double *f_out, double *g_out)
{
    int err;
    cn_series *se = cn_prepare(sp, T, ds, y, 0.0, 0.0, extra, &err);
    if (!se) return err;
    if (!(sp->growth == 0 && se->Ka == se->K)) { free_series(se); return -100; }
    se->gram = 1; cn_build_gram(se);
    double thr[CN_MAX_P], th[CN_MAX_P], g[CN_MAX_P], fr;
    to_internal(se, theta_ref, thr);
    to_internal(se, theta, th);
    int rc = cn_resid_q(se, thr, &fr, g);
    cn_set_ref(se, thr);
    rc |= cn_eval_gram(se, th, f_out, g);
    to_original(se, g, g_out, 0);
    free_series(se);
    return rc;
}

/* Full fit.  theta_out: [3+S+K] original order; tchange_out: [S]. */
int cn_fit(const cn_spec *sp, int T, const int64_t *ds, const double *y, double floor_,
           double cap, const double *extra, double *theta_out, double *tchange_out,
           cn_fitinfo *info)
{
    int err;
    memset(info, 0, sizeof(*info));
    cn_series *se = cn_prepare(sp, T, ds, y, floor_, cap, extra, &err);
    if (!se) { info->status = err; return 0; }
    fill_info(se, info);
    double th0[CN_MAX_P], th[CN_MAX_P];
    memset(th0, 0, sizeof(th0));
    th0[0] = se->k0; th0[1] = se->m0; th0[2] = 0.0;
    if (se->constant_y) {
        /* fbprophet: "Nothing to fit": params = init, sigma_obs = 1e-9 */
        memcpy(th, th0, sizeof(th));
        th[2] = -20.72326583694641;
        info->status = CN_CONSTANT; info->n_iter = 0; info->n_eval = 0; info->f = 0.0;
    } else {
        if (sp->eval_mode == 1 && sp->growth == 0 && se->Ka == se->K) { se->gram = 1; cn_build_gram(se); }
        cn_lbfgs(se, sp, th0, th, info);
        info->pad_ = se->n_resid;       /* residual-form evaluations (quadratic form only) */
    }
    to_original(se, th, theta_out, 1);
    if (tchange_out) memcpy(tchange_out, se->t_change, sizeof(double) * se->S_out);
    free_series(se);
    return 0;
}

/* Full fit with Stan's Newton optimiser (cn_newton).  Same outputs as cn_fit. */
int cn_fit_newton(const cn_spec *sp, int T, const int64_t *ds, const double *y, double floor_,
                  double cap, const double *extra, double *theta_out, double *tchange_out,
                  cn_fitinfo *info)
{
    int err;
    memset(info, 0, sizeof(*info));
    cn_series *se = cn_prepare(sp, T, ds, y, floor_, cap, extra, &err);
    if (!se) { info->status = err; return 0; }
    fill_info(se, info);
    double th0[CN_MAX_P], th[CN_MAX_P];
    memset(th0, 0, sizeof(th0));
    th0[0] = se->k0; th0[1] = se->m0; th0[2] = 0.0;
    if (se->constant_y) {
        memcpy(th, th0, sizeof(th));
        th[2] = -20.72326583694641;
        info->status = CN_CONSTANT; info->n_iter = 0; info->n_eval = 0; info->f = 0.0;
    } else {
        if (sp->eval_mode == 1 && sp->growth == 0 && se->Ka == se->K) { se->gram = 1; cn_build_gram(se); }
        cn_newton(se, sp, th0, th, info);
    }
    to_original(se, th, theta_out, 1);
    if (tchange_out) memcpy(tchange_out, se->t_change, sizeof(double) * se->S_out);
    free_series(se);
    return 0;
}

/* Eigen-decomposition hook for tests: A [n][n] symmetric (copied), V [n][n], lam [n]. */
int cn_jacobi_eigh(int n, const double *A_in, double *V, double *lam)
{
    double A[CN_NEWTON_MAX_P * CN_NEWTON_MAX_P];
    if (n < 1 || n > CN_NEWTON_MAX_P) return -1;
    memcpy(A, A_in, sizeof(double) * (size_t)(n * n));
    return cn_jacobi(n, A, V, lam);
}

/* The eigen-solver cn_newton uses (tridiagonalisation + implicit QL), for tests. */
int cn_ql_eigh(int n, const double *A_in, double *V, double *lam)
{
    double A[CN_NEWTON_MAX_P * CN_NEWTON_MAX_P];
    if (n < 1 || n > CN_NEWTON_MAX_P) return -1;
    memcpy(A, A_in, sizeof(double) * (size_t)(n * n));
    return cn_tridiag_ql(n, A, V, lam);
}

/* Point forecast.  theta original order; extra_future [n_extra][H]. */
int cn_predict(const cn_spec *sp, const cn_fitinfo *info, const double *theta,
               const double *t_change, int H, const int64_t *ds, double floor_, double cap,
               const double *extra_future, double *yhat, double *trend_out)
{
    const int S = info->S, K = info->K;
    const double k = theta[0], m = theta[1];
    const double *delta = theta + 3, *beta = theta + 3 + S;
    const double fl = (sp->growth == 1) ? floor_ : 0.0;
    const double cap_sc = (sp->growth == 1) ? (cap - fl) / info->y_scale : 0.0;
    double ks[CN_MAX_S + 1], mc[CN_MAX_S + 1];
    int perm[CN_MAX_P], Ka;
    double prior[CN_MAX_P];
    build_perm(sp, K, perm, prior, &Ka);
    ks[0] = k; mc[0] = m;
    for (int j = 0; j < S; ++j) ks[j + 1] = ks[j] + delta[j];
    if (sp->growth == 0) {
        for (int j = 0; j < S; ++j) mc[j + 1] = mc[j] + ((-t_change[j]) * delta[j]);
    } else {
        for (int j = 0; j < S; ++j) {
            const double gamma = (t_change[j] - mc[j]) * (1.0 - ks[j] / ks[j + 1]);
            mc[j + 1] = mc[j] + gamma;
        }
    }
    const double tsc = (double)info->t_scale_ns;
    const int nf = K - sp->n_extra;
    for (int h = 0; h < H; ++h) {
        const double t = (double)(ds[h] - info->start_ns) / tsc;
        int c = 0;
        while (c < S && t >= t_change[c]) ++c;
        double row[CN_MAX_P];
        fourier_row(sp, ds[h], row);
        for (int e = 0; e < sp->n_extra; ++e) row[nf + e] = extra_future[(size_t)e * H + h];
        double xa = 0.0, xm = 0.0;
        for (int j = 0; j < Ka; ++j) xa = fma(row[perm[j]], beta[perm[j]], xa);
        for (int j = Ka; j < K; ++j) xm = fma(row[perm[j]], beta[perm[j]], xm);
        double gtr;
        if (sp->growth == 0) {
            gtr = fma(ks[c], t, mc[c]);
        } else {
            const double z = ks[c] * (t - mc[c]);
            gtr = cap_sc * (1.0 / (1.0 + det_exp(-z)));
        }
        const double trend = gtr * info->y_scale + fl;
        if (trend_out) trend_out[h] = trend;
        yhat[h] = trend * (1.0 + xm) + xa * info->y_scale;
    }
    return 0;
}

/* ---- uncertainty intervals (Prophet.predict_uncertainty), seeded -------------------------------
 * fbprophet 0.5: sample_posterior_predictive draws uncertainty_samples (1000) futures per series
 * -- sample_predictive_trend: n ~ Poisson(S (T - 1)) new changepoints uniform on [1, T] (T = largest
 * scaled future time), slope changes Laplace(0, mean|delta| + 1e-8), appended to the fitted ones;
 * sample_model: yhat = trend (1 + Xb_m) + Xb_a + N(0, sigma_obs) y_scale -- and predict_uncertainty
 * takes the (1 -+ interval_width) / 2 percentiles per future row (np.nanpercentile, linear).
 * The reference computes them (prophet_scorer.py:70) and drops them (:86).  numpy's global generator
 * is unseeded there, so parity is DEFINED here: a counter-based generator keyed by
 * (seed, series key, sample, stream) -- stream 0: changepoints, consumed in order; stream 1: the
 * noise of future row h at counters 2h, 2h + 1 -- with every transcendental from det_math.h.  The
 * n uniform changepoint times are generated already sorted (order statistics by the spacing
 * recurrence): the same joint distribution as numpy's draw-then-sort, without storing them.  The HIP
 * kernels (tsf_interval_kernels.h) consume the same stream in the same order: bit-identical. */
static uint64_t cn_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static uint64_t cn_rng_key(uint64_t seed, uint64_t series_key, uint64_t sample, uint64_t stream)
{
    return cn_mix64(cn_mix64(cn_mix64(cn_mix64(seed) ^ series_key) ^ sample) ^ stream);
}
static double cn_u01(uint64_t key, uint64_t ctr)        /* in (0, 1) */
{
    const uint64_t x = cn_mix64(key + 0x9E3779B97F4A7C15ULL * ctr);
    return ((double)(x >> 11) + 0.5) * 1.1102230246251565e-16;
}
static int cn_poisson(double lam, uint64_t key, uint64_t *ctr)
{
    int n = 0;
    double rest = lam;
    while (rest > 0.0 && n < 100000) {       /* Knuth's product method on pieces of at most 8 */
        const double piece = rest < 8.0 ? rest : 8.0;
        const double L = det_exp(-piece);
        double p = 1.0;
        int k = 0;
        do { p = p * cn_u01(key, (*ctr)++); ++k; } while (p > L);
        n += k - 1;
        rest = rest - piece;
    }
    return n;
}

typedef struct {            /* trend state swept along increasing t */
    double k, m;
    int ih;                 /* fitted changepoints already passed */
    int inew, n_new;        /* sampled changepoints already passed / in total */
    double u_prev;          /* last order statistic */
    double next_t, next_delta;
    uint64_t ctr;
} cn_trend_sweep;

int cn_predict_intervals(const cn_spec *sp, const cn_fitinfo *info, const double *theta,
                         const double *t_change, int H, const int64_t *ds, double floor_, double cap,
                         const double *extra_future, int n_samples, double interval_width,
                         uint64_t seed, uint64_t series_key, double *lower, double *upper)
{
    const int S = info->S, K = info->K;
    const double *delta = theta + 3, *beta = theta + 3 + S;
    const double fl = (sp->growth == 1) ? floor_ : 0.0;
    const double ys = info->y_scale;
    const double cap_sc = (sp->growth == 1) ? (cap - fl) / ys : 0.0;
    const double sigma = det_exp(theta[2]);
    int perm[CN_MAX_P], Ka;
    double prior[CN_MAX_P];
    build_perm(sp, K, perm, prior, &Ka);
    const double tsc = (double)info->t_scale_ns;
    const int nf = K - sp->n_extra;
    double *t = (double *)malloc(sizeof(double) * H), *xa = (double *)malloc(sizeof(double) * H);
    double *opm = (double *)malloc(sizeof(double) * H);
    double *samp = (double *)malloc(sizeof(double) * (size_t)H * n_samples);
    double Tm = -INFINITY;
    for (int h = 0; h < H; ++h) {
        t[h] = (double)(ds[h] - info->start_ns) / tsc;
        if (t[h] > Tm) Tm = t[h];
        double row[CN_MAX_P], a = 0.0, mm = 0.0;
        fourier_row(sp, ds[h], row);
        for (int e = 0; e < sp->n_extra; ++e) row[nf + e] = extra_future[(size_t)e * H + h];
        for (int j = 0; j < Ka; ++j) a = fma(row[perm[j]], beta[perm[j]], a);
        for (int j = Ka; j < K; ++j) mm = fma(row[perm[j]], beta[perm[j]], mm);
        xa[h] = a * ys; opm[h] = 1.0 + mm;
    }
    /* fbprophet: S = len(changepoints_t) (1 with the dummy changepoint), lambda = mean|delta| + 1e-8 */
    const int S_cp = S > 0 ? S : 1;
    double lam_sum = 0.0;
    for (int j = 0; j < S; ++j) lam_sum = lam_sum + fabs(delta[j]);
    const double lambda_ = lam_sum / (double)S_cp + 1e-8;
    const double rate = (Tm > 1.0) ? (double)S_cp * (Tm - 1.0) : 0.0;
    for (int s = 0; s < n_samples; ++s) {
        const uint64_t kcp = cn_rng_key(seed, series_key, (uint64_t)s, 0);
        const uint64_t knz = cn_rng_key(seed, series_key, (uint64_t)s, 1);
        cn_trend_sweep w;
        double t_last = -INFINITY;
        memset(&w, 0, sizeof(w));
        for (int h = 0; h < H; ++h) {
            if (h == 0 || t[h] < t_last) {       /* (re)start the sweep: the stream is replayed from 0 */
                w.k = theta[0]; w.m = theta[1]; w.ih = 0; w.inew = 0; w.ctr = 0; w.u_prev = 0.0;
                w.n_new = (rate > 0.0) ? cn_poisson(rate, kcp, &w.ctr) : 0;
                w.next_t = INFINITY;
            }
            t_last = t[h];
            while (w.ih < S && t[h] >= t_change[w.ih]) {           /* fitted changepoints */
                const double dj = delta[w.ih], kn = w.k + dj;
                if (sp->growth == 0) w.m = w.m + ((-t_change[w.ih]) * dj);
                else w.m = w.m + (t_change[w.ih] - w.m) * (1.0 - w.k / kn);
                w.k = kn; w.ih++;
            }
            for (;;) {                                               /* sampled changepoints */
                if (w.next_t == INFINITY && w.inew < w.n_new) {
                    const double v = cn_u01(kcp, w.ctr++);
                    const double rem = (double)(w.n_new - w.inew);
                    const double pw = det_exp(det_log(v) / rem);
                    w.u_prev = 1.0 - (1.0 - w.u_prev) * pw;
                    w.next_t = 1.0 + w.u_prev * (Tm - 1.0);
                    const double ul = cn_u01(kcp, w.ctr++);
                    w.next_delta = (ul < 0.5) ? lambda_ * det_log(2.0 * ul) : -(lambda_ * det_log(2.0 * (1.0 - ul)));
                }
                if (!(w.next_t <= t[h])) break;
                const double dj = w.next_delta, kn = w.k + dj;
                if (sp->growth == 0) w.m = w.m + ((-w.next_t) * dj);
                else w.m = w.m + (w.next_t - w.m) * (1.0 - w.k / kn);
                w.k = kn; w.inew++; w.next_t = INFINITY;
            }
            double gtr;
            if (sp->growth == 0) gtr = fma(w.k, t[h], w.m);
            else gtr = cap_sc * (1.0 / (1.0 + det_exp(-(w.k * (t[h] - w.m)))));
            const double trend = gtr * ys + fl;
            const double u1 = cn_u01(knz, 2 * (uint64_t)h), u2 = cn_u01(knz, 2 * (uint64_t)h + 1);
            double sn, cs;
            det_sincos(6.283185307179586 * u2, &sn, &cs);
            const double z = sqrt(-2.0 * det_log(u1)) * cs;
            samp[(size_t)h * n_samples + s] = trend * opm[h] + xa[h] + (z * sigma) * ys;
        }
    }
    const double pl = (1.0 - interval_width) / 2.0, pu = (1.0 + interval_width) / 2.0;
    for (int h = 0; h < H; ++h) {
        double *v = samp + (size_t)h * n_samples;
        for (int i = 1; i < n_samples; ++i) {       /* insertion sort: exact, n is ~1000 */
            const double x = v[i];
            int j = i - 1;
            while (j >= 0 && v[j] > x) { v[j + 1] = v[j]; --j; }
            v[j + 1] = x;
        }
        for (int which = 0; which < 2; ++which) {
            const double pos = (which ? pu : pl) * (double)(n_samples - 1);
            int lo = (int)floor(pos);
            if (lo > n_samples - 1) lo = n_samples - 1;
            const int hi = lo + 1 < n_samples ? lo + 1 : n_samples - 1;
            const double val = v[lo] + (v[hi] - v[lo]) * (pos - (double)lo);
            if (which) upper[h] = val; else lower[h] = val;
        }
    }
    free(t); free(xa); free(opm); free(samp);
    return 0;
}

/* Quadratic-form fit with every evaluation cross-checked against the residual form.
 * chk_out[0] = max |f_quad - f_resid| / max(1, |f_resid|), chk_out[1] = max relative 2-norm
 * gradient difference, over all evaluations of the fit. */
int cn_fit_checked(const cn_spec *sp, int T, const int64_t *ds, const double *y, double floor_,
                   double cap, const double *extra, double *theta_out, cn_fitinfo *info,
                   double *chk_out)
{
    int err;
    memset(info, 0, sizeof(*info));
    cn_series *se = cn_prepare(sp, T, ds, y, floor_, cap, extra, &err);
    if (!se) { info->status = err; return 0; }
    fill_info(se, info);
    double th0[CN_MAX_P], th[CN_MAX_P];
    memset(th0, 0, sizeof(th0));
    th0[0] = se->k0; th0[1] = se->m0;
    chk_out[0] = chk_out[1] = 0.0;
    if (se->constant_y || !(sp->growth == 0 && se->Ka == se->K)) { info->status = CN_ERR_SIZE; free_series(se); return 0; }
    se->gram = 1; se->check = 1;
    cn_build_gram(se);
    cn_lbfgs(se, sp, th0, th, info);
    info->pad_ = se->n_resid;
    chk_out[0] = se->chk_f; chk_out[1] = se->chk_g;
    to_original(se, th, theta_out, 1);
    free_series(se);
    return 0;
}

/* libm cross-check hooks for tests */
double cn_det_exp(double x) { return det_exp(x); }
double cn_det_log(double x) { return det_log(x); }
void cn_det_sincos(double x, double *s, double *c) { det_sincos(x, s, c); }
