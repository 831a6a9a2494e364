/*
 * oracle/stan_lbfgs.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped, never timed
 * as the product).  PARITY UNPINNED: see oracle/fbprophet_restated.py header.
 *
 * Plain-C restatement of what `StanModel.optimizing(data, init, algorithm='LBFGS',
 * iter=1e4)` does for fbprophet 0.5's prophet.stan -- the call the reference reaches through
 * `Prophet.fit` at /root/reference/src/jobs/prophet_modeler.py:66:
 *
 *   oracle_fg()        -log_prob(theta) and its gradient (prophet.stan model block,
 *                      jacobian=false, `~` constants dropped).  Cumulative-sum form of the
 *                      dense A*delta products; tests check it against the literal dense-A
 *                      numpy form in fbprophet_restated.py and against finite differences.
 *   oracle_lbfgs()     stan::optimization::BFGSMinimizer<..., LBFGSUpdate> ::step() loop
 *                      (stan 2.19 src/stan/optimization/bfgs.hpp), with
 *   wolfe_line_search(), wolfe_zoom(), cubic_interp()
 *                      (bfgs_linesearch.hpp) and the two-loop recursion of
 *                      LBFGSUpdate::search_direction (bfgs_update.hpp).
 *   Defaults (stan::services::optimize::lbfgs as driven by pystan 2.19.1.1):
 *     history 5, init_alpha 1e-3, tol_obj 1e-12, tol_rel_obj 1e4, tol_grad 1e-8,
 *     tol_rel_grad 1e7, tol_param 1e-8, c1 1e-4, c2 0.9, minAlpha 1e-12, maxLSIts 20,
 *     maxLSRestarts 10.  The relative tolerances are multiplied by DBL_EPSILON.
 *
 * theta layout: [k, m, log(sigma_obs), delta[S], beta[K]].
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_MAX_HIST 16

typedef struct {
    int32_t T, K, S, growth;      /* growth: 0 linear, 1 logistic */
    const double *t;              /* [T] scaled time */
    const double *y;              /* [T] scaled y */
    const double *cap;            /* [T] scaled cap (logistic) */
    const double *X;              /* [T*K] row-major design matrix */
    const double *s_a, *s_m;      /* [K] additive / multiplicative masks */
    const double *sigmas;         /* [K] prior scales */
    const double *t_change;       /* [S] */
    double tau;
} oracle_data;

typedef struct {
    int32_t max_iter;      /* 10000 */
    int32_t history;       /* 5 */
    double init_alpha;     /* 1e-3 */
    double tol_obj;        /* 1e-12 */
    double tol_rel_obj;    /* 1e4  (x eps) */
    double tol_grad;       /* 1e-8 */
    double tol_rel_grad;   /* 1e7  (x eps) */
    double tol_param;      /* 1e-8 */
} oracle_opts;

typedef struct {
    int32_t status;        /* Stan TERM_* code */
    int32_t n_iter;
    int32_t n_eval;
    double f;
} oracle_result;

enum { TERM_SUCCESS = 0, TERM_ABSX = 10, TERM_ABSF = 20, TERM_RELF = 21, TERM_ABSGRAD = 30,
       TERM_RELGRAD = 31, TERM_MAXIT = 40, TERM_LSFAIL = -1 };

static double dot(const double *a, const double *b, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* f = -log_prob, g = d f / d theta.  Returns 0 if f and g are finite, non-zero otherwise
 * (stan::optimization::ModelAdaptor::operator()). */
int oracle_fg(const oracle_data *d, const double *th, double *f_out, double *g)
{
    const int T = d->T, K = d->K, S = d->S;
    const double k = th[0], m = th[1], ls = th[2];
    const double *delta = th + 3, *beta = th + 3 + S;
    const double sigma = exp(ls);
    const double inv_s2 = 1.0 / (sigma * sigma);
    const int P = 3 + S + K;
    double ks[S + 2], mc[S + 2], dK[S + 2], dM[S + 2], gb_a[K + 1], gb_m[K + 1];

    ks[0] = k;
    for (int j = 0; j < S; ++j) ks[j + 1] = ks[j] + delta[j];
    mc[0] = m;
    if (d->growth == 0) {
        for (int j = 0; j < S; ++j) mc[j + 1] = mc[j] + (-d->t_change[j] * delta[j]);
    } else {
        for (int j = 0; j < S; ++j) {
            double gamma = (d->t_change[j] - mc[j]) * (1.0 - ks[j] / ks[j + 1]);
            mc[j + 1] = mc[j] + gamma;
        }
    }
    for (int c = 0; c <= S; ++c) dK[c] = dM[c] = 0.0;
    for (int j = 0; j < K; ++j) gb_a[j] = gb_m[j] = 0.0;

    double sse = 0.0;
    for (int i = 0; i < T; ++i) {
        const double ti = d->t[i];
        int c = 0;
        while (c < S && ti >= d->t_change[c]) ++c;   /* A[i,:] has c leading ones */
        const double *x = d->X + (size_t)i * K;
        double xm = 0.0, xa = 0.0;
        for (int j = 0; j < K; ++j) {
            xm += x[j] * (beta[j] * d->s_m[j]);
            xa += x[j] * (beta[j] * d->s_a[j]);
        }
        double trend, dz_dtrend = 0.0;
        if (d->growth == 0) {
            trend = ks[c] * ti + mc[c];
        } else {
            double z = ks[c] * (ti - mc[c]);
            double sg = 1.0 / (1.0 + exp(-z));
            trend = d->cap[i] * sg;
            dz_dtrend = d->cap[i] * sg * (1.0 - sg);
        }
        const double mu = trend * (1.0 + xm) + xa;
        const double r = d->y[i] - mu;
        sse += r * r;
        const double dmu = -r * inv_s2;
        const double dmt = dmu * trend;
        for (int j = 0; j < K; ++j) {
            gb_a[j] += x[j] * dmu;
            gb_m[j] += x[j] * dmt;
        }
        const double dtrend = dmu * (1.0 + xm);
        if (d->growth == 0) {
            dK[c] += dtrend * ti;
            dM[c] += dtrend;
        } else {
            const double dz = dtrend * dz_dtrend;
            dK[c] += dz * (ti - mc[c]);
            dM[c] += -dz * ks[c];
        }
    }

    double sabs = 0.0, sb = 0.0;
    for (int j = 0; j < S; ++j) sabs += fabs(delta[j]);
    for (int j = 0; j < K; ++j) { double q = beta[j] / d->sigmas[j]; sb += q * q; }
    const double f = 0.5 * k * k / 25.0 + 0.5 * m * m / 25.0 + sabs / d->tau
                   + 2.0 * sigma * sigma + 0.5 * sb + (double)T * ls + 0.5 * sse * inv_s2;

    double gk, gm;
    double *gd = g + 3;
    if (d->growth == 0) {
        /* g_delta[j] = sum_{c>j} (dK[c] - t_change[j] * dM[c]) */
        double sK = 0.0, sM = 0.0;
        for (int c = S; c >= 1; --c) {
            sK += dK[c]; sM += dM[c];
            gd[c - 1] = sK - d->t_change[c - 1] * sM;
        }
        gk = sK + dK[0];
        gm = sM + dM[0];
    } else {
        double abar = dM[S];
        for (int c = S - 1; c >= 0; --c) {
            const double ratio = ks[c] / ks[c + 1];
            const double rho_bar = abar * (d->t_change[c] - mc[c]);
            dK[c] += rho_bar * (-1.0 / ks[c + 1]);
            dK[c + 1] += rho_bar * (ratio / ks[c + 1]);
            abar = dM[c] + abar * ratio;
        }
        gm = abar;
        double sK = 0.0;
        for (int c = S; c >= 1; --c) { sK += dK[c]; gd[c - 1] = sK; }
        gk = sK + dK[0];
    }
    g[0] = gk + k / 25.0;
    g[1] = gm + m / 25.0;
    g[2] = (double)T - sse * inv_s2 + 4.0 * sigma * sigma;
    for (int j = 0; j < S; ++j) {
        double sgn = (delta[j] > 0.0) - (delta[j] < 0.0);
        gd[j] += sgn / d->tau;
    }
    double *gbeta = g + 3 + S;
    for (int j = 0; j < K; ++j)
        gbeta[j] = gb_a[j] * d->s_a[j] + gb_m[j] * d->s_m[j]
                 + beta[j] / (d->sigmas[j] * d->sigmas[j]);
    *f_out = f;
    if (!isfinite(f)) return 2;
    for (int i = 0; i < P; ++i) if (!isfinite(g[i])) return 3;
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* bfgs_linesearch.hpp                                                                   */
/* ------------------------------------------------------------------------------------ */

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

typedef struct {
    const oracle_data *d;
    int P;
    int n_eval;
} fctx;

static int eval(fctx *c, const double *x, double *f, double *g)
{
    c->n_eval++;
    return oracle_fg(c->d, x, f, g);
}

static void axpy_to(double *out, const double *x, double a, const double *p, int n)
{
    for (int i = 0; i < n; ++i) out[i] = x[i] + a * p[i];
}

static int wolfe_zoom(fctx *c, double *alpha, double *newX, double *newF, double *newDF,
                      const double *x, double f, const double *p, double c1dfp, double c2dfp,
                      double alo, double aloF, double aloDFp, double ahi, double ahiF,
                      double ahiDFp, double min_range)
{
    const int n = c->P;
    double d1, d2, newDFp;
    int itNum = 0;
    while (1) {
        itNum++;
        if (fabs(alo - ahi) < min_range) return 1;
        if (itNum % 5 == 0) {
            *alpha = 0.5 * (alo + ahi);
        } else {
            d1 = aloDFp + ahiDFp - 3.0 * (aloF - ahiF) / (alo - ahi);
            d2 = sqrt(d1 * d1 - aloDFp * ahiDFp);
            if (ahi < alo) d2 = -d2;
            *alpha = ahi - (ahi - alo) * (ahiDFp + d2 - d1) / (ahiDFp - aloDFp + 2.0 * d2);
            double lo = fmin(alo, ahi), hi = fmax(alo, ahi), w = fabs(alo - ahi);
            if (!isfinite(*alpha) || *alpha < lo + 0.01 * w || *alpha > hi - 0.01 * w)
                *alpha = 0.5 * (alo + ahi);
        }
        axpy_to(newX, x, *alpha, p, n);
        while (eval(c, newX, newF, newDF)) {
            *alpha = 0.5 * (*alpha + fmin(alo, ahi));
            if (fabs(fmin(alo, ahi) - *alpha) < min_range) return 1;
            axpy_to(newX, x, *alpha, p, n);
        }
        newDFp = dot(newDF, p, n);
        if (*newF > (f + *alpha * c1dfp) || *newF >= aloF) {
            ahi = *alpha; ahiF = *newF; ahiDFp = newDFp;
        } else {
            if (fabs(newDFp) <= -c2dfp) break;
            if (newDFp * (ahi - alo) >= 0) { ahi = alo; ahiF = aloF; ahiDFp = aloDFp; }
            alo = *alpha; aloF = *newF; aloDFp = newDFp;
        }
    }
    return 0;
}

static int wolfe_line_search(fctx *c, double *alpha, double *x1, double *f1, double *g1,
                             const double *p, const double *x0, double f0, const double *g0,
                             double c1, double c2, double minAlpha, int maxLSIts,
                             int maxLSRestarts)
{
    const int n = c->P;
    const double dfp = dot(g0, p, n);
    const double c1dfp = c1 * dfp, c2dfp = c2 * dfp;
    double alpha0 = minAlpha, prevF = f0, prevDFp = dfp, newDFp;
    int retCode = 0, nits = 0, lsRestarts = 0, ret;
    while (1) {
        if (nits >= maxLSIts) { retCode = 1; break; }
        axpy_to(x1, x0, *alpha, p, n);
        ret = eval(c, x1, f1, g1);
        if (ret != 0) {
            if (lsRestarts >= maxLSRestarts) { retCode = 1; break; }
            *alpha = 0.5 * (alpha0 + *alpha);
            lsRestarts++;
            continue;
        }
        lsRestarts = 0;
        newDFp = dot(g1, p, n);
        if (*f1 > f0 + *alpha * c1dfp || (*f1 >= prevF && nits > 0)) {
            retCode = wolfe_zoom(c, alpha, x1, f1, g1, x0, f0, p, c1dfp, c2dfp,
                                 alpha0, prevF, prevDFp, *alpha, *f1, newDFp, 1e-16);
            break;
        }
        if (fabs(newDFp) <= -c2dfp) { retCode = 0; break; }
        if (newDFp >= 0) {
            retCode = wolfe_zoom(c, alpha, x1, f1, g1, x0, f0, p, c1dfp, c2dfp,
                                 *alpha, *f1, newDFp, alpha0, prevF, prevDFp, 1e-16);
            break;
        }
        alpha0 = *alpha; prevF = *f1; prevDFp = newDFp;
        *alpha *= 10.0;
        nits++;
    }
    return retCode;
}

/* ------------------------------------------------------------------------------------ */
/* bfgs.hpp BFGSMinimizer::initialize + step loop, LBFGSUpdate                            */
/* ------------------------------------------------------------------------------------ */

void oracle_default_opts(oracle_opts *o)
{
    o->max_iter = 10000; o->history = 5; o->init_alpha = 1e-3; o->tol_obj = 1e-12;
    o->tol_rel_obj = 1e4; o->tol_grad = 1e-8; o->tol_rel_grad = 1e7; o->tol_param = 1e-8;
}

int oracle_lbfgs(const oracle_data *d, const oracle_opts *o, const double *theta0,
                 double *theta_out, oracle_result *res)
{
    const int P = 3 + d->S + d->K;
    const int H = o->history > ORACLE_MAX_HIST ? ORACLE_MAX_HIST : o->history;
    const double eps = DBL_EPSILON;
    const double c1 = 1e-4, c2 = 0.9, minAlpha = 1e-12;
    const int maxLSIts = 20, maxLSRestarts = 10;
    fctx c = { d, P, 0 };

    double *buf = (double *)calloc((size_t)P * (8 + 2 * H), sizeof(double));
    if (!buf) return -100;
    double *xk = buf, *gk = xk + P, *pk = gk + P, *xk_1 = pk + P, *gk_1 = xk_1 + P,
           *pk_1 = gk_1 + P, *sk = pk_1 + P, *yk = sk + P;
    double *Sb = yk + P, *Yb = Sb + (size_t)P * H;
    double rho[ORACLE_MAX_HIST], alphas[ORACLE_MAX_HIST];
    int hist_len = 0, hist_head = 0;   /* circular buffer: oldest at hist_head */
    double gammak = 1.0;
    double fk, fk_1 = 0.0, alpha = o->init_alpha, alpha0;
    int itNum = 0, ret = 0;

    memcpy(xk, theta0, sizeof(double) * P);
    /* initialize(): evaluate at the initial point; Stan throws if non-finite */
    if (eval(&c, xk, &fk, gk)) {
        memcpy(theta_out, theta0, sizeof(double) * P);
        res->status = -2; res->n_iter = 0; res->n_eval = c.n_eval; res->f = fk;
        free(buf);
        return 0;
    }
    for (int i = 0; i < P; ++i) pk[i] = -gk[i];

    while (ret == 0) {
        int resetB;
        itNum++;
        resetB = (itNum == 1) ? 1 : 0;
        while (1) {
            if (resetB) for (int i = 0; i < P; ++i) pk[i] = -gk[i];
            if (itNum > 1 && resetB != 2) {
                alpha0 = alpha = fmin(1.0, 1.01 * cubic_interp6(dot(gk_1, pk_1, P), alpha,
                                                                fk - fk_1, dot(gk, pk, P),
                                                                minAlpha, 1.0));
            } else {
                alpha0 = alpha = o->init_alpha;
            }
            (void)alpha0;
            int rc = wolfe_line_search(&c, &alpha, xk_1, &fk_1, gk_1, pk, xk, fk, gk,
                                       c1, c2, minAlpha, maxLSIts, maxLSRestarts);
            if (rc) {
                if (resetB) { ret = TERM_LSFAIL; goto done; }
                resetB = 2;
                continue;
            }
            break;
        }
        /* swap so that k is the most recent iterate */
        { double tf = fk; fk = fk_1; fk_1 = tf; }
        { double *tp; tp = xk; xk = xk_1; xk_1 = tp; tp = gk; gk = gk_1; gk_1 = tp;
          tp = pk; pk = pk_1; pk_1 = tp; }
        for (int i = 0; i < P; ++i) { sk[i] = xk[i] - xk_1[i]; yk[i] = gk[i] - gk_1[i]; }
        const double gradNorm = sqrt(dot(gk, gk, P));
        const double stepNorm = sqrt(dot(sk, sk, P));

        /* LBFGSUpdate::update */
        const double skyk = dot(yk, sk, P);
        const double ykyk = dot(yk, yk, P);
        if (resetB) {
            const double B0fact = ykyk / skyk;
            hist_len = 0; hist_head = 0;
            for (int i = 0; i < P; ++i) pk_1[i] /= B0fact;
            alpha = alpha * B0fact;
        }
        gammak = skyk / ykyk;
        {
            int slot;
            if (hist_len < H) { slot = (hist_head + hist_len) % H; hist_len++; }
            else { slot = hist_head; hist_head = (hist_head + 1) % H; }
            rho[slot] = 1.0 / skyk;
            memcpy(Sb + (size_t)slot * P, sk, sizeof(double) * P);
            memcpy(Yb + (size_t)slot * P, yk, sizeof(double) * P);
        }
        /* LBFGSUpdate::search_direction */
        for (int i = 0; i < P; ++i) pk[i] = -gk[i];
        for (int h = hist_len - 1; h >= 0; --h) {
            int slot = (hist_head + h) % H;
            const double *si = Sb + (size_t)slot * P, *yi = Yb + (size_t)slot * P;
            double a = rho[slot] * dot(si, pk, P);
            for (int i = 0; i < P; ++i) pk[i] -= a * yi[i];
            alphas[h] = a;
        }
        for (int i = 0; i < P; ++i) pk[i] *= gammak;
        for (int h = 0; h < hist_len; ++h) {
            int slot = (hist_head + h) % H;
            const double *si = Sb + (size_t)slot * P, *yi = Yb + (size_t)slot * P;
            double b = rho[slot] * dot(yi, pk, P);
            for (int i = 0; i < P; ++i) pk[i] += (alphas[h] - b) * si[i];
        }

        /* convergence checks */
        const double dF = fabs(fk_1 - fk);
        const double fmaxv = fmax(fabs(fk_1), fmax(fabs(fk), 1.0));
        if (dF < o->tol_obj) ret = TERM_ABSF;
        else if (dF < o->tol_rel_obj * eps * fmaxv) ret = TERM_RELF;
        else if (gradNorm < o->tol_grad) ret = TERM_ABSGRAD;
        else if (-dot(gk, pk, P) / fmax(fabs(fk), 1.0) < o->tol_rel_grad * eps) ret = TERM_RELGRAD;
        else if (stepNorm < o->tol_param) ret = TERM_ABSX;
        else if (itNum >= o->max_iter) ret = TERM_MAXIT;
        else ret = TERM_SUCCESS;
    }
done:
    memcpy(theta_out, xk, sizeof(double) * P);
    res->status = ret; res->n_iter = itNum; res->n_eval = c.n_eval; res->f = fk;
    free(buf);
    return 0;
}
