// tsf_newton_kernels.h -- Stan's Newton optimiser, one wavefront per series.
//
// fbprophet 0.5 runs optimizing(algorithm='Newton') when a history has fewer than 100 rows and
// as the retry after an L-BFGS RuntimeError (UPSTREAM-RECALL forecaster.py fit; SURVEY.md 8a
// U9); the reference reaches it through Prophet().fit(pdf) at
// /root/reference/src/jobs/prophet_modeler.py:65-66.  The algorithm (stan 2.19
// model/grad_hess_log_prob.hpp, optimization/newton.hpp, services/optimize/newton.hpp) and its
// canonical operation order are stated in oracle/prophet_canon.c (cn_newton, cn_tridiag_ql); this
// file executes exactly that sequence:
//
//   lane p                = parameter p (P <= 64: Newton is for short series, K is small)
//   gradient evaluations  = eval_fg (residual form), 4 per parameter for the finite-difference
//                           Hessian, the perturbed coordinate selected by lane
//   A[d][p]               = fma chain over the 4 perturbations, lane p, written to LDS row d
//   H = A + A^T           in place, pair (a, b) handled by lane b
//   eigen-decomposition   = Householder tridiagonalisation + implicit QL in LDS (ql_lds; round 1
//                           used a round-robin Jacobi: ten times the arithmetic)
//   proj, step            = lane-parallel fma chains with the other operand broadcast by readlane
//   step halving          = Stan's loop, one evaluation per trial
//
// Cost per Newton iteration: 4 P + 1 evaluations, one P x P eigen-decomposition, ~30 trial
// evaluations.
#pragma once
#include "tsf_fit_kernels.h"

namespace tsf {

// Symmetric eigen-decomposition by Householder tridiagonalisation + implicit QL with shifts, in the
// operation order of oracle cn_tridiag_ql (every "for all j" there is one lane per j here, sums
// over lanes are bfly_sum, sums over k inside a lane are sequential fma chains, scalars are
// computed identically by every lane).  About a tenth of the Jacobi's arithmetic for P = 34 and a
// far shorter dependent chain (round 1: ~0.5 M dependent instructions per decomposition).
// Am: n x n symmetric, row stride PM; the eigenvectors replace it (columns; Vm must be Am: one
// matrix in LDS instead of two), sc: d, e, hh, q scratch of 64 doubles each.  Returns the eigenvalue of lane j (0 for j >= n).
struct QlScratch { double d[W], e[W], hh[W], q[W]; };

__device__ __forceinline__ double ql_pythag(double a, double b)
{
    const double absa = __builtin_fabs(a), absb = __builtin_fabs(b);
    if (absa > absb) { const double r = absb / absa; return absa * __builtin_sqrt(1.0 + r * r); }
    if (absb == 0.0) return 0.0;
    const double r = absa / absb;
    return absb * __builtin_sqrt(1.0 + r * r);
}

#ifdef TSF_QUAD_TIMING
#define QL_LAP(k) do { if (qlt) { const long long t_ = __builtin_readcyclecounter(); qlt[k] += t_ - qlt0; qlt0 = t_; } } while (0)
#else
#define QL_LAP(k) do { } while (0)
#endif
// part 1: Householder tridiagonalisation + Q in place; leaves d in sc.d, e in sc.e (e[i] couples i-1 and i)
__device__ __forceinline__ void ql_tridiag_q(int n, int PM, double *Am, double *Vm, QlScratch &sc, long long *qlt = nullptr)
{
#ifdef TSF_QUAD_TIMING
    long long qlt0 = __builtin_readcyclecounter();
#endif
    const int lane = lane_id();
    sc.e[lane] = 0.0; sc.hh[lane] = 0.0;
    TSF_WAVE_SYNC();
    // ---- Householder: zero A[i][0..i-2] for i = n-1 .. 2; reflector u kept in row i, u.u/2 in hh[i]
    for (int i = n - 1; i >= 2; --i) {
        const int l = i - 1;
        const double xj = (lane <= l) ? Am[i * PM + lane] : 0.0;
        const double sigma = bfly_sum((lane < l) ? xj * xj : 0.0);
        const double alpha = readlane_f64(xj, l);
        if (sigma == 0.0) {
            if (lane == 0) { sc.e[i] = alpha; sc.hh[i] = 0.0; }
            continue;
        }
        const double mu = __builtin_sqrt(sigma + alpha * alpha);
        const double beta = (alpha >= 0.0) ? -mu : mu;
        const double ul = alpha - beta;
        const double uj = (lane == l) ? ul : xj;
        const double H = 0.5 * (sigma + ul * ul);
        if (lane == l) Am[i * PM + l] = ul;
        TSF_WAVE_SYNC();
        double a = 0.0;                                     // p = A u / H, lane = row j
        // (the k loops of this routine read LDS in batches of 8 / 4 ahead of the fma chain: rolled,
        // every step of the chain waited for its own two LDS reads -- same operands, same order)
        if (lane <= l) {
            const double *rowp = Am + lane * PM, *up = Am + i * PM;
            int k = 0;
            for (; k + 8 <= l + 1; k += 8) {
                double rv[8], uv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { rv[u] = rowp[k + u]; uv[u] = up[k + u]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) a = __builtin_fma(rv[u], uv[u], a);
            }
            if (k + 4 <= l + 1) {
                double rv[4], uv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { rv[u] = rowp[k + u]; uv[u] = up[k + u]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) a = __builtin_fma(rv[u], uv[u], a);
                k += 4;
            }
            for (; k <= l; ++k) a = __builtin_fma(rowp[k], up[k], a);
        }
        const double pj = a / H;
        const double K = bfly_sum((lane <= l) ? uj * pj : 0.0) / (2.0 * H);
        const double qj = pj - K * uj;
        sc.q[lane] = qj;
        TSF_WAVE_SYNC();
        if (lane <= l) {                                    // A <- A - u q^T - q u^T
            double *rowp = Am + lane * PM;
            const double *up = Am + i * PM;
            int k = 0;
            for (; k + 4 <= l + 1; k += 4) {
                double rv[4], uv[4], qv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { rv[u] = rowp[k + u]; uv[u] = up[k + u]; qv[u] = sc.q[k + u]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) rowp[k + u] = __builtin_fma(-qj, uv[u], __builtin_fma(-uj, qv[u], rv[u]));
            }
            for (; k <= l; ++k) rowp[k] = __builtin_fma(-qj, up[k], __builtin_fma(-uj, sc.q[k], rowp[k]));
        }
        if (lane == 0) { sc.e[i] = beta; sc.hh[i] = H; }
        TSF_WAVE_SYNC();
    }
    QL_LAP(0);
    if (lane == 0 && n > 1) sc.e[1] = Am[1 * PM + 0];
    if (lane < n) sc.d[lane] = Am[lane * PM + lane];
    TSF_WAVE_SYNC();
    // ---- Q = H_{n-1} ... H_2 applied to the identity, IN PLACE (Vm == Am): H_i only mixes the
    // leading i x i block, which by then holds the product of the earlier reflectors (their own
    // rows, inside that block, have been consumed), while row i still holds u_i; lane = column c.
    // Same values as the oracle's separate V: outside the leading block V is the identity.
    if (lane < 2 && n > 0) {
        Vm[0 * PM + lane] = (lane == 0) ? 1.0 : 0.0;
        if (n > 1) Vm[1 * PM + lane] = (lane == 1) ? 1.0 : 0.0;
    }
    TSF_WAVE_SYNC();
    for (int i = 2; i < n; ++i) {
        const double Hi = sc.hh[i];
        const int l = i - 1;
        if (Hi != 0.0 && lane <= l) {
            double w = 0.0;
            const double *up = Am + i * PM;
            double *colp = Vm + lane;
            int k = 0;
            for (; k + 8 <= l + 1; k += 8) {
                double uv[8], vv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { uv[u] = up[k + u]; vv[u] = colp[(k + u) * PM]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) w = __builtin_fma(uv[u], vv[u], w);
            }
            for (; k <= l; ++k) w = __builtin_fma(up[k], colp[k * PM], w);
            w = w / Hi;
            int r = 0;
            for (; r + 4 <= l + 1; r += 4) {
                double uv[4], vv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { uv[u] = up[r + u]; vv[u] = colp[(r + u) * PM]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) colp[(r + u) * PM] = __builtin_fma(-uv[u], w, vv[u]);
            }
            for (; r <= l; ++r) colp[r * PM] = __builtin_fma(-up[r], w, colp[r * PM]);
        }
        TSF_WAVE_SYNC();
        if (lane <= i) { Vm[i * PM + lane] = (lane == i) ? 1.0 : 0.0; Vm[lane * PM + i] = (lane == i) ? 1.0 : 0.0; }
        TSF_WAVE_SYNC();
    }
}

// part 2: implicit QL on (sc.d, sc.e), rotations applied to the columns of V
__device__ __forceinline__ double ql_chain(int n, int PM, double *Vm, QlScratch &sc, long long *qlt = nullptr)
{
    const int lane = lane_id();
#ifdef TSF_QUAD_TIMING
    long long qlt0 = __builtin_readcyclecounter();
#endif
    // ---- implicit QL on (d, e); rotations applied to the columns of V, lane = row k.
    // d and e live in REGISTERS here, entry j in lane j (n <= 64): the scalar rotation chain reads
    // them with v_readlane instead of waiting for an LDS round trip in every step, the search for
    // the first negligible off-diagonal element is one lane-parallel test + ballot instead of up to
    // n dependent LDS reads, and no LDS hand-off is left inside the chain (a lane only touches its
    // own row of V).  Same operations on the same operands as the oracle's sequential loops.
    TSF_WAVE_SYNC();
    QL_LAP(1);
    double dv = (lane < n) ? sc.d[lane] : 0.0;
    double ev = (lane + 1 < n) ? sc.e[lane + 1] : 0.0;      // e shifted down by one, e[n-1] = 0
    for (int l = 0; l < n; ++l) {
        for (int guard = 0; guard < 60; ++guard) {
            // m: first index in [l, n-2] whose off-diagonal element is negligible, else n-1
            const double dn = __shfl_down(dv, 1, W);
            const double dd = __builtin_fabs(dv) + __builtin_fabs(dn);
            const bool tiny = lane >= l && lane < n - 1 && (__builtin_fabs(ev) + dd == dd);
            const unsigned long long mask = __ballot(tiny);
            const int m = mask ? (int)__builtin_ctzll(mask) : n - 1;
            if (m == l) break;
            const double dl = readlane_f64(dv, l), el = readlane_f64(ev, l);
            double g = (readlane_f64(dv, l + 1) - dl) / (2.0 * el);
            double r = ql_pythag(g, 1.0);
            g = readlane_f64(dv, m) - dl + el / (g + (g >= 0.0 ? __builtin_fabs(r) : -__builtin_fabs(r)));
            double s = 1.0, c = 1.0, p = 0.0;
            int i = m - 1;
            bool underflow = false;
            // V: rotation i mixes columns i and i + 1 (lane = row).  Column i + 1 is the column i of
            // the previous rotation: carried in a register (vcar) and stored once it is final; column
            // i is read one rotation ahead (its address does not depend on the chain), so the chain
            // never waits for LDS.
            double *vrow = Vm + lane * PM;
            double vcar = (lane < n) ? vrow[m] : 0.0;
            double v0n = (lane < n) ? vrow[m - 1] : 0.0;
            for (; i >= l; --i) {
                const double ei = readlane_f64(ev, i), di = readlane_f64(dv, i), di1 = readlane_f64(dv, i + 1);
                const double v0 = v0n;
                if (i > l && lane < n) v0n = vrow[i - 1];
                double f = s * ei;
                const double b = c * ei;
                r = __builtin_sqrt(__builtin_fma(f, f, g * g));
                if (lane == i + 1) ev = r;
                if (r == 0.0) {
                    if (lane == i + 1) dv = di1 - p;
                    if (lane == m) ev = 0.0;
                    underflow = true;
                    break;
                }
                { const double ri = 1.0 / r; s = f * ri; c = g * ri; }
                g = di1 - p;
                r = (di - g) * s + 2.0 * c * b;
                p = s * r;
                if (lane == i + 1) dv = g + p;
                g = c * r - b;
                if (lane < n) {
                    vrow[i + 1] = __builtin_fma(s, v0, c * vcar);
                    vcar = __builtin_fma(c, v0, -(s * vcar));
                }
            }
            if (lane < n) vrow[i + 1] = vcar;       // i = l - 1 after a full sweep, or the rotation that underflowed
            if (underflow) continue;
            if (lane == l) { dv = dv - p; ev = g; }
            if (lane == m) ev = 0.0;
        }
    }
    TSF_WAVE_SYNC();
    QL_LAP(2);
    return (lane < n) ? dv : 0.0;
}

// make_negative_definite_and_solve where the finite-difference Hessian already IS negative definite (cn_chol_neg_solve,
// round 6: a third of the iterations on BASELINE cfg5): -H = L L^T by Cholesky and (-H) step = g by two substitutions
// instead of an eigen-decomposition.  Lane i = row i.  H is the full symmetric matrix in Am; L goes into its STRICTLY lower
// triangle (the diagonal of L stays in lane j's register), -H[i][j] is read from row j (the upper triangle, never
// written), so a failed attempt -- a pivot that is not positive: H is not negative definite -- only has to copy the upper
// triangle back down before the eigen route runs.  gl: the lane's gradient entry.  Same operations in the same order as
// the oracle's.
__device__ __forceinline__ bool chol_neg_solve(int P, int PM, double *Am, double gl, double &step_out)
{
    const int lane = lane_id();
    double ljj_own = 1.0;
    bool ok = true;
    int j = 0;
    for (; j < P; ++j) {
        double s = 0.0;
        if (lane >= j && lane < P) {
            // four fma chains over k (chain k mod 4), their eight LDS reads in flight together
            double a0 = -Am[j * PM + lane], a1 = 0.0, a2 = 0.0, a3 = 0.0;
            const double *li = Am + lane * PM, *lj = Am + j * PM;
            int kk = 0;
            for (; kk + 3 < j; kk += 4) {
                const double l0 = li[kk], l1 = li[kk + 1], l2 = li[kk + 2], l3 = li[kk + 3];
                const double r0 = lj[kk], r1 = lj[kk + 1], r2 = lj[kk + 2], r3 = lj[kk + 3];
                a0 = __builtin_fma(-l0, r0, a0); a1 = __builtin_fma(-l1, r1, a1);
                a2 = __builtin_fma(-l2, r2, a2); a3 = __builtin_fma(-l3, r3, a3);
            }
            if (kk < j) a0 = __builtin_fma(-li[kk], lj[kk], a0);
            if (kk + 1 < j) a1 = __builtin_fma(-li[kk + 1], lj[kk + 1], a1);
            if (kk + 2 < j) a2 = __builtin_fma(-li[kk + 2], lj[kk + 2], a2);
            s = (a0 + a1) + (a2 + a3);
        }
        const double d = readlane_f64(s, j);
        if (!(d > 0.0)) { ok = false; break; }
        const double ljj = __builtin_sqrt(d);
        if (lane == j) ljj_own = ljj;
        if (lane > j && lane < P) Am[lane * PM + j] = s / ljj;
        TSF_WAVE_SYNC();
    }
    if (!ok) {
        // columns 0 .. j-1 of the strictly lower triangle hold L: H again, from the upper triangle
        for (int c = 0; c < j; ++c)
            if (lane > c && lane < P) Am[lane * PM + c] = Am[c * PM + lane];
        TSF_WAVE_SYNC();
        return false;
    }
    double r = (lane < P) ? gl : 0.0, z = 0.0;
    for (j = 0; j < P; ++j) {
        const double zj = readlane_f64(r, j) / readlane_f64(ljj_own, j);
        if (lane == j) z = zj;
        if (lane > j && lane < P) r = __builtin_fma(-Am[lane * PM + j], zj, r);
    }
    double r2 = z, st = 0.0;
    for (j = P - 1; j >= 0; --j) {
        const double sj = readlane_f64(r2, j) / readlane_f64(ljj_own, j);
        if (lane == j) st = sj;
        if (lane < j) r2 = __builtin_fma(-Am[j * PM + lane], sj, r2);
    }
    step_out = (lane < P) ? st : 0.0;
    TSF_WAVE_SYNC();
    return true;
}

__device__ __forceinline__ double ql_lds(int n, int PM, double *Am, double *Vm, QlScratch &sc, long long *qlt = nullptr)
{
    ql_tridiag_q(n, PM, Am, Vm, sc, qlt);
    return ql_chain(n, PM, Vm, sc, qlt);
}

// LDS of one Newton wave: the evaluation tables of eval_fg (no L-BFGS history), the QL scratch,
// then Am [PM][PM], Vm [PM][PM]
template <int KP>
struct NewtonLds {
    double th[TSF_MAX_P + W];
    double ks[NTAB + 1], mc[NTAB + 1];
    double tp1[NTAB], tp2[NTAB];
    double tot1[W + 1], tot2[W + 1];
    double d1[NTAB + 1], d2[NTAB + 1], rb[NTAB + 1], ab[NTAB + 1];
    double accR[KP];
    QlScratch ql;
};

template <int KP>
constexpr size_t newton_lds_bytes(int PM)
{
    return ((sizeof(NewtonLds<KP>) + 15) & ~(size_t)15) + (size_t)PM * PM * sizeof(double);
}

template <int KP, int GROWTH, int MODE>
__global__ __launch_bounds__(64) void newton_kernel(FitArgs a, int PM)
{
    constexpr int PPL = 1;
    extern __shared__ __align__(16) unsigned char smem[];
    NewtonLds<KP> &lds = *reinterpret_cast<NewtonLds<KP> *>(smem);
    double *Am = reinterpret_cast<double *>(smem + ((sizeof(NewtonLds<KP>) + 15) & ~(size_t)15));
    double *Vm = Am;          // ql_lds leaves the eigenvectors where the matrix was
    const int64_t n = blockIdx.x;
    if (n >= a.N) return;
    const int lane = threadIdx.x;
    const DevSpec *sp = a.sp;
    SeriesView sv;
    make_view<KP, PPL>(a, n, sv);
    for (int i = threadIdx.x; i < TSF_MAX_P + W; i += W) lds.th[i] = 0.0;
    TSF_WAVE_SYNC();
    const SeriesTab st = a.stab[n];
    if (lane == 0) {
        a.y_scale[n] = st.y_scale;
        if (!a.aligned || n == 0) a.grid_out[a.aligned ? 0 : n] = a.gtab[grid_index(a, n)].info;
    }
    double th[PPL], x[PPL], g[PPL], gx[PPL], step[PPL];
    th[0] = (lane == 0) ? st.k0 : (lane == 1 ? st.m0 : 0.0);
    x[0] = th[0]; g[0] = 0.0; gx[0] = 0.0; step[0] = 0.0;
    if (st.status0 != 0) {
        if (st.status0 == TSF_ST_CONSTANT && lane == 2) th[0] = -20.72326583694641;
        store_theta<PPL>(a, sv, n, th, a.theta);
        if (lane == 0) { a.status[n] = st.status0; a.n_iter[n] = 0; a.n_eval[n] = 0; a.fval[n] = 0.0; }
        return;
    }
    const int P = sv.P;
    const double epsilon = 1e-3, half_epsilon = 0.5 * epsilon;

    enum { S_INIT = 0, S_F0, S_FD, S_HALVE };
    int stage = S_INIT, ret = TSF_ST_MAXIT, it = 0, mI = 0, d = 0, pi = 0;
    double lp = 0.0, lastlp = 0.0, f0 = 0.0, f1 = 0.0, size = 2.0, acc = 0.0, fx = 0.0;
    for (;;) {
        FT_DECL;
        const bool bad = eval_fg<KP, GROWTH, MODE, PPL, false, NewtonLds<KP>>(sp, sv, lds, x, fx, gx FT_PASS);
        bool finish_iter = false, moved = false;
        if (stage == S_INIT) {
            if (bad) { ret = TSF_ST_INIT_NONFINITE; lp = -fx; break; }
            lp = -fx;
            stage = S_F0;
            x[0] = th[0];
            continue;
        }
        if (stage == S_F0) {
            if (bad) { ret = TSF_ST_NEWTON_FAIL; break; }
            lastlp = lp;
            f0 = -fx;
            g[0] = gx[0];
            d = 0; pi = 0; acc = 0.0;
            stage = S_FD;
            x[0] = (lane == 0) ? th[0] + (-2 * epsilon) : th[0];
            continue;
        }
        if (stage == S_FD) {
            if (bad) { ret = TSF_ST_NEWTON_FAIL; break; }
            const double coef = (pi == 0) ? 1.0 / 12.0 : (pi == 1 ? -2.0 / 3.0 : (pi == 2 ? 2.0 / 3.0 : -1.0 / 12.0));
            acc = __builtin_fma(half_epsilon * coef, -gx[0], acc);
            if (++pi == 4) {
                if (lane < P) Am[d * PM + lane] = acc;
                acc = 0.0; pi = 0; ++d;
            }
            if (d < P) {
                const double pert = (pi == 0) ? -2 * epsilon : (pi == 1 ? -1 * epsilon : (pi == 2 ? epsilon : 2 * epsilon));
                x[0] = (lane == d) ? th[0] + pert : th[0];
                continue;
            }
            // ---- H = A + A^T (in place; lane b owns the pairs (a, b), a < b, and its diagonal)
            TSF_WAVE_SYNC();
            for (int r = 0; r < P; ++r) {
                double u = 0.0, v = 0.0;
                const bool mine = lane < P && r <= lane;
                if (mine) { u = Am[r * PM + lane]; v = Am[lane * PM + r]; }
                TSF_WAVE_SYNC();
                if (mine) { const double h = u + v; Am[r * PM + lane] = h; Am[lane * PM + r] = h; }
                TSF_WAVE_SYNC();
            }
            // ---- make_negative_definite_and_solve: Cholesky where H is negative definite, else the eigen route
            if (!chol_neg_solve(P, PM, Am, g[0], step[0])) {
            const double lam = ql_lds(P, PM, Am, Vm, lds.ql);
            double pa = 0.0;
            for (int i = 0; i < P; ++i) {
                const double gi = -readlane_f64(g[0], i);
                const double vij = (lane < P) ? Vm[i * PM + lane] : 0.0;
                pa = __builtin_fma(vij, gi, pa);
            }
            const double proj = (lane < P) ? -pa / __builtin_fabs(lam) : 0.0;
            double sa = 0.0;
            for (int j = 0; j < P; ++j) {
                const double pj = readlane_f64(proj, j);
                const double vij = (lane < P) ? Vm[lane * PM + j] : 0.0;
                sa = __builtin_fma(vij, pj, sa);
            }
            step[0] = (lane < P) ? sa : 0.0;
            }
            x[0] = th[0];
            size = 2.0; f1 = -1e100;
            stage = S_HALVE;
            // fall through to the loop test below with no trial evaluated yet
        } else {   // S_HALVE: a trial point was evaluated
            f1 = bad ? -1e100 : -fx;
        }
        // ---- Stan's `while (f1 < f0)` step-halving loop
        if (f1 < f0) {
            size *= 0.5;
            if (size < 1e-50) { finish_iter = true; moved = false; }
            else { x[0] = th[0] - size * step[0]; continue; }
        } else {
            finish_iter = true; moved = true;
        }
        if (finish_iter) {
            ++it;
            if (moved) { th[0] = x[0]; lp = f1; }
            else lp = f0;
            // (the first comparison in Stan is against an lp that includes the constant terms)
            if (mI > 0 && __builtin_fabs(lp - lastlp) < 1e-8) { ret = TSF_ST_NEWTON_CONVERGED; break; }
            if (++mI >= a.opt.max_iter) { ret = TSF_ST_MAXIT; break; }
            stage = S_F0;
            x[0] = th[0];
        }
    }
    store_theta<PPL>(a, sv, n, th, a.theta);
    if (lane == 0) { a.status[n] = ret; a.n_iter[n] = it; a.n_eval[n] = sv.n_eval; a.fval[n] = -lp; }
}


// =========================================================================================
// Two parameters per lane (64 < P <= 128, or mixed additive / multiplicative columns, whose
// kernels hold two parameters per lane whatever P is): newton_kernel2.  Round 4: fbprophet
// retries EVERY failed L-BFGS fit with Stan's Newton (and starts with it below 100 rows); without
// this kernel such a series of a wide model was "dropped" where the reference would have fitted it.
// It serves a handful of series per call, so it is written for correctness, not for speed: the
// operations and their order are cn_tridiag_ql's / cn_newton's (oracle), entry p = lane + 64 s,
// sums over more than 64 entries by dotc's rule (the p >= 64 term fma'd onto the p - 64 term, then
// the butterfly), d and e of the QL iteration in LDS (every lane computes the scalar chain from
// broadcast reads), the eigenvectors in place.  One wave per workgroup; the 129 x 129 matrix takes
// 133 KB of LDS, so one workgroup per CU.
// =========================================================================================
struct QlScratch2 { double d[2 * W], e[2 * W], hh[2 * W], q[2 * W]; };

__device__ __forceinline__ void ql2_tridiag_q(int n, int PM, double *Am, QlScratch2 &sc)
{
    const int lane = lane_id();
#pragma unroll
    for (int s = 0; s < 2; ++s) { sc.e[lane + s * W] = 0.0; sc.hh[lane + s * W] = 0.0; }
    TSF_WAVE_SYNC();
    for (int i = n - 1; i >= 2; --i) {
        const int l = i - 1;
        double xj[2], uj[2], pj[2], qj[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) xj[s] = (lane + s * W <= l) ? Am[i * PM + lane + s * W] : 0.0;
        double part = (lane < l) ? xj[0] * xj[0] : 0.0;
        if (lane + W < l) part = __builtin_fma(xj[1], xj[1], part);
        const double sigma = bfly_sum(part);
        const double alpha = Am[i * PM + l];
        if (sigma == 0.0) {
            if (lane == 0) { sc.e[i] = alpha; sc.hh[i] = 0.0; }
            TSF_WAVE_SYNC();
            continue;
        }
        const double mu = __builtin_sqrt(sigma + alpha * alpha);
        const double beta = (alpha >= 0.0) ? -mu : mu;
        const double ul = alpha - beta;
#pragma unroll
        for (int s = 0; s < 2; ++s) uj[s] = (lane + s * W == l) ? ul : xj[s];
        const double H = 0.5 * (sigma + ul * ul);
        TSF_WAVE_SYNC();
        if (lane == 0) Am[i * PM + l] = ul;
        TSF_WAVE_SYNC();
#pragma unroll
        for (int s = 0; s < 2; ++s) {                          // p = A u / H, row j = lane + 64 s
            const int j = lane + s * W;
            double a = 0.0;
            if (j <= l) {
                const double *rowp = Am + j * PM, *up = Am + i * PM;
                for (int k = 0; k <= l; ++k) a = __builtin_fma(rowp[k], up[k], a);
            }
            pj[s] = a / H;
        }
        part = (lane <= l) ? uj[0] * pj[0] : 0.0;
        if (lane + W <= l) part = __builtin_fma(uj[1], pj[1], part);
        const double K = bfly_sum(part) / (2.0 * H);
#pragma unroll
        for (int s = 0; s < 2; ++s) { qj[s] = pj[s] - K * uj[s]; sc.q[lane + s * W] = qj[s]; }
        TSF_WAVE_SYNC();
#pragma unroll
        for (int s = 0; s < 2; ++s) {                          // A <- A - u q^T - q u^T, row j
            const int j = lane + s * W;
            if (j <= l) {
                double *rowp = Am + j * PM;
                const double *up = Am + i * PM;
                for (int k = 0; k <= l; ++k) rowp[k] = __builtin_fma(-qj[s], up[k], __builtin_fma(-uj[s], sc.q[k], rowp[k]));
            }
        }
        if (lane == 0) { sc.e[i] = beta; sc.hh[i] = H; }
        TSF_WAVE_SYNC();
    }
    if (lane == 0 && n > 1) sc.e[1] = Am[1 * PM + 0];
#pragma unroll
    for (int s = 0; s < 2; ++s) if (lane + s * W < n) sc.d[lane + s * W] = Am[(lane + s * W) * PM + lane + s * W];
    TSF_WAVE_SYNC();
    // Q = H_{n-1} ... H_2 applied to the identity, in place (see ql_tridiag_q): lane + 64 s = column c
    double *Vm = Am;
    if (lane < 2 && n > 0) {
        Vm[0 * PM + lane] = (lane == 0) ? 1.0 : 0.0;
        if (n > 1) Vm[1 * PM + lane] = (lane == 1) ? 1.0 : 0.0;
    }
    TSF_WAVE_SYNC();
    for (int i = 2; i < n; ++i) {
        const double Hi = sc.hh[i];
        const int l = i - 1;
        if (Hi != 0.0) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int c = lane + s * W;
                if (c <= l) {
                    double w = 0.0;
                    const double *up = Am + i * PM;
                    double *colp = Vm + c;
                    for (int k = 0; k <= l; ++k) w = __builtin_fma(up[k], colp[k * PM], w);
                    w = w / Hi;
                    for (int r = 0; r <= l; ++r) colp[r * PM] = __builtin_fma(-up[r], w, colp[r * PM]);
                }
            }
        }
        TSF_WAVE_SYNC();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int c = lane + s * W;
            if (c <= i) { Vm[i * PM + c] = (c == i) ? 1.0 : 0.0; Vm[c * PM + i] = (c == i) ? 1.0 : 0.0; }
        }
        TSF_WAVE_SYNC();
    }
}

// implicit QL on (sc.d, sc.e) in LDS, rotations applied to the columns of V (lane + 64 s = row); the
// eigenvalue of entry lane + 64 s comes back in lam[s]
__device__ __forceinline__ void ql2_chain(int n, int PM, double *Vm, QlScratch2 &sc, double (&lam)[2])
{
    const int lane = lane_id();
    TSF_WAVE_SYNC();
    {   // e shifted down by one, e[n-1] = 0
        double ev[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) ev[s] = (lane + s * W + 1 < n) ? sc.e[lane + s * W + 1] : 0.0;
        TSF_WAVE_SYNC();
#pragma unroll
        for (int s = 0; s < 2; ++s) sc.e[lane + s * W] = ev[s];
        TSF_WAVE_SYNC();
    }
    for (int l = 0; l < n; ++l) {
        for (int guard = 0; guard < 60; ++guard) {
            // m: first index in [l, n-2] whose off-diagonal element is negligible, else n-1
            bool tiny[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int j = lane + s * W;
                const bool in = j >= l && j < n - 1;
                const double dj = in ? sc.d[j] : 0.0, dn = in ? sc.d[j + 1] : 0.0, ej = in ? sc.e[j] : 0.0;
                const double dd = __builtin_fabs(dj) + __builtin_fabs(dn);
                tiny[s] = in && (__builtin_fabs(ej) + dd == dd);
            }
            const unsigned long long m0 = __ballot(tiny[0]), m1 = __ballot(tiny[1]);
            const int m = m0 ? (int)__builtin_ctzll(m0) : (m1 ? W + (int)__builtin_ctzll(m1) : n - 1);
            if (m == l) break;
            const double dl = sc.d[l], el = sc.e[l];
            double g = (sc.d[l + 1] - dl) / (2.0 * el);
            double r = ql_pythag(g, 1.0);
            g = sc.d[m] - dl + el / (g + (g >= 0.0 ? __builtin_fabs(r) : -__builtin_fabs(r)));
            double sn = 1.0, c = 1.0, p = 0.0;
            int i = m - 1;
            bool underflow = false;
            for (; i >= l; --i) {
                const double ei = sc.e[i], di = sc.d[i], di1 = sc.d[i + 1];
                const double f = sn * ei;
                const double b = c * ei;
                r = __builtin_sqrt(__builtin_fma(f, f, g * g));
                TSF_WAVE_SYNC();
                if (lane == 0) sc.e[i + 1] = r;
                if (r == 0.0) {
                    if (lane == 0) { sc.d[i + 1] = di1 - p; sc.e[m] = 0.0; }
                    TSF_WAVE_SYNC();
                    underflow = true;
                    break;
                }
                { const double ri = 1.0 / r; sn = f * ri; c = g * ri; }
                g = di1 - p;
                r = (di - g) * sn + 2.0 * c * b;
                p = sn * r;
                if (lane == 0) sc.d[i + 1] = g + p;
                g = c * r - b;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int k = lane + s * W;
                    if (k < n) {
                        double *vrow = Vm + k * PM;
                        const double v1 = vrow[i + 1], v0 = vrow[i];
                        vrow[i + 1] = __builtin_fma(sn, v0, c * v1);
                        vrow[i] = __builtin_fma(c, v0, -(sn * v1));
                    }
                }
                TSF_WAVE_SYNC();
            }
            if (underflow) continue;
            if (lane == 0) { sc.d[l] = sc.d[l] - p; sc.e[l] = g; sc.e[m] = 0.0; }
            TSF_WAVE_SYNC();
        }
    }
    TSF_WAVE_SYNC();
#pragma unroll
    for (int s = 0; s < 2; ++s) lam[s] = (lane + s * W < n) ? sc.d[lane + s * W] : 0.0;
}

template <int KP>
struct NewtonLds2 {
    double th[TSF_MAX_P + W];
    double ks[NTAB + 1], mc[NTAB + 1];
    double tp1[NTAB], tp2[NTAB];
    double tot1[W + 1], tot2[W + 1];
    double d1[NTAB + 1], d2[NTAB + 1], rb[NTAB + 1], ab[NTAB + 1];
    double accR[KP];
    QlScratch2 ql;
};

template <int KP>
constexpr size_t newton2_lds_bytes(int PM)
{
    return ((sizeof(NewtonLds2<KP>) + 15) & ~(size_t)15) + (size_t)PM * PM * sizeof(double);
}

// entry i (wave-uniform) of a two-slot vector as a scalar
__device__ __forceinline__ double entry2(const double (&v)[2], int i)
{
    return (i < W) ? readlane_f64(v[0], i) : readlane_f64(v[1], i - W);
}

template <int KP, int GROWTH, int MODE>
__global__ __launch_bounds__(64) void newton_kernel2(FitArgs a, int PM)
{
    constexpr int PPL = 2;
    extern __shared__ __align__(16) unsigned char smem[];
    NewtonLds2<KP> &lds = *reinterpret_cast<NewtonLds2<KP> *>(smem);
    double *Am = reinterpret_cast<double *>(smem + ((sizeof(NewtonLds2<KP>) + 15) & ~(size_t)15));
    double *Vm = Am;
    const int lane = threadIdx.x;
    const DevSpec *sp = a.sp;
    for (int64_t n = blockIdx.x; n < a.N; n += gridDim.x) {
    SeriesView sv;
    make_view<KP, PPL>(a, n, sv);
    for (int i = threadIdx.x; i < TSF_MAX_P + W; i += W) lds.th[i] = 0.0;
    TSF_WAVE_SYNC();
    const SeriesTab st = a.stab[n];
    if (lane == 0) {
        a.y_scale[n] = st.y_scale;
        if (!a.aligned || n == 0) a.grid_out[a.aligned ? 0 : n] = a.gtab[grid_index(a, n)].info;
    }
    double th[PPL], x[PPL], g[PPL], gx[PPL], step[PPL], acc[PPL];
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const int p = lane + s * W;
        th[s] = (p == 0) ? st.k0 : (p == 1 ? st.m0 : 0.0);
        x[s] = th[s]; g[s] = 0.0; gx[s] = 0.0; step[s] = 0.0; acc[s] = 0.0;
    }
    if (st.status0 != 0) {
        if (st.status0 == TSF_ST_CONSTANT && lane == 2) th[0] = -20.72326583694641;
        store_theta<PPL>(a, sv, n, th, a.theta);
        if (lane == 0) { a.status[n] = st.status0; a.n_iter[n] = 0; a.n_eval[n] = 0; a.fval[n] = 0.0; }
        continue;
    }
    const int P = sv.P;
    const double epsilon = 1e-3, half_epsilon = 0.5 * epsilon;
    enum { S_INIT = 0, S_F0, S_FD, S_HALVE };
    int stage = S_INIT, ret = TSF_ST_MAXIT, it = 0, mI = 0, d = 0, pi = 0;
    double lp = 0.0, lastlp = 0.0, f0 = 0.0, f1 = 0.0, size = 2.0, fx = 0.0;
    for (;;) {
        FT_DECL;
        const bool bad = eval_fg<KP, GROWTH, MODE, PPL, false, NewtonLds2<KP>>(sp, sv, lds, x, fx, gx FT_PASS);
        bool finish_iter = false, moved = false;
        if (stage == S_INIT) {
            if (bad) { ret = TSF_ST_INIT_NONFINITE; lp = -fx; break; }
            lp = -fx;
            stage = S_F0;
#pragma unroll
            for (int s = 0; s < PPL; ++s) x[s] = th[s];
            continue;
        }
        if (stage == S_F0) {
            if (bad) { ret = TSF_ST_NEWTON_FAIL; break; }
            lastlp = lp;
            f0 = -fx;
            d = 0; pi = 0;
#pragma unroll
            for (int s = 0; s < PPL; ++s) { g[s] = gx[s]; acc[s] = 0.0; x[s] = (lane + s * W == 0) ? th[s] + (-2 * epsilon) : th[s]; }
            stage = S_FD;
            continue;
        }
        if (stage == S_FD) {
            if (bad) { ret = TSF_ST_NEWTON_FAIL; break; }
            const double coef = (pi == 0) ? 1.0 / 12.0 : (pi == 1 ? -2.0 / 3.0 : (pi == 2 ? 2.0 / 3.0 : -1.0 / 12.0));
#pragma unroll
            for (int s = 0; s < PPL; ++s) acc[s] = __builtin_fma(half_epsilon * coef, -gx[s], acc[s]);
            if (++pi == 4) {
#pragma unroll
                for (int s = 0; s < PPL; ++s) { if (lane + s * W < P) Am[d * PM + lane + s * W] = acc[s]; acc[s] = 0.0; }
                pi = 0; ++d;
            }
            if (d < P) {
                const double pert = (pi == 0) ? -2 * epsilon : (pi == 1 ? -1 * epsilon : (pi == 2 ? epsilon : 2 * epsilon));
#pragma unroll
                for (int s = 0; s < PPL; ++s) x[s] = (lane + s * W == d) ? th[s] + pert : th[s];
                continue;
            }
            // ---- H = A + A^T (in place; entry b owns the pairs (r, b), r <= b)
            TSF_WAVE_SYNC();
            for (int r = 0; r < P; ++r) {
                double u[PPL], v[PPL];
#pragma unroll
                for (int s = 0; s < PPL; ++s) {
                    const int b = lane + s * W;
                    const bool mine = b < P && r <= b;
                    u[s] = mine ? Am[r * PM + b] : 0.0; v[s] = mine ? Am[b * PM + r] : 0.0;
                }
                TSF_WAVE_SYNC();
#pragma unroll
                for (int s = 0; s < PPL; ++s) {
                    const int b = lane + s * W;
                    if (b < P && r <= b) { const double h = u[s] + v[s]; Am[r * PM + b] = h; Am[b * PM + r] = h; }
                }
                TSF_WAVE_SYNC();
            }
            // ---- make_negative_definite_and_solve
            double lam[PPL], proj[PPL];
            ql2_tridiag_q(P, PM, Am, lds.ql);
            ql2_chain(P, PM, Vm, lds.ql, lam);
#pragma unroll
            for (int s = 0; s < PPL; ++s) {
                const int c = lane + s * W;
                double pa = 0.0;
                for (int i = 0; i < P; ++i) {
                    const double gi = -entry2(g, i);
                    const double vij = (c < P) ? Vm[i * PM + c] : 0.0;
                    pa = __builtin_fma(vij, gi, pa);
                }
                proj[s] = (c < P) ? -pa / __builtin_fabs(lam[s]) : 0.0;
            }
#pragma unroll
            for (int s = 0; s < PPL; ++s) {
                const int r = lane + s * W;
                double sa = 0.0;
                for (int j = 0; j < P; ++j) {
                    const double pjv = entry2(proj, j);
                    const double vij = (r < P) ? Vm[r * PM + j] : 0.0;
                    sa = __builtin_fma(vij, pjv, sa);
                }
                step[s] = (r < P) ? sa : 0.0;
                x[s] = th[s];
            }
            size = 2.0; f1 = -1e100;
            stage = S_HALVE;
        } else {   // S_HALVE: a trial point was evaluated
            f1 = bad ? -1e100 : -fx;
        }
        // ---- Stan's `while (f1 < f0)` step-halving loop
        if (f1 < f0) {
            size *= 0.5;
            if (size < 1e-50) { finish_iter = true; moved = false; }
            else {
#pragma unroll
                for (int s = 0; s < PPL; ++s) x[s] = th[s] - size * step[s];
                continue;
            }
        } else {
            finish_iter = true; moved = true;
        }
        if (finish_iter) {
            ++it;
            if (moved) {
#pragma unroll
                for (int s = 0; s < PPL; ++s) th[s] = x[s];
                lp = f1;
            } else lp = f0;
            if (mI > 0 && __builtin_fabs(lp - lastlp) < 1e-8) { ret = TSF_ST_NEWTON_CONVERGED; break; }
            if (++mI >= a.opt.max_iter) { ret = TSF_ST_MAXIT; break; }
            stage = S_F0;
#pragma unroll
            for (int s = 0; s < PPL; ++s) x[s] = th[s];
        }
    }
    store_theta<PPL>(a, sv, n, th, a.theta);
    if (lane == 0) { a.status[n] = ret; a.n_iter[n] = it; a.n_eval[n] = sv.n_eval; a.fval[n] = -lp; }
    TSF_WAVE_SYNC();
    }
}

}  // namespace tsf
