// tsf_map_quad.h -- converge = MAP for linear growth with additive seasonality (round 6; aligned and ragged panels, P <= 64):
// the maximum a posteriori estimate of prophet.stan's model DIRECTLY, without an L-BFGS trajectory before it.
//
// For this model the data enter the posterior through a quadratic form.  With u = (k, m, delta, beta), Z the design of
// the trend and the seasonal columns, M = Z^T Z (one matrix for the whole aligned panel: gram_build_kernel), c = Z^T y,
// w = sigma_obs^2 and T rows,
//
//     -log p(u, w | y) = T/2 log w + (u'Mu - 2 c'u + y'y) / (2 w) + 2 w + 1/2 u'Du + C sum |delta_j|
//
// (D: 1/25 for k and m, 1/sigma_beta^2 for the seasonal coefficients, 0 for delta; C = 1 / tau; the 2 w is the half-normal
// prior of sigma_obs -- cn_assemble_q's function, term by term).  Two blocks, each minimised exactly:
//   * w for fixed u:  4 w^2 + T w - SSE(u) = 0, the positive root in closed form;
//   * u for fixed w:  the L1-regularised quadratic programme  1/2 u'(M + wD)u - c'u + wC |delta|_1, by the classical
//     primal active-set method -- deltas held at zero form the working set; Newton's step on the free parameters is one
//     Cholesky solve (lane = row, the matrix in LDS); a step that would carry a delta through zero stops there and the
//     delta joins the working set; when the free parameters are stationary, the held delta whose multiplier violates
//     |g_j| <= wC the most is released into the orthant its gradient points to.
// Alternated until the pseudo-gradient of the whole function (map_kernel's KKT residual) is below map_tol; the sequence of
// w is a scalar fixed-point iteration, and every third round the quadratic programme is solved at its Aitken extrapolate
// instead (short histories couple w and u through the L1 weight wC: 15 rounds on average for 90-row series without it,
// 9 with it).  From
// fbprophet's initial values that takes 4-6 rounds and 6-11 Cholesky solves per series, where Stan's L-BFGS spends ~450
// evaluations to stop a median 1e-3 short of this point and map_kernel another ~340 to get there (prototype against
// oracle/true_map.py: the parameters agree to 1e-8, the function value to its last digits).
//
// The function is NOT convex in (u, w) together, and the alternation is run twice, from the two ends of sigma (below: "Two
// passes"); the result is the lower of the two local minima found -- for BASELINE's shapes there is only one.  Randomised
// runs against the continuation (tools/dev/route_stress.py, 231 898 series of 30-800 rows): the same objective in 99.85 %,
// this solver lower in 0.13 %, the continuation lower in 0.016 % (a third local minimum between the two).
//
// Not bit-pinned to a CPU twin (like map_kernel): the result is defined by the model; the GPU test compares its forecasts
// with the independent solver's at 1e-4 over the whole horizon.  One wavefront per series, parameter p in lane p (P <= 64).
#pragma once
#include "tsf_quad_kernels.h"

namespace tsf {

constexpr int MQ_MAX_OUTER = 120, MQ_MAX_INNER = 800, MQ_MAX_SOLVES = 3000;

__device__ __forceinline__ double mq_wave_max(double v)
{
#pragma unroll
    for (int off = 1; off < W; off <<= 1) v = __builtin_fmax(v, __shfl_xor(v, off, W));
    return v;
}
__device__ __forceinline__ double mq_wave_min(double v)
{
#pragma unroll
    for (int off = 1; off < W; off <<= 1) v = __builtin_fmin(v, __shfl_xor(v, off, W));
    return v;
}

// The Cholesky factor in two phases.  Rows are ordered [k, m, seasonal coefficients | deltas]: the leading block (nN rows)
// is free in every step of the active-set method and depends on w only, so its columns of L -- for ALL rows, the deltas'
// too -- are computed ONCE per round (mq_factor_leading); a step then only needs the columns of the deltas that are free
// (mq_solve_free: a held delta's column is written as zeros, its row is skipped) and the two substitutions.  A lane keeps
// the variable it always has; `row` is that variable's row in this order, mq_owner(j) the lane that holds row j.  Am holds
// A[j][i], i >= j, at Am[j * PM + i] (upper triangle with the diagonal, never overwritten) and L in the strictly lower
// triangle; inv_l: the reciprocal of the lane's diagonal entry of L.
__device__ __forceinline__ int mq_owner(int j, int nN, int S) { return j < 2 ? j : (j < nN ? 3 + S + (j - 2) : 3 + (j - nN)); }

__device__ __forceinline__ double mq_column(const double *Am, int PM, int row, int j)
{
    // A[j][row] - sum_{k < j} L[row][k] L[j][k]: four fma chains, their eight LDS reads in flight together
    double a0 = Am[j * PM + row], a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const double *li = Am + row * PM, *lj = Am + j * PM;
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
    return (a0 + a1) + (a2 + a3);
}

// columns 0 .. nN-1 of L, every row (`in`: the lane holds a variable at all).  False: a pivot was not positive.
__device__ __forceinline__ bool mq_factor_leading(int nN, int S, int PM, double *Am, bool in, int row, double &inv_l)
{
    const int lane = lane_id();
    for (int j = 0; j < nN; ++j) {
        double s = 0.0;
        if (in && row >= j) s = mq_column(Am, PM, row, j);
        const int oj = mq_owner(j, nN, S);
        const double d = readlane_f64(s, oj);
        if (!(d > 0.0)) return false;
        const double il = 1.0 / __builtin_sqrt(d);
        if (lane == oj) inv_l = il;
        if (in && row > j) Am[row * PM + j] = s * il;
        wave_sync();
    }
    return true;
}

// the columns of the free deltas, then L z = rhs and L^T x = z over the free rows (fr: the lane's variable is free; every
// variable of the leading block is).  False: a pivot was not positive.
__device__ __forceinline__ bool mq_solve_free(int nN, int S, int PM, double *Am, bool in, bool fr, unsigned long long fmask, int row,
                                              double &inv_l, double rhs, double &sol)
{
    const int lane = lane_id();
    const int n = nN + S;
    for (int j = nN; j < n; ++j) {
        const int oj = 3 + (j - nN);
        if (!((fmask >> oj) & 1ull)) {
            if (in && row > j) Am[row * PM + j] = 0.0;          // a held delta: no column (and, below, no row)
            wave_sync();
            continue;
        }
        double s = 0.0;
        if (fr && row >= j) s = mq_column(Am, PM, row, j);
        const double d = readlane_f64(s, oj);
        if (!(d > 0.0)) return false;
        const double il = 1.0 / __builtin_sqrt(d);
        if (lane == oj) inv_l = il;
        if (fr && row > j) Am[row * PM + j] = s * il;
        wave_sync();
    }
    wave_sync();
    double r = fr ? rhs : 0.0, z = 0.0;
    for (int j = 0; j < n; ++j) {
        const int oj = mq_owner(j, nN, S);
        if (!((fmask >> oj) & 1ull)) continue;
        const double zj = readlane_f64(r, oj) * readlane_f64(inv_l, oj);
        if (lane == oj) z = zj;
        if (fr && row > j) r = __builtin_fma(-Am[row * PM + j], zj, r);
    }
    double r2 = z, st = 0.0;
    for (int j = n - 1; j >= 0; --j) {
        const int oj = mq_owner(j, nN, S);
        if (!((fmask >> oj) & 1ull)) continue;
        const double sj = readlane_f64(r2, oj) * readlane_f64(inv_l, oj);
        if (lane == oj) st = sj;
        if (fr && row < j) r2 = __builtin_fma(-Am[j * PM + row], sj, r2);
    }
    sol = fr ? st : 0.0;
    wave_sync();
    return true;
}

template <int KP>
constexpr size_t map_quad_lds_bytes(int PM)
{
    return quad_lanec_bytes<1>() + ((sizeof(QuadLds<KP, 1>) + 15) & ~(size_t)15) + sizeof(double) * (size_t)PM * PM;
}

// RAGGED: every series has its own rows, hence its own M -- the Gram matrix of its calendar where calendars are shared
// (QuadArgs::Mpre, built ahead by gram_grids_kernel), else built here column by column into the workgroup's slot of
// global memory (QuadArgs::Mslot; lane p writes and later reads its own column entries only).
template <int KP, int NTR, bool RAGGED = false>
__global__ __launch_bounds__(64) void map_quad_kernel(QuadArgs qa)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const FitArgs &a = qa.f;
    const int lane = lane_id();
    double *lanec = reinterpret_cast<double *>(smem);
    QuadLds<KP, 1> &wl = *reinterpret_cast<QuadLds<KP, 1> *>(smem + quad_lanec_bytes<1>());
    double *Am = reinterpret_cast<double *>(smem + quad_lanec_bytes<1>() + ((sizeof(QuadLds<KP, 1>) + 15) & ~(size_t)15));
    const double *Mg = qa.Mg;
    double *const Mown = RAGGED ? qa.Mslot + (size_t)blockIdx.x * qa.P4 * W : nullptr;
    for (int i = lane; i < 2 * W; i += W) wl.th[i] = 0.0;
    wave_sync();
    double *rb = qa.rbuf + (size_t)blockIdx.x * a.NTmax * W;
    const int max_outer = a.map_max_iter > 0 && a.map_max_iter < MQ_MAX_OUTER ? a.map_max_iter : MQ_MAX_OUTER;
    const double tol = a.map_tol > 0.0 ? a.map_tol : 1e-7;
    for (int64_t n = blockIdx.x; n < a.N; n += gridDim.x) {
        SeriesView sv;
        make_view_q<KP, 1>(a, n, sv);
        const SeriesTab st = a.stab[n];
        if (lane == 0) {
            a.y_scale[n] = st.y_scale;
            if (!a.aligned || n == 0) a.grid_out[a.aligned ? 0 : n] = a.gtab[grid_index(a, n)].info;
        }
        double x[1];
        x[0] = (lane == 0) ? st.k0 : (lane == 1 ? st.m0 : 0.0);
        if (st.status0 != 0) {
            // fbprophet raises (too few rows) or skips optimisation (constant y): as every fit kernel reports it
            if (st.status0 == TSF_ST_CONSTANT && lane == 2) x[0] = -20.72326583694641;
            store_theta<1>(a, sv, n, x, a.theta);
            if (lane == 0) { a.status[n] = st.status0; a.n_iter[n] = 0; a.n_eval[n] = 0; a.fval[n] = 0.0; }
            continue;
        }
        LaneConst<1> lk;
        quad_const_table(lanec + 3 * W, a.opt, qa.recenter_ratio);
        lane_consts<1>(a.sp, sv, lanec, lk);
        lk.ct = lanec + 3 * W;
        if constexpr (RAGGED) {
            if (qa.Mpre) {
                Mg = qa.Mpre + (size_t)grid_index(a, n) * qa.P4 * W;
            } else {
                for (int q = 0; q < qa.P4; ++q) {
                    double col[1] = {0.0};
                    if (q != 2 && q < sv.P) gram_column<KP, 1>(sv, wl, rb, q, col);
                    Mown[(size_t)q * W + lane] = col[0];
                }
                wave_sync();
                Mg = Mown;
            }
        }
        // c = Z^T y and y'y: the residual pass at u = 0
        double zero[1] = {0.0}, g0[1], f0, yy, cv[1];
        resid_eval_q<KP, 1, NTR>(sv, wl, lk, rb, zero, f0, g0, yy, cv);
        const int P = sv.P, S = sv.S, nN = P - 1 - S, PM = (P - 1) | 1;
        const bool par = lane < P && lane != 2;
        const bool isD = lane >= 3 && lane < 3 + S;
        const int row = lane < 2 ? lane : (isD ? nN + (lane - 3) : 2 + (lane - 3 - S));      // the order of the factor: [k, m, beta | delta]
        const double c = par ? cv[0] : 0.0;
        const double Dl = par ? lk.lc[lane] : 0.0;
        const double C = lk.inv_tau, Tn = (double)sv.T;
        const double cmax = mq_wave_max(__builtin_fabs(c));
        auto matvec = [&](double uu) -> double {
            double acc = 0.0;
            for (int q = 0; q < P; ++q) acc = __builtin_fma(Mg[(size_t)q * W + lane], readlane_f64(uu, q), acc);
            return par ? acc : 0.0;
        };
        // Two passes.  The function is not convex in (u, w) together -- T/2 log w is concave -- and histories of a few hundred
        // noisy rows do have two local minima: one with a large sigma and hardly a changepoint, one with a small sigma and
        // many.  The map w -> w*(SSE(u*(w))) the alternation iterates is monotone, so started from ABOVE (pass 0: every delta
        // held at zero, the largest SSE) it ends at the largest stable fixed point, started from BELOW (pass 1: the
        // unpenalised least-squares fit first, the smallest SSE) at the smallest; where the two differ, the lower function
        // value is taken (12 % of 60-500-row weekly panels in the prototype, neither start the better one as a rule; none of
        // the 10 000 cfg2 series).  Pass 1 stops as soon as it has arrived at pass 0's point.
        double best_u = 0.0, best_w = 1.0, best_F = 0.0, w_above = 0.0;
        unsigned long long held_above = 0ull;
        int best_status = TSF_ST_MAP_MAXIT, n_outer = 0, n_solve = 0;
        for (int pass = 0; pass < 2; ++pass) {
        double u = par ? x[0] : 0.0;               // fbprophet's initial k and m, every delta at zero
        double w = 1.0, sse = yy, mu = 0.0;
        int status = TSF_ST_MAP_MAXIT;
        bool same_as_above = false;
        double w1 = 0.0, w2 = 0.0;                  // w of the last and of the last but one round
        unsigned long long zmask_prev = ~0ull;
        for (int outer = 0; ; ++outer) {
            mu = matvec(u);
            sse = yy + bfly_sum(u * (mu - 2.0 * c));
            if (!(sse > 1e-300)) sse = 1e-300;
            w = (__builtin_sqrt(__builtin_fma(Tn, Tn, 16.0 * sse)) - Tn) * 0.125;
            if (!(w > 1e-300)) w = 1e-300;
            // the pseudo-gradient of the whole function at (u, w); its log-sigma entry is zero by the closed form
            const double g = par ? ((mu - c) / w + Dl * u) : 0.0;
            double pg = g;
            if (isD) {
                if (u > 0.0) pg = g + C;
                else if (u < 0.0) pg = g - C;
                else if (g + C < 0.0) pg = g + C;
                else if (g - C > 0.0) pg = g - C;
                else pg = 0.0;
            }
            const double kkt = mq_wave_max(__builtin_fabs(pg));
            if (pass == 1 && outer >= 1 && w >= w_above * (1.0 - 1e-4) && __ballot(isD && u == 0.0) == held_above) {
                // the iterates from below are below their limit, the limit is at most pass 0's: within 1e-4 of it with the same
                // deltas at zero this IS pass 0's minimum (or one whose function value differs in the tenth digit)
                same_as_above = true;
                break;
            }
            if (!(kkt > tol) && !(pass == 1 && outer == 0)) { status = TSF_ST_MAP_KKT; break; }
            {
                // nothing moves any more -- w to its last digits, the same deltas at zero -- but the residual of the test above is
                // rounding noise that w amplifies (an almost noiseless history: SSE is a difference of sums 1e8 times its size):
                // the function value has converged
                const unsigned long long zmask = __ballot(isD && u == 0.0);
                if (outer >= 2 && __builtin_fabs(w - w1) <= 1e-13 * w && zmask == zmask_prev) { status = TSF_ST_MAP_FTOL; break; }
                zmask_prev = zmask;
            }
            if (outer >= max_outer) break;
            n_outer++;
            // Aitken's extrapolate of the fixed-point sequence w_k, every third round (wq: the w the programme is solved at)
            double wq = w;
            if (outer >= 2 && outer % 3 == 2) {
                // (only where the last two differences show a geometric sequence: same sign, shrinking)
                const double d1 = w - w1, d0 = w1 - w2, d2 = d1 - d0;
                const double rho = d1 / d0;
                const double wa = w - d1 * d1 / d2;
                if (rho > 0.0 && rho < 0.95 && wa > 0.0 && __builtin_fabs(wa - w) < 0.5 * w) wq = wa;
            }
            w2 = w1; w1 = w;
            // ---- u at fixed w: primal active-set method on 1/2 u'(M + wD)u - c'u + wC |delta|_1
            const double Cw = wq * C;
            const double tolq = __builtin_fmax(0.05 * tol * wq, 4e-14 * __builtin_fmax(1.0, cmax));
            const bool ls_round = pass == 1 && outer == 0;       // from below: the unpenalised least-squares fit, every delta free
            bool held = isD && u == 0.0 && !ls_round;            // the working set: deltas held at zero
            double zs = isD ? (double)((u > 0.0) - (u < 0.0)) : 0.0;
            bool gave_up = false;
            int n_newton = 0;
            // A = M + wD (upper triangle, the factor's row order) and the leading columns of its factor: once per round
            bool tabu = false;                     // released in this round and pushed straight back to zero: not released again
            double ridge = 0.0, inv_l = 1.0;
            bool lead = false;
            for (int attempt = 0; attempt < 8 && !lead; ++attempt) {
                for (int j = 0; j < P - 1; ++j) {
                    const int q = mq_owner(j, nN, S);
                    if (par && row >= j) {
                        double v = Mg[(size_t)q * W + lane];
                        if (row == j) v = v + wq * Dl + ridge;
                        Am[j * PM + row] = v;
                    }
                }
                wave_sync();
                lead = mq_factor_leading(nN, S, PM, Am, par, row, inv_l);
                if (!lead) ridge = ridge == 0.0 ? 1e-12 * __builtin_fmax(1.0, mq_wave_max(par ? Mg[(size_t)lane * W + lane] : 0.0)) : ridge * 100.0;
            }
            if (!lead) { status = TSF_ST_MAP_LS; break; }
            for (int inner = 0; inner < MQ_MAX_INNER; ++inner) {
                if (inner > 0) mu = matvec(u);
                const double gq = par ? (__builtin_fma(wq * Dl, u, mu) - c + Cw * zs) : 0.0;
                const bool fr = par && !held;
                const double gF = mq_wave_max(fr ? __builtin_fabs(gq) : 0.0);
                if (ls_round && inner > 0) break;
                if (!(gF > tolq)) {
                    // stationary on the free set: the multipliers of the held deltas
                    const double viol = (held && !tabu) ? __builtin_fabs(gq) - Cw : -1.0;
                    const double vmax = mq_wave_max(viol);
                    if (!(vmax > tolq)) break;
                    const unsigned long long who = __ballot(held && viol == vmax);
                    const int j = __builtin_ctzll(who);
                    if (lane == j) { held = false; zs = gq > 0.0 ? -1.0 : 1.0; }
                    continue;
                }
                // Newton's step on the free set: (M + wD)_FF d = -g_F
                const unsigned long long fmask = __ballot(fr);
                double d = 0.0;
                const bool ok = mq_solve_free(nN, S, PM, Am, par, fr, fmask, row, inv_l, -gq, d);
                n_solve++;
                if (!ok) { gave_up = true; break; }
                const double ustar = u + d;
                if (ls_round) { if (fr) u = ustar; continue; }
                // a delta may not change sign inside a step: stop at the first one that reaches zero
                const bool wrong = fr && isD && ustar * zs < 0.0;
                if (pass == 1 && n_newton++ < 3 && __ballot(wrong) != 0ull) {
                    // from below, the first steps of a round: EVERY delta that would change sign goes to zero at once (the
                    // least-squares start has all 25 of them nonzero; one at a time is a Cholesky solve each) -- the
                    // multiplier test releases what was held wrongly
                    if (fr) u = ustar;
                    if (wrong) { u = 0.0; held = true; zs = 0.0; }
                    continue;
                }
                const double aj = wrong ? (u != 0.0 ? u / (u - ustar) : 0.0) : 2.0;
                const double amin = mq_wave_min(aj);
                if (amin >= 1.0) {
                    if (fr) u = ustar;
                } else {
                    if (fr) u = __builtin_fma(amin, d, u);
                    const unsigned long long who = __ballot(wrong && aj == amin);
                    const int j = __builtin_ctzll(who);
                    if (lane == j) { if (u == 0.0) tabu = true; u = 0.0; held = true; zs = 0.0; }
                }
                if (n_solve > MQ_MAX_SOLVES) { gave_up = true; break; }
            }
            if (gave_up) { status = n_solve > MQ_MAX_SOLVES ? TSF_ST_MAP_MAXIT : TSF_ST_MAP_LS; break; }
        }
        if (same_as_above) break;
        // the function value of the pass's point (cn_assemble_q's terms)
        if (status != TSF_ST_MAP_KKT) {             // (left inside a round: the sums of the point it was left at)
            mu = matvec(u);
            sse = yy + bfly_sum(u * (mu - 2.0 * c));
            if (!(sse > 1e-300)) sse = 1e-300;
            w = (__builtin_sqrt(__builtin_fma(Tn, Tn, 16.0 * sse)) - Tn) * 0.125;
            if (!(w > 1e-300)) w = 1e-300;
        }
        x[0] = par ? u : 0.0;
        if (lane == 2) x[0] = 0.5 * dm_log(w);
        double ztr[1] = {par ? c - mu : 0.0}, gf[1], fv;
        const bool badf = assemble_q<1>(sv, lk, x, sse, ztr, fv, gf);
        if (pass == 0) { w_above = w; held_above = __ballot(isD && u == 0.0); }
        if (pass == 0 || (!badf && status != TSF_ST_MAP_LS && fv < best_F)) { best_u = u; best_w = w; best_F = fv; best_status = status; }
        if (n_solve > MQ_MAX_SOLVES) break;
        }
        // theta, the function value and the counts of both passes
        x[0] = par ? best_u : 0.0;
        if (lane == 2) x[0] = 0.5 * dm_log(best_w);
        store_theta<1>(a, sv, n, x, a.theta);
        if (lane == 0) { a.fval[n] = best_F; a.status[n] = best_status; a.n_iter[n] = n_outer; a.n_eval[n] = n_solve + 1; }
        wave_sync();
    }
}

}  // namespace tsf
