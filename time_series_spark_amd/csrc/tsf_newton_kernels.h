// tsf_newton_kernels.h -- Stan's Newton optimiser, one wavefront per series.
//
// fbprophet 0.5 runs optimizing(algorithm='Newton') when a history has fewer than 100 rows and
// as the retry after an L-BFGS RuntimeError (UPSTREAM-RECALL forecaster.py fit; SURVEY.md 8a
// U9); the reference reaches it through Prophet().fit(pdf) at
// /root/reference/src/jobs/prophet_modeler.py:65-66.  The algorithm (stan 2.19
// model/grad_hess_log_prob.hpp, optimization/newton.hpp, services/optimize/newton.hpp) and its
// canonical operation order are stated in oracle/prophet_canon.c (cn_newton, cn_jacobi); this
// file executes exactly that sequence:
//
//   lane p                = parameter p (P <= 64: Newton is for short series, K is small)
//   gradient evaluations  = eval_fg (residual form), 4 per parameter for the finite-difference
//                           Hessian, the perturbed coordinate selected by lane
//   A[d][p]               = fma chain over the 4 perturbations, lane p, written to LDS row d
//   H = A + A^T           in place, pair (a, b) handled by lane b
//   eigen-decomposition   = round-robin Jacobi in LDS: per round n/2 disjoint rotations, angles
//                           computed by the lanes of each pair, A <- A J and V <- V J row by row
//                           (lane = column), A <- J^T A pair of rows by pair of rows
//   proj, step            = lane-parallel fma chains with the other operand broadcast by readlane
//   step halving          = Stan's loop, one evaluation per trial
//
// Cost per Newton iteration: 4 P + 1 evaluations, one P x P eigen-decomposition, ~30 trial
// evaluations.
#pragma once
#include "tsf_fit_kernels.h"

namespace tsf {

// Symmetric eigen-decomposition of the n x n matrix in Am (row stride PM, destroyed: its
// diagonal ends as the eigenvalues), eigenvectors to the columns of Vm.  One wave; lane j owns
// column j in the row passes.  Returns the eigenvalue of lane j (0 for j >= n).
__device__ __forceinline__ double jacobi_lds(int n, int PM, double *Am, double *Vm)
{
    const int lane = lane_id();
    const int m = n + (n & 1);
    const bool live = lane < n;
    for (int i = 0; i < n; ++i)
        if (live) Vm[i * PM + lane] = (i == lane) ? 1.0 : 0.0;
    TSF_WAVE_SYNC();
    for (int sweep = 0; sweep < 30; ++sweep) {
        double so = 0.0, sd = 0.0;
        if (live) {
            for (int i = 0; i < n; ++i) {
                const double v = Am[i * PM + lane];
                if (i == lane) sd = v * v;
                else so = __builtin_fma(v, v, so);
            }
        }
        const double off2 = bfly_sum(so), dia2 = bfly_sum(sd);
        if (off2 <= 1e-26 * dia2) break;
        for (int r = 0; r < m - 1; ++r) {
            // rotation of the pair this lane's index belongs to
            int q;
            if (lane == m - 1) q = r;
            else if (lane == r) q = m - 1;
            else { q = (2 * r - lane) % (m - 1); if (q < 0) q += m - 1; }
            double c = 1.0, kap = 0.0;
            if (live && q < n) {
                const int lo = lane < q ? lane : q, hi = lane < q ? q : lane;
                const double apq = Am[lo * PM + hi];
                if (apq != 0.0) {
                    const double tau = (Am[hi * PM + hi] - Am[lo * PM + lo]) / (2.0 * apq);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (__builtin_fabs(tau) + __builtin_sqrt(1.0 + tau * tau));
                    c = 1.0 / __builtin_sqrt(1.0 + t * t);
                    const double s = t * c;
                    kap = (lane == lo) ? -s : s;
                }
            }
            if (!live) q = lane;
            TSF_WAVE_SYNC();
            // A <- A J, V <- V J: column `lane` mixes with column q, row by row (in place: both
            // operands are read before either result is written)
            const bool mix = live && q < n;
            for (int rr = 0; rr < n; ++rr) {
                double ai = 0.0, aq = 0.0, vi = 0.0, vq = 0.0;
                if (live) { ai = Am[rr * PM + lane]; vi = Vm[rr * PM + lane]; }
                if (mix) { aq = Am[rr * PM + q]; vq = Vm[rr * PM + q]; }
                TSF_WAVE_SYNC();
                if (live) {
                    Am[rr * PM + lane] = __builtin_fma(aq, kap, ai * c);
                    Vm[rr * PM + lane] = __builtin_fma(vq, kap, vi * c);
                }
                TSF_WAVE_SYNC();
            }
            // A <- J^T A: rows i and q(i) mix; lane = column
            for (int i = 0; i < n; ++i) {
                const int qi = __builtin_amdgcn_readlane(q, i);
                if (qi >= n || qi < i) continue;          // dummy partner: unchanged; pair done at its lower index
                const double ci = readlane_f64(c, i), ki = readlane_f64(kap, i);
                const double cq = readlane_f64(c, qi), kq = readlane_f64(kap, qi);
                double bi = 0.0, bq = 0.0;
                if (live) { bi = Am[i * PM + lane]; bq = Am[qi * PM + lane]; }
                TSF_WAVE_SYNC();
                if (live) {
                    Am[i * PM + lane] = __builtin_fma(ki, bq, ci * bi);
                    Am[qi * PM + lane] = __builtin_fma(kq, bi, cq * bq);
                }
                TSF_WAVE_SYNC();
            }
        }
    }
    TSF_WAVE_SYNC();
    return live ? Am[lane * PM + lane] : 0.0;
}

// LDS after WaveLds: Am [PM][PM], Vm [PM][PM] doubles
template <int KP>
constexpr size_t newton_lds_bytes(int PM)
{
    return ((sizeof(WaveLds<KP, 1>) + 15) & ~(size_t)15) + 2 * (size_t)PM * PM * sizeof(double);
}

template <int KP, int GROWTH, int MODE>
__global__ __launch_bounds__(64) void newton_kernel(FitArgs a, int PM)
{
    constexpr int PPL = 1;
    extern __shared__ __align__(16) unsigned char smem[];
    WaveLds<KP, PPL> &lds = *reinterpret_cast<WaveLds<KP, PPL> *>(smem);
    double *Am = reinterpret_cast<double *>(smem + ((sizeof(WaveLds<KP, PPL>) + 15) & ~(size_t)15));
    double *Vm = Am + (size_t)PM * PM;
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
        if (!a.aligned || n == 0) a.grid_out[a.aligned ? 0 : n] = a.gtab[a.aligned ? 0 : n].info;
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
        const bool bad = eval_fg<KP, GROWTH, MODE, PPL>(sp, sv, lds, x, fx, gx);
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
            // ---- make_negative_definite_and_solve
            const double lam = jacobi_lds(P, PM, Am, Vm);
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

}  // namespace tsf
