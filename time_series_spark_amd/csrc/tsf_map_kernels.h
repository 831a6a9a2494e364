// tsf_map_kernels.h -- the converged-MAP option (tsf_spec.converge = TSF_CONVERGE_MAP, round 6).
//
// Stan's L-BFGS, at Stan's tolerances, stops on the kinks of the Laplace prior sum |delta_j| / tau, a median 1e-3
// (linear / additive models) to 1e-2 (the reference's own model) away from the optimum in forecast, so
// what `Prophet.fit` returns (/root/reference/src/jobs/prophet_modeler.py:65-66) is a property of a floating-point
// trajectory.  With converge = MAP the fit goes on from where Stan's rule stopped it until it IS the maximum a
// posteriori estimate of prophet.stan's model: forecasts that are a property of the model (run-to-run and
// implementation-to-implementation differences of 1e-7 instead of 1e-3), checked on the GPU against an independent solver
// (oracle/true_map.py: delta split into positive and negative parts, scipy's L-BFGS-B) at the north star's 1e-4.
//
// The method.  F(x) = f(x) + C sum_{j in D} |x_j|, f smooth, D = the changepoint parameters, C = 1 / tau.  Inside one
// orthant of the delta the function is smooth; what a quasi-Newton method must not do is carry curvature pairs across
// the kinks or step through them.  One iteration:
//   * pseudo-gradient pg (Andrew & Gao's orthant-wise definition: the one-sided derivative that descends, 0 where
//     |df/dx_j| <= C at x_j = 0);  max |pg| is the KKT residual, 0 exactly at the optimum;
//   * FREE set = every parameter outside D, every nonzero delta, every zero delta whose pseudo-gradient is not 0.  The
//     L-BFGS two-loop recursion runs on the history pairs MASKED to the free set (pairs whose masked curvature is not
//     positive are skipped): a Newton-like step in the subspace that can move -- plain OWL-QN, which runs the recursion on
//     the full vectors and projects afterwards, needs 10-50 x the iterations on these posteriors (measured on the
//     prototype: 750-20 000 against 100-500);
//   * a delta at zero only moves into the orthant its pseudo-gradient points to; the trial point is projected back onto
//     the orthant of the current point (a delta that would change sign lands on 0 and joins the active set);
//   * backtracking line search on F with the Armijo rule on the pseudo-gradient.
// Ends at max |pg| <= tol (TSF_ST_MAP_KKT), or when 20 iterations together gained less than 1e-13 |F| (TSF_ST_MAP_FTOL:
// the function value has converged to the last bits; the KKT residual is then typically 1e-5 .. 1e-4 on gradients of
// 1e+3 at the start), or at map_max_iter (TSF_ST_MAP_MAXIT).  Evaluations are the residual-form evaluator of every
// model (eval_fg on the design tables): one wavefront per series, parameter p in lane p mod 64.
//
// Not bit-pinned to a CPU twin (unlike the Stan-rule fit): the result is defined by the model, and the test is the
// distance to the independent solver's optimum.
#pragma once
#include "tsf_fit_kernels.h"

namespace tsf {

constexpr int MAP_M = 24;               // curvature pairs kept (the valleys are flat: condition numbers of 1e8 and more)
constexpr int MAP_FWIN = 20;            // iterations over which the function value must still move

template <int PPL>
__device__ __forceinline__ double map_dot(const double (&a)[PPL], const double (&b)[PPL])
{
    double t = 0.0;
#pragma unroll
    for (int s = 0; s < PPL; ++s) t = __builtin_fma(a[s], b[s], t);
    return bfly_sum(t);
}

__device__ __forceinline__ double map_wave_max(double v)
{
#pragma unroll
    for (int off = 1; off < W; off <<= 1) v = __builtin_fmax(v, __shfl_xor(v, off, W));
    return v;
}

template <int KP, int PPL>
__host__ __device__ inline size_t map_lds_bytes()
{
    return ((wave_lds_bytes<KP, PPL>(1) + 15) & ~(size_t)15) + sizeof(double) * (2 * MAP_M * PPL * W + MAP_FWIN + 4);
}

// HARM != 0: the evaluator that reads a row as its base pairs and expands the Fourier columns in registers (eval_fg HARM:
// 168 registers, three waves per SIMD) -- the models whose harmonic structure has a compiled expansion, where the call
// built the base-pair table; else the table-streaming evaluator (one wave per SIMD).
template <int KP, int GROWTH, int MODE, int PPL, bool XIDX, int HARM = 0>
__global__ __launch_bounds__(64, HARM != 0 ? TSF_HARM_WPS : 1) void map_kernel(FitArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    WaveLds<KP, PPL> &lds = *reinterpret_cast<WaveLds<KP, PPL> *>(smem);
    double *Hs = reinterpret_cast<double *>(smem + ((wave_lds_bytes<KP, PPL>(1) + 15) & ~(size_t)15));   // s: [MAP_M][PPL][64]
    double *Hy = Hs + MAP_M * PPL * W;                                                                    // y: the same
    double *Fwin = Hy + MAP_M * PPL * W;                                                                  // F of the last MAP_FWIN iterates
    if ((int64_t)blockIdx.x >= a.N) return;
    const int64_t n = (int64_t)blockIdx.x;
    const int st0 = a.status[n];
    if (st0 < 0 || st0 == TSF_ST_CONSTANT) return;      // no model / fbprophet's constant-history shortcut: nothing to converge
    const int lane = lane_id();
    SeriesView sv;
    make_view<KP, PPL>(a, n, sv);
    if constexpr (HARM != 0) {
        const int kd = a.sp->K < KP ? a.sp->K : KP;         // dense columns of the model
        sv.n_xd = kd - harm_kf(HARM);
    }
    if (sv.T < 2) return;
    for (int i = threadIdx.x; i < TSF_MAX_P + W; i += W) lds.th[i] = 0.0;
    TSF_WAVE_SYNC();
    const double C = 1.0 / sv.tau;
    const int S = sv.S;
    bool isD[PPL], live[PPL];
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const int p = lane + s * W;
        isD[s] = p >= 3 && p < 3 + S;
        live[s] = p < sv.P;
    }
    double x[PPL], v[PPL], g[PPL], F;
    load_theta<PPL>(a, sv, n, a.theta, x);
    FT_DECL;
    // smooth part of the gradient: the evaluator's gradient without the Laplace term's sign(delta) / tau
    auto smooth = [&](const double (&xx)[PPL], double &Fo, double (&vo)[PPL]) -> bool {
        const bool bad = eval_fg<KP, GROWTH, MODE, PPL, XIDX, WaveLds<KP, PPL>, 0, false, HARM>(a.sp, sv, lds, xx, Fo, g FT_PASS);
#pragma unroll
        for (int s = 0; s < PPL; ++s) {
            double gv = live[s] ? g[s] : 0.0;
            if (isD[s]) gv = gv - C * (double)((xx[s] > 0.0) - (xx[s] < 0.0));
            vo[s] = gv;
        }
        return bad || !finite_f64(Fo);
    };
    auto pseudo = [&](const double (&xx)[PPL], const double (&vv)[PPL], double (&pg)[PPL]) {
#pragma unroll
        for (int s = 0; s < PPL; ++s) {
            double p_ = vv[s];
            if (isD[s]) {
                if (xx[s] > 0.0) p_ = vv[s] + C;
                else if (xx[s] < 0.0) p_ = vv[s] - C;
                else if (vv[s] + C < 0.0) p_ = vv[s] + C;
                else if (vv[s] - C > 0.0) p_ = vv[s] - C;
                else p_ = 0.0;
            }
            pg[s] = p_;
        }
    };
    int n_it = 0, n_ev = 0, status = TSF_ST_MAP_MAXIT;
    if (smooth(x, F, v)) { return; }            // the stopped fit's own point does not evaluate: leave the Stan-rule result
    n_ev++;
    int hcount = 0, hhead = 0;                   // pairs held; slot the next pair goes to
    int win0 = 0;                                // iteration at which the current stall window started
    bool stalled = false;
    const int max_iter = a.map_max_iter > 0 ? a.map_max_iter : 10000;
    const double tol = a.map_tol > 0.0 ? a.map_tol : 1e-7;
    double pg[PPL], q[PPL], d[PPL], xn[PPL], vn[PPL], fm[PPL];
    for (int it = 0; it < max_iter; ++it) {
        pseudo(x, v, pg);
        double mx = 0.0;
#pragma unroll
        for (int s = 0; s < PPL; ++s) mx = __builtin_fmax(mx, __builtin_fabs(pg[s]));
        const double kkt = map_wave_max(mx);
        if (!(kkt > tol)) { status = TSF_ST_MAP_KKT; break; }
        if (it - win0 >= MAP_FWIN) {
            const double Fold = Fwin[it % MAP_FWIN];            // F of MAP_FWIN iterations ago
            if (Fold - F <= 1e-13 * __builtin_fmax(1.0, __builtin_fabs(F))) {
                // stalled: once more from here with an empty memory (the pairs may describe another face of the orthant
                // structure than the one the iterate has settled on); a second stall in a row ends the fit
                if (stalled) { status = TSF_ST_MAP_FTOL; break; }
                stalled = true; hcount = 0; win0 = it;
            } else {
                stalled = false;
            }
        }
        TSF_WAVE_SYNC();
        Fwin[it % MAP_FWIN] = F;                 // (every lane writes the same value)
        n_it++;
        // ---- direction: two-loop recursion on the pairs masked to the free set
#pragma unroll
        for (int s = 0; s < PPL; ++s) {
            fm[s] = (live[s] && (!isD[s] || x[s] != 0.0 || pg[s] != 0.0)) ? 1.0 : 0.0;
            q[s] = pg[s];
        }
        // (loops over MAP_M written out with `k < hcount` guards: alpha_i / ys_i / used stay wave-uniform registers)
        double alpha_i[MAP_M], ys_i[MAP_M];
        bool used[MAP_M];
        bool any_used = false;
        double gam = 1.0;
#pragma unroll
        for (int k = 0; k < MAP_M; ++k) {        // newest first
            used[k] = false; alpha_i[k] = 0.0; ys_i[k] = 1.0;
            if (k < hcount) {
                const int slot = (hhead - 1 - k + 2 * MAP_M) % MAP_M;
                double sv_[PPL], yv_[PPL];
#pragma unroll
                for (int s = 0; s < PPL; ++s) { sv_[s] = fm[s] * Hs[(slot * PPL + s) * W + lane]; yv_[s] = fm[s] * Hy[(slot * PPL + s) * W + lane]; }
                double tys = 0.0, tyy = 0.0, tss = 0.0, tsq = 0.0;
#pragma unroll
                for (int s = 0; s < PPL; ++s) {
                    tys = __builtin_fma(yv_[s], sv_[s], tys); tyy = __builtin_fma(yv_[s], yv_[s], tyy);
                    tss = __builtin_fma(sv_[s], sv_[s], tss); tsq = __builtin_fma(sv_[s], q[s], tsq);
                }
                double ys, yy, ss, sq;
                bfly_sum4(tys, tyy, tss, tsq, ys, yy, ss, sq);
                if (ys > 1e-12 * __builtin_sqrt(yy * ss)) {
                    used[k] = true;
                    ys_i[k] = ys;
                    const double al = sq / ys;
                    alpha_i[k] = al;
#pragma unroll
                    for (int s = 0; s < PPL; ++s) q[s] = __builtin_fma(-al, yv_[s], q[s]);
                    if (!any_used) { any_used = true; gam = ys / yy; }      // initial inverse Hessian: (s.y) / (y.y) of the newest usable pair
                }
            }
        }
        if (any_used) {
#pragma unroll
            for (int s = 0; s < PPL; ++s) q[s] = q[s] * gam;
        }
#pragma unroll
        for (int kk = 0; kk < MAP_M; ++kk) {     // oldest first
            const int k = MAP_M - 1 - kk;
            if (k < hcount && used[k]) {
                const int slot = (hhead - 1 - k + 2 * MAP_M) % MAP_M;
                double sv_[PPL], yv_[PPL];
#pragma unroll
                for (int s = 0; s < PPL; ++s) { sv_[s] = fm[s] * Hs[(slot * PPL + s) * W + lane]; yv_[s] = fm[s] * Hy[(slot * PPL + s) * W + lane]; }
                const double b = map_dot<PPL>(yv_, q) / ys_i[k];
                const double c = alpha_i[k] - b;
#pragma unroll
                for (int s = 0; s < PPL; ++s) q[s] = __builtin_fma(c, sv_[s], q[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < PPL; ++s) {
            double dv = -q[s] * fm[s];
            if (isD[s] && x[s] == 0.0 && !(dv * (-pg[s]) > 0.0)) dv = 0.0;     // a delta at zero only moves where its pseudo-gradient points
            d[s] = dv;
        }
        double dg = map_dot<PPL>(d, pg);
        bool steepest = !any_used;
        if (!(dg < 0.0)) {                       // not a descent direction: steepest descent on the pseudo-gradient, empty memory
#pragma unroll
            for (int s = 0; s < PPL; ++s) d[s] = -pg[s];
            hcount = 0;
            steepest = true;
        }
        double alpha = 1.0;
        if (steepest) {
            const double nrm = __builtin_sqrt(map_dot<PPL>(pg, pg));
            alpha = 1.0 / __builtin_fmax(nrm, 1e-300);
        }
        // ---- projected backtracking line search
        bool ok = false;
        double Fn = F;
        for (int ls = 0; ls < 60; ++ls) {
#pragma unroll
            for (int s = 0; s < PPL; ++s) {
                double t = __builtin_fma(alpha, d[s], x[s]);
                if (isD[s]) {
                    const double xi = x[s] != 0.0 ? (double)((x[s] > 0.0) - (x[s] < 0.0)) : (double)((pg[s] < 0.0) - (pg[s] > 0.0));
                    if (t != 0.0 && (double)((t > 0.0) - (t < 0.0)) != xi) t = 0.0;
                }
                xn[s] = live[s] ? t : 0.0;
            }
            const bool bad = smooth(xn, Fn, vn);
            n_ev++;
            double dec = 0.0;
#pragma unroll
            for (int s = 0; s < PPL; ++s) dec = __builtin_fma(pg[s], xn[s] - x[s], dec);
            dec = bfly_sum(dec);
            if (!bad && Fn <= F + 1e-4 * dec) { ok = true; break; }
            alpha = alpha * 0.5;
        }
        if (!ok) {
            if (hcount > 0) { hcount = 0; continue; }       // once more from this point with an empty memory
            status = TSF_ST_MAP_LS;
            break;
        }
        // ---- curvature pair of the smooth part
        double sd[PPL], yd[PPL];
#pragma unroll
        for (int s = 0; s < PPL; ++s) { sd[s] = xn[s] - x[s]; yd[s] = vn[s] - v[s]; }
        if (map_dot<PPL>(sd, yd) > 0.0) {
#pragma unroll
            for (int s = 0; s < PPL; ++s) { Hs[(hhead * PPL + s) * W + lane] = sd[s]; Hy[(hhead * PPL + s) * W + lane] = yd[s]; }
            TSF_WAVE_SYNC();
            hhead = (hhead + 1) % MAP_M;
            if (hcount < MAP_M) hcount++;
        }
#pragma unroll
        for (int s = 0; s < PPL; ++s) { x[s] = xn[s]; v[s] = vn[s]; }
        F = Fn;
    }
    store_theta<PPL>(a, sv, n, x, a.theta);
    if (lane == 0) {
        a.fval[n] = F;
        a.status[n] = status;
        a.n_iter[n] = a.n_iter[n] + n_it;
        a.n_eval[n] = a.n_eval[n] + n_ev;
    }
}

}  // namespace tsf
