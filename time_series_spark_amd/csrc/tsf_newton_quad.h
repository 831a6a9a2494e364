// tsf_newton_quad.h -- Stan's Newton optimiser (tsf_newton_kernels.h) for models that are LINEAR in
// (k, m, delta, beta): linear growth, additive columns -- the BASELINE cfg5 shape, which fbprophet
// fits with Newton because its histories have fewer than 100 rows.
//
// A Newton iteration costs 4 P + 1 + ~30 log_prob/gradient evaluations (finite-difference Hessian,
// step halving).  Here the accepted point of every iteration is evaluated in residual form, which
// also re-centres the quadratic (Gram) form of tsf_quad_kernels.h there, and the 4 P perturbed
// points and the halving trials are P x P mat-vecs against the shared Z^T Z (read from L2: it is
// the same for every series of an aligned panel).  oracle cn_newton does the same when
// cn_spec.eval_mode = 1; bit-identical.
#pragma once
#include "tsf_quad_kernels.h"
#include "tsf_newton_kernels.h"

namespace tsf {

#ifndef TSF_NEWTON_QUAD_WPS
#define TSF_NEWTON_QUAD_WPS 3        // waves per SIMD the compiler budgets registers for (KP <= 16: 162 VGPRs, no scratch)
#endif

// LDS of one Newton wave.  What must survive a whole iteration: the reference point and c of the
// quadratic form, the lane constants, the eigen-solver's scratch.  The matrix (finite-difference
// Hessian, then its eigenvectors: PM x PM doubles) and the scratch of a residual pass (tables + r
// staging) are never live at the same time -- the pass is the first thing of an iteration, the
// Hessian is built after it and dead once the step is formed -- so they SHARE one region: 14.4
// instead of 19.6 KB per wave at T = 90, 11 instead of 8 waves per CU.
template <int KP>
struct NewtonQuadLds {
    double ref[W], cvec[W];         // reference point and c = Z^T r_ref
    double lanec[3 * W];            // cn_assemble_q lane constants (lane_consts)
    QlScratch ql;
};

template <int KP>
constexpr size_t newton_quad_shared_bytes(int PM, int NTmax)
{
    const size_t pass = sizeof(QuadLds<KP, 1>) + sizeof(double) * (size_t)NTmax * W;
    const size_t mat = (size_t)PM * PM * sizeof(double);
    return ((pass > mat ? pass : mat) + 15) & ~(size_t)15;
}

template <int KP>
constexpr size_t newton_quad_lds_bytes(int PM, int NTmax)
{
    return ((sizeof(NewtonQuadLds<KP>) + 15) & ~(size_t)15) + newton_quad_shared_bytes<KP>(PM, NTmax);
}

// One series, start to finish, by one wave.  Mp: Z^T Z of the series' grid (aligned panels: the
// shared one; ragged panels: built here into Mown, this block's slot of global memory).
template <int KP, bool RAGGED>
__device__ __forceinline__ void newton_one_quad(const QuadArgs &qa, NewtonQuadLds<KP> &lds,
                                                QuadLds<KP, 1> &wl, double *rb,
                                                double *Am, double *Vm, int PM, const double *Mp,
                                                double *Mown, int64_t n)
{
    constexpr int PPL = 1;
    const FitArgs &a = qa.f;
    const int lane = lane_id();
    const DevSpec *sp = a.sp;
    SeriesView sv;
    make_view_q<KP, PPL>(a, n, sv);
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
    LaneConst<PPL> lk;
    lane_consts<PPL>(sp, sv, lds.lanec, lk);
    if (RAGGED) {
        // this series has its own grid, hence its own Z^T Z: built column by column as fit_one_quad does
#pragma unroll 1
        for (int q = 0; q < qa.P4; ++q) {
            double col[PPL] = {0.0};
            if (q != 2 && q < sv.P) gram_column<KP, PPL>(sv, wl, rb, q, col);
            Mown[(size_t)q * W + lane] = col[0];
        }
        Mp = Mown;
    }
    const int P = sv.P;
    const double epsilon = 1e-3, half_epsilon = 0.5 * epsilon;
    QT_DECL;        // -DTSF_QUAD_TIMING: 0 residual passes, 1 finite differences, 2 A + A^T, 3 eigen-solver, 4 projection + step, 5 halving evaluations, 6 the rest

    enum { S_INIT = 0, S_F0, S_HALVE };
    int stage = S_INIT, ret = TSF_ST_MAXIT, it = 0, mI = 0, d = 0, pi = 0;
    double lp = 0.0, lastlp = 0.0, f0 = 0.0, f1 = 0.0, size = 2.0, fx = 0.0, s0 = 0.0;
    double sw = 0.0, cs = 0.0;      // step . (Z^T Z step) and c . step of the running iteration (the halving trials' line)
    for (;;) {
        bool bad;
        sv.n_eval++;
        QT_LAP(6);
        if (stage == S_INIT || stage == S_F0) {
            double sse_e, ztr_e[PPL];
            wl.th[W + lane] = 0.0;      // theta of a pass is zero beyond P (the region held the matrix)
            bad = resid_eval_q<KP, PPL>(sv, wl, lk, rb, x, fx, gx, sse_e, ztr_e);
            if (!bad && stage == S_F0) {
                // the accepted point becomes the reference of the quadratic form (cn_set_ref)
                lds.ref[lane] = (lane == 2) ? 0.0 : x[0];
                lds.cvec[lane] = (lane == 2) ? 0.0 : ztr_e[0];
                s0 = sse_e;
                wave_sync();
            }
            QT_LAP(6);
        } else {
            // a halving trial x = th - size * step: on the line through the point the quadratic form was just re-centred at,
            // SSE = s0 + 2 size (c . step) + size^2 (step . Z^T Z step) -- three scalars (cn_newton, round 6), not a mat-vec
            const double q2l = (size * size) * sw;
            const double cdl = -(size * cs);
            const double ssel = __builtin_fma(-2.0, cdl, s0) + q2l;
            const double zero[PPL] = {0.0};
            bad = assemble_q<PPL>(sv, lk, x, ssel, zero, fx, gx);
            QT_LAP(5);
        }
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
            // ---- finite-difference Hessian: 4 P gradient evaluations at th + pert e_d.  The
            // quadratic form has just been re-centred at th, so D = x - ref has ONE non-zero entry
            // and cn_eval_gram's mat-vec, dot products and butterflies collapse to what is
            // written here -- bit for bit: fma(m, 0, a) = a, x + 0 = x (a -0 product becomes +0
            // in the sums, hence the "+ 0.0").  Only the gradient is formed: for finite theta a
            // non-finite log_prob value comes with a non-finite gradient entry (sigma = 0 or inf).
            {
                const double thl = th[0], refl = lds.ref[lane], cvl = lds.cvec[lane];
                const double lcl = lk.lc[lane], scl = lk.sc[lane];
                const double Td = (double)sv.T;
                // sigma terms of the centre (every point except the four that perturb log sigma)
                const double ls0 = readlane_f64(thl, 2);
                const double sigma0 = dm_exp(ls0);
                const double s2_0 = sigma0 * sigma0;
                const double inv_s2_0 = 1.0 / s2_0;
                bool fd_bad = false;
                double mnext = Mp[lane];                    // column d + 1 of Z^T Z is fetched (L2) while d is worked on
                for (d = 0; d < P; ++d) {
                    const double mcol = (d == 2) ? 0.0 : mnext;
                    if (d + 1 < P) mnext = Mp[(size_t)(d + 1) * W + lane];
                    const double cvd = readlane_f64(cvl, d);
                    double accd = 0.0;
#pragma unroll
                    for (pi = 0; pi < 4; ++pi) {
                        const double pert = (pi == 0) ? -2 * epsilon : (pi == 1 ? -1 * epsilon : (pi == 2 ? epsilon : 2 * epsilon));
                        const double coef = (pi == 0) ? 1.0 / 12.0 : (pi == 1 ? -2.0 / 3.0 : (pi == 2 ? 2.0 / 3.0 : -1.0 / 12.0));
                        const double thp = (lane == d) ? thl + pert : thl;
                        const double Dd = (d == 2) ? 0.0 : readlane_f64(thp - refl, d);
                        const double v = mcol * Dd + 0.0;
                        const double q2d = Dd * readlane_f64(v, d) + 0.0;
                        const double cd = cvd * Dd + 0.0;
                        const double sse = __builtin_fma(-2.0, cd, s0) + q2d;
                        const double ztr = cvl - v;
                        double s2 = s2_0, inv_s2 = inv_s2_0;
                        if (d == 2) {
                            const double sigma = dm_exp(ls0 + pert);
                            s2 = sigma * sigma;
                            inv_s2 = 1.0 / s2;
                        }
                        const double nis = -inv_s2;
                        const double sgn = (double)((thp > 0.0) - (thp < 0.0));
                        double gv = __builtin_fma(thp, lcl, nis * ztr) + sgn * scl;
                        if (lane == 2) gv = (Td - sse * inv_s2) + 4.0 * s2;
                        if (lane >= P) gv = 0.0;
                        fd_bad = fd_bad || !finite_f64(gv);
                        accd = __builtin_fma(half_epsilon * coef, -gv, accd);
                    }
                    if (lane < P) Am[d * PM + lane] = accd;
                }
                sv.n_eval += 4 * P;
                if (__any(fd_bad)) { ret = TSF_ST_NEWTON_FAIL; break; }
            }
            // ---- H = A + A^T (in place; lane b owns the pairs (a, b), a < b, and its diagonal)
            wave_sync();
            QT_LAP(6);
            for (int r = 0; r < P; ++r) {
                double u = 0.0, v = 0.0;
                const bool mine = lane < P && r <= lane;
                if (mine) { u = Am[r * PM + lane]; v = Am[lane * PM + r]; }
                wave_sync();
                if (mine) { const double h = u + v; Am[r * PM + lane] = h; Am[lane * PM + r] = h; }
                wave_sync();
            }
            // ---- make_negative_definite_and_solve
            QT_LAP(6);
            // (Cholesky where H is negative definite -- chol_neg_solve, round 6 --, else the eigen route)
            if (!chol_neg_solve(P, PM, Am, g[0], step[0])) {
#ifdef TSF_QUAD_TIMING      // temporary: the eigen-solver's three parts in slots 0..2 (0 Householder, 1 Q, 2 QL)
            long long qlt[3] = {0, 0, 0};
            const double lam = ql_lds(P, PM, Am, Vm, lds.ql, qlt);
            qt_acc[0] += qlt[0]; qt_acc[1] += qlt[1]; qt_acc[2] += qlt[2];
            qt_t0 = __builtin_readcyclecounter();
#else
            const double lam = ql_lds(P, PM, Am, Vm, lds.ql);
#endif
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
            {   // the line of this iteration's halving trials: w = Z^T Z step (cn_eval_gram's four fma chains over the
                // rows, log sigma's entry of the direction taken out), sw = step . w, cs = c . step
                const double Dl = (lane == 2 || lane >= P) ? 0.0 : step[0];
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                const int P4l = (P + 3) & ~3;       // (rows >= P of Z^T Z are zero and meet D = 0)
                for (int qq = 0; qq < P4l; qq += 4) {
                    a0 = __builtin_fma(Mp[(size_t)(qq + 0) * W + lane], readlane_f64(Dl, qq + 0), a0);
                    a1 = __builtin_fma(Mp[(size_t)(qq + 1) * W + lane], readlane_f64(Dl, qq + 1), a1);
                    a2 = __builtin_fma(Mp[(size_t)(qq + 2) * W + lane], readlane_f64(Dl, qq + 2), a2);
                    a3 = __builtin_fma(Mp[(size_t)(qq + 3) * W + lane], readlane_f64(Dl, qq + 3), a3);
                }
                const double wv = (a0 + a1) + (a2 + a3);
                sw = bfly_sum(Dl * wv + 0.0);
                cs = bfly_sum(lds.cvec[lane] * Dl + 0.0);
            }
            QT_LAP(4);
            x[0] = th[0];
            size = 2.0; f1 = -1e100;
            stage = S_HALVE;
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
            if (mI > 0 && __builtin_fabs(lp - lastlp) < 1e-8) { ret = TSF_ST_NEWTON_CONVERGED; break; }
            if (++mI >= a.opt.max_iter) { ret = TSF_ST_MAXIT; break; }
            stage = S_F0;
            x[0] = th[0];
        }
    }
    store_theta<PPL>(a, sv, n, th, a.theta);
    if (lane == 0) { a.status[n] = ret; a.n_iter[n] = it; a.n_eval[n] = sv.n_eval; a.fval[n] = -lp; }
    QT_LAP(6);
    QT_FLUSH();
}

// persistent one-wave workgroups pulling series from a queue (the longest series needs ~7x the
// mean number of evaluations)
template <int KP, bool RAGGED>
__global__ __launch_bounds__(64, (KP <= 16 ? TSF_NEWTON_QUAD_WPS : 1)) void newton_quad_kernel(QuadArgs qa, int PM)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const FitArgs &a = qa.f;
    NewtonQuadLds<KP> &lds = *reinterpret_cast<NewtonQuadLds<KP> *>(smem);
    unsigned char *shared = smem + ((sizeof(NewtonQuadLds<KP>) + 15) & ~(size_t)15);
    QuadLds<KP, 1> &wl = *reinterpret_cast<QuadLds<KP, 1> *>(shared);      // residual-pass scratch ...
    double *rb = reinterpret_cast<double *>(shared + sizeof(QuadLds<KP, 1>));
    double *Am = reinterpret_cast<double *>(shared);                          // ... and the matrix, same bytes
    double *Vm = Am;          // ql_lds leaves the eigenvectors where the matrix was
    const int lane = lane_id();
    double *Mown = RAGGED ? qa.Mslot + (size_t)blockIdx.x * qa.P4 * W : nullptr;
    for (;;) {
        // every lane takes part in the fetch (see fit_quad_kernel)
        int n32 = atomicAdd(qa.counter, lane == 0 ? 1 : 0);
        n32 = __builtin_amdgcn_readfirstlane(n32);
        if (n32 >= a.N) break;
        newton_one_quad<KP, RAGGED>(qa, lds, wl, rb, Am, Vm, PM, qa.Mg, Mown, n32);
    }
}

}  // namespace tsf
