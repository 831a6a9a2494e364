// tsf_tile_kernels.h -- residual-form fit for ALIGNED panels with the design tiles shared
// through LDS (every model: logistic / linear growth, additive / multiplicative / mixed columns).
//
// fit_kernel (one 64-thread workgroup per series) streams the 172 KB step-major design matrix
// from L2 at EVERY evaluation of EVERY series: on cfg2 that is 12 TB/s of L2 reads and the
// bound of that kernel.  On an aligned panel the matrix is the same for all series, so here a
// persistent workgroup of NW waves (one series per wave, pulled from an atomic queue) walks the
// NT steps of an evaluation TOGETHER: step q of Xw/tw/cw is brought into LDS once per
// workgroup (double buffered, every thread carries its share of step q-1 while the waves
// compute step q, one workgroup barrier per step) and read from LDS by all NW waves.
// The waves therefore evaluate in rounds: in each round every wave that has a point to
// evaluate evaluates it, the others only take part in the tile traffic; between rounds each
// wave runs its own L-BFGS / line-search state machine.  Arithmetic is eval_fg's, untouched:
// results are bit-identical to fit_kernel and to oracle/prophet_canon.c (cn_eval, cn_lbfgs).
#pragma once
#include "tsf_fit_kernels.h"

namespace tsf {

template <int KP, int PPL>
__device__ __forceinline__ void make_view_t(const FitArgs &a, int64_t n, SeriesView &sv)
{
    const GridTab &gt = a.gtab[0];
    sv.T = gt.info.T; sv.NT = gt.info.NT; sv.S = gt.info.S;
    sv.P = 3 + sv.S + a.sp->K;
    int cnt = sv.T - lane_id() * sv.NT;
    cnt = cnt < 0 ? 0 : (cnt > sv.NT ? sv.NT : cnt);
    sv.cnt = cnt;
    sv.tw = a.tw; sv.cw = a.cw; sv.Xw = a.Xw;
    sv.yw = a.yw + (size_t)n * a.NTmax * W;
    sv.Lj = gt.Lj;
    sv.t_change = gt.info.t_change;
    sv.cap = a.stab[n].cap;
    sv.tau = a.sp->tau;
    sv.n_eval = 0;
}

template <int KP, int PPL, int NW>
struct TileLayout {
    static constexpr size_t xbytes = sizeof(double) * 2 * KP * W;
    static constexpr size_t tbytes = sizeof(double) * 2 * W;
    static constexpr size_t cbytes = sizeof(uint16_t) * 2 * W;
    static constexpr size_t ctl = 64;                              // n_active
    static constexpr size_t waves = xbytes + tbytes + cbytes + ctl;
    static constexpr size_t total = waves + sizeof(WaveLds<KP, PPL>) * NW;
};

template <int KP, int GROWTH, int MODE, int PPL, int NW>
__global__ __launch_bounds__(NW * 64) void fit_tile_kernel(FitArgs a, int *counter, long long *dbg)
{
#ifdef TSF_TILE_TIMING
    long long tq[4] = {0, 0, 0, 0}, tq0 = __builtin_readcyclecounter();
#define TT_LAP(k) do { const long long t_ = __builtin_readcyclecounter(); tq[k] += t_ - tq0; tq0 = t_; } while (0)
#else
#define TT_LAP(k) do { } while (0)
#endif
    unsigned char *smem = tsf_dyn_lds;
    using L = TileLayout<KP, PPL, NW>;
    const int lane = lane_id(), wid = (int)threadIdx.x >> 6;
    const DevSpec *sp = a.sp;
    TileCtx tc;
    tc.xoff = 0; tc.toff = (unsigned)L::xbytes; tc.coff = (unsigned)(L::xbytes + L::tbytes);
    tc.Xg = a.Xw; tc.tg = a.tw; tc.cg = a.cw;
    volatile int *n_active = reinterpret_cast<volatile int *>(smem + L::xbytes + L::tbytes + L::cbytes);
    WaveLds<KP, PPL> &lds = *reinterpret_cast<WaveLds<KP, PPL> *>(smem + L::waves + sizeof(WaveLds<KP, PPL>) * wid);
    for (int i = lane; i < TSF_MAX_P + W; i += W) lds.th[i] = 0.0;
    if (threadIdx.x == 0) *n_active = NW;
    __syncthreads();

    const int H = a.opt.history > MAXH ? MAXH : a.opt.history;
    const double eps = 2.220446049250313e-16;
    const double c1 = 1e-4, c2 = 0.9, minAlpha = 1e-12, min_range = 1e-16;
    const int maxLSIts = 20, maxLSRestarts = 10;
    const int eval_limit = 64 * a.opt.max_iter + 1024;

    // ---- per-wave state (one series at a time) ----
    SeriesView sv;
    make_view_t<KP, PPL>(a, 0, sv);
    int64_t n = 0;
    double xk[PPL], gk[PPL], pk[PPL], xk1[PPL], gk1[PPL], pk1[PPL];
#pragma unroll
    for (int s = 0; s < PPL; ++s) { xk[s] = gk[s] = pk[s] = xk1[s] = gk1[s] = pk1[s] = 0.0; }
    double fk = 0.0, fk1 = 0.0, alpha = 0.0, gammak = 1.0;
    int itNum = 0, ret = 0, resetB = 0, hist_len = 0, hist_head = 0;
    double dfp = 0, c1dfp = 0, c2dfp = 0, alpha0 = 0, prevF = 0, prevDFp = 0;
    double alo = 0, aloF = 0, aloDFp = 0, ahi = 0, ahiF = 0, ahiDFp = 0;
    int nits = 0, lsRestarts = 0, zoom = 0, zit = 0;
    enum { ST_FETCH = 0, ST_INIT, ST_START_ITER, ST_START_LS, ST_LS_PRE, ST_LS_EVAL, ST_POST, ST_STORE };
    int stage = ST_FETCH;
    bool idle = false;

    for (;;) {
        // ================= advance this wave's state machine to its next evaluation =========
        bool need_eval = false;
        while (!idle && !need_eval) {
            if (stage == ST_STORE) {
                store_theta<PPL>(a, sv, n, xk, a.theta);
                if (lane == 0) { a.status[n] = ret; a.n_iter[n] = itNum; a.n_eval[n] = sv.n_eval; a.fval[n] = fk; }
                stage = ST_FETCH;
            }
            if (stage == ST_FETCH) {
                int n32 = atomicAdd(counter, lane == 0 ? 1 : 0);     // branch-free, see fit_quad_kernel
                n32 = __builtin_amdgcn_readfirstlane(n32);
                n = n32;
                if (n >= a.N) {
                    idle = true;
                    if (lane == 0) atomicSub((int *)n_active, 1);
                    break;
                }
                make_view_t<KP, PPL>(a, n, sv);
                const SeriesTab st = a.stab[n];
                if (lane == 0) {
                    a.y_scale[n] = st.y_scale;
                    if (n == 0) a.grid_out[0] = a.gtab[0].info;
                }
#pragma unroll
                for (int s = 0; s < PPL; ++s) {
                    const int p = lane + s * W;
                    xk[s] = (p == 0) ? st.k0 : (p == 1 ? st.m0 : 0.0);
                    gk[s] = 0.0; pk[s] = 0.0; xk1[s] = xk[s]; gk1[s] = 0.0; pk1[s] = 0.0;
                }
                fk = 0.0; fk1 = 0.0; alpha = a.opt.init_alpha; gammak = 1.0;
                itNum = 0; ret = 0; resetB = 0; hist_len = 0; hist_head = 0;
                if (st.status0 != 0) {
                    if (st.status0 == TSF_ST_CONSTANT) {
#pragma unroll
                        for (int s = 0; s < PPL; ++s) if (lane + s * W == 2) xk[s] = -20.72326583694641;
                    }
                    ret = st.status0; fk = 0.0;
                    stage = ST_STORE;
                    continue;
                }
                stage = ST_INIT;
                need_eval = true;
                break;
            }
            if (stage == ST_POST) {
                // ---- accepted step: k is the most recent iterate ----
                double sk[PPL], yk[PPL];
#pragma unroll
                for (int s = 0; s < PPL; ++s) { sk[s] = xk[s] - xk1[s]; yk[s] = gk[s] - gk1[s]; }
                const double gradNorm = __builtin_sqrt(pdot<PPL>(gk, gk));
                const double stepNorm = __builtin_sqrt(pdot<PPL>(sk, sk));
                const double skyk = pdot<PPL>(yk, sk);
                const double ykyk = pdot<PPL>(yk, yk);
                if (resetB) {
                    const double B0fact = ykyk / skyk;
                    hist_len = 0; hist_head = 0;
#pragma unroll
                    for (int s = 0; s < PPL; ++s) pk1[s] = pk1[s] / B0fact;
                    alpha = alpha * B0fact;
                }
                gammak = skyk / ykyk;
                {
                    int slot;
                    if (hist_len < H) { slot = (hist_head + hist_len) % H; hist_len++; }
                    else { slot = hist_head; hist_head = (hist_head + 1) % H; }
                    if (lane == 0) lds.rho[slot] = 1.0 / skyk;
#pragma unroll
                    for (int s = 0; s < PPL; ++s) {
                        lds.Sb[(slot * PPL + s) * W + lane] = sk[s];
                        lds.Yb[(slot * PPL + s) * W + lane] = yk[s];
                    }
                }
                TSF_WAVE_SYNC();
#pragma unroll
                for (int s = 0; s < PPL; ++s) pk[s] = -gk[s];
                for (int h = hist_len - 1; h >= 0; --h) {
                    const int slot = (hist_head + h) % H;
                    double si[PPL], yi[PPL];
#pragma unroll
                    for (int s = 0; s < PPL; ++s) {
                        si[s] = lds.Sb[(slot * PPL + s) * W + lane];
                        yi[s] = lds.Yb[(slot * PPL + s) * W + lane];
                    }
                    const double aa = lds.rho[slot] * pdot<PPL>(si, pk);
#pragma unroll
                    for (int s = 0; s < PPL; ++s) pk[s] = __builtin_fma(-aa, yi[s], pk[s]);
                    if (lane == 0) lds.alphas[h] = aa;
                }
                TSF_WAVE_SYNC();
#pragma unroll
                for (int s = 0; s < PPL; ++s) pk[s] = pk[s] * gammak;
                for (int h = 0; h < hist_len; ++h) {
                    const int slot = (hist_head + h) % H;
                    double si[PPL], yi[PPL];
#pragma unroll
                    for (int s = 0; s < PPL; ++s) {
                        si[s] = lds.Sb[(slot * PPL + s) * W + lane];
                        yi[s] = lds.Yb[(slot * PPL + s) * W + lane];
                    }
                    const double bb = lds.rho[slot] * pdot<PPL>(yi, pk);
                    const double cc = lds.alphas[h] - bb;
#pragma unroll
                    for (int s = 0; s < PPL; ++s) pk[s] = __builtin_fma(cc, si[s], pk[s]);
                }
                TSF_WAVE_SYNC();
                const double dF = __builtin_fabs(fk1 - fk);
                const double fmaxv = __builtin_fmax(__builtin_fabs(fk1),
                                                    __builtin_fmax(__builtin_fabs(fk), 1.0));
                if (dF < a.opt.tol_obj) ret = TSF_ST_ABSF;
                else if (dF < a.opt.tol_rel_obj_eps * fmaxv) ret = TSF_ST_RELF;
                else if (gradNorm < a.opt.tol_grad) ret = TSF_ST_ABSGRAD;
                else if (-pdot<PPL>(gk, pk) / __builtin_fmax(__builtin_fabs(fk), 1.0) < a.opt.tol_rel_grad_eps) ret = TSF_ST_RELGRAD;
                else if (stepNorm < a.opt.tol_param) ret = TSF_ST_ABSX;
                else if (itNum >= a.opt.max_iter) ret = TSF_ST_MAXIT;
                else ret = 0;
                if (ret != 0) { stage = ST_STORE; continue; }
                stage = ST_START_ITER;
            }
            if (stage == ST_START_ITER) {
                itNum++;
                resetB = (itNum == 1) ? 1 : 0;
                stage = ST_START_LS;
            }
            if (stage == ST_START_LS) {
                if (resetB) {
#pragma unroll
                    for (int s = 0; s < PPL; ++s) pk[s] = -gk[s];
                }
                if (itNum > 1 && resetB != 2) {
                    const double ci = cubic_interp6(pdot<PPL>(gk1, pk1), alpha, fk - fk1,
                                                    pdot<PPL>(gk, pk), minAlpha, 1.0);
                    alpha = __builtin_fmin(1.0, 1.01 * ci);
                } else {
                    alpha = a.opt.init_alpha;
                }
                dfp = pdot<PPL>(gk, pk);
                c1dfp = c1 * dfp; c2dfp = c2 * dfp;
                alpha0 = minAlpha; prevF = fk; prevDFp = dfp;
                nits = 0; lsRestarts = 0; zoom = 0; zit = 0;
                stage = ST_LS_PRE;
            }
            if (stage == ST_LS_PRE) {
                bool ls_fail = false;
                if (!zoom) {
                    if (nits >= maxLSIts) ls_fail = true;
                } else {
                    zit++;
                    if (__builtin_fabs(alo - ahi) < min_range) {
                        ls_fail = true;
                    } else if (zit % 5 == 0) {
                        alpha = 0.5 * (alo + ahi);
                    } else {
                        const double d1 = aloDFp + ahiDFp - 3.0 * (aloF - ahiF) / (alo - ahi);
                        double d2 = __builtin_sqrt(d1 * d1 - aloDFp * ahiDFp);
                        if (ahi < alo) d2 = -d2;
                        alpha = ahi - (ahi - alo) * (ahiDFp + d2 - d1) / (ahiDFp - aloDFp + 2.0 * d2);
                        const double lo = __builtin_fmin(alo, ahi), hi = __builtin_fmax(alo, ahi),
                                     w = __builtin_fabs(alo - ahi);
                        if (!finite_f64(alpha) || alpha < lo + 0.01 * w || alpha > hi - 0.01 * w)
                            alpha = 0.5 * (alo + ahi);
                    }
                }
                if (ls_fail) {
                    if (resetB) { ret = TSF_ST_LSFAIL; stage = ST_STORE; continue; }
                    resetB = 2;
                    stage = ST_START_LS;
                    continue;
                }
                stage = ST_LS_EVAL;
            }
            if (stage == ST_LS_EVAL) {
                if (sv.n_eval >= eval_limit) { ret = TSF_ST_EVAL_LIMIT; stage = ST_STORE; continue; }
#pragma unroll
                for (int s = 0; s < PPL; ++s) xk1[s] = __builtin_fma(alpha, pk[s], xk[s]);
                need_eval = true;
            }
        }

        // ================= one evaluation round of the whole workgroup ======================
        TT_LAP(0);
        __syncthreads();
        TT_LAP(1);
        if (*n_active == 0) break;                  // every wave has run out of series
        double f1 = 0.0;
        const bool bad = eval_fg<KP, GROWTH, MODE, PPL, NW>(sp, sv, lds, xk1, f1, gk1, tc, need_eval);
        TT_LAP(2);
        if (!need_eval) continue;

        // ================= what the evaluation means for this wave ===========================
        if (stage == ST_INIT) {
            fk = f1;
            if (bad) { ret = TSF_ST_INIT_NONFINITE; stage = ST_STORE; continue; }
#pragma unroll
            for (int s = 0; s < PPL; ++s) { gk[s] = gk1[s]; pk[s] = -gk[s]; gk1[s] = 0.0; xk1[s] = 0.0; }
            stage = ST_START_ITER;
            continue;
        }
        bool ls_fail = false;
        if (bad) {
            if (!zoom) {
                if (lsRestarts >= maxLSRestarts) ls_fail = true;
                else { alpha = 0.5 * (alpha0 + alpha); lsRestarts++; }
            } else {
                alpha = 0.5 * (alpha + __builtin_fmin(alo, ahi));
                if (__builtin_fabs(__builtin_fmin(alo, ahi) - alpha) < min_range) ls_fail = true;
            }
            if (!ls_fail) continue;                 // stage stays ST_LS_EVAL: shortened step
        }
        if (!ls_fail) {
            const double newDFp = pdot<PPL>(gk1, pk);
            bool ls_ok = false;
            if (!zoom) {
                lsRestarts = 0;
                if (f1 > fk + alpha * c1dfp || (f1 >= prevF && nits > 0)) {
                    zoom = 1; alo = alpha0; aloF = prevF; aloDFp = prevDFp;
                    ahi = alpha; ahiF = f1; ahiDFp = newDFp;
                } else if (__builtin_fabs(newDFp) <= -c2dfp) {
                    ls_ok = true;
                } else if (newDFp >= 0) {
                    zoom = 1; alo = alpha; aloF = f1; aloDFp = newDFp;
                    ahi = alpha0; ahiF = prevF; ahiDFp = prevDFp;
                } else {
                    alpha0 = alpha; prevF = f1; prevDFp = newDFp;
                    alpha *= 10.0;
                    nits++;
                }
            } else {
                if (f1 > (fk + alpha * c1dfp) || f1 >= aloF) {
                    ahi = alpha; ahiF = f1; ahiDFp = newDFp;
                } else if (__builtin_fabs(newDFp) <= -c2dfp) {
                    ls_ok = true;
                } else {
                    if (newDFp * (ahi - alo) >= 0) { ahi = alo; ahiF = aloF; ahiDFp = aloDFp; }
                    alo = alpha; aloF = f1; aloDFp = newDFp;
                }
            }
            if (!ls_ok) { stage = ST_LS_PRE; continue; }
            fk1 = f1;
            { const double tf = fk; fk = fk1; fk1 = tf; }
#pragma unroll
            for (int s = 0; s < PPL; ++s) {
                const double tx = xk[s]; xk[s] = xk1[s]; xk1[s] = tx;
                const double tg = gk[s]; gk[s] = gk1[s]; gk1[s] = tg;
                const double tp = pk[s]; pk[s] = pk1[s]; pk1[s] = tp;
            }
            stage = ST_POST;
            continue;
        }
        // line search failed at the evaluation
        if (resetB) { ret = TSF_ST_LSFAIL; stage = ST_STORE; continue; }
        resetB = 2;
        stage = ST_START_LS;
    }
#ifdef TSF_TILE_TIMING
    if (dbg && lane == 0) {
        long long *o = dbg + ((size_t)blockIdx.x * NW + wid) * 4;
        o[0] = tq[0]; o[1] = tq[1]; o[2] = tq[2]; o[3] = 0;
    }
#endif
#undef TT_LAP
}

}  // namespace tsf