// tsf_mfma_kernels.h -- residual-form fit of an ALIGNED panel on the matrix cores: 16 series per
// workgroup evaluated together, the design matrix read once per 16 series instead of once per
// series, X.beta and X^T r as v_mfma_f64_16x16x4_f64 products.
//
// This is the kernel for the reference's own model -- Prophet(growth='logistic',
// seasonality_mode='multiplicative').fit(pdf), /root/reference/src/jobs/prophet_modeler.py:65-66 --
// whose likelihood is not quadratic in the parameters, so every log_prob + gradient evaluation of
// Stan's L-BFGS walks the T x K design matrix.
//
// SAME BITS as fit_kernel / oracle cn_eval.  Measured on MI355X (tools/probes/mfma_f64_probe.hip,
// profiles/r02_mfma_f64_probe.txt): v_mfma_f64_16x16x4_f64 computes
//       D[i][j] = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C[i][j]))))
// exactly (k ascending, C first, one rounding per fma), chained MFMAs continue the chain, and the
// operand layout is a = A[lane%16][lane/16], b = B[lane/16][lane%16], c[r] = C[4r + lane/16][lane%16].
// The canonical W64 order (tsf_common.h) is made of exactly such chains:
//   * X.beta of a row: fma chain over the columns, ascending, from 0           -> K/4 chained MFMAs
//   * chunk partial of a column sum X^T r: fma chain over the rows of the chunk, last row first
//     -> rows of one chunk, in that order, as the k dimension of chained MFMAs, one accumulator
//     per chunk
//   * per-chunk trend sums rt1 = sum v t, rt2 = sum v and their values "as of" a changepoint row
//     (tp1 / tp2): the same, with A = [t, 1, t*mask_i, mask_i ...] (mask_i = rows of the chunk
//     from its end down to changepoint row i: later rows contribute fma(0, v, acc) = acc)
//   * per-chunk sum of squares: diagonal of R^T R, A = B = r
// and of butterflies over the 64 chunk partials, which are done here exactly as the one-wave kernel
// does them (stages 32 and 16 inside a wave, which owns chunks w, w+16, w+32, w+48; the balanced
// tree over the remaining 16 through LDS; sse / trend sums by the owner wave's DPP butterflies).
//
// Execution model: persistent workgroups of 8 waves and 16 series slots.  Wave w owns slots 2w and
// 2w+1: it runs those series' L-BFGS state machines (state parked in LDS between rounds, L-BFGS
// history in global memory) up to their next evaluation request.  Then all 8 waves evaluate all 16
// requested points together (wave w: the chunk classes w and w+8 modulo 16 of every series), the
// owners assemble f and the gradient, and the next round starts.  A finished slot pulls the next
// series from a global queue.  A straggling series keeps its workgroup alive alone -- and is then
// evaluated by 8 waves instead of one.
#pragma once
#include "tsf_fit_kernels.h"
#include "tsf_mfma_tabs.h"
#include <type_traits>

namespace tsf {


// ---------------------------------------------------------------------------------------
// table re-layout: one block per chunk
// ---------------------------------------------------------------------------------------
__global__ void mfma_layout_kernel(const GridTab *__restrict__ gtab, const double *__restrict__ tw,
                                   const uint16_t *__restrict__ cw, const double *__restrict__ Xw,
                                   int KP, int NG, int KF, int NCB, double *__restrict__ XF,
                                   double *__restrict__ XB, double *__restrict__ XT,
                                   double *__restrict__ tq, uint16_t *__restrict__ cq,
                                   int8_t *__restrict__ cpof, int *__restrict__ overflow)
{
    const int L = blockIdx.x;
    const GridTab &gt = gtab[0];
    const int T = gt.info.T, NT = gt.info.NT;
    int cnt = T - L * NT;
    cnt = cnt < 0 ? 0 : (cnt > NT ? NT : cnt);
    const int NTP = 16 * NG;
    __shared__ int cp_j[MT_MAXCP + 1], cp_q[MT_MAXCP + 1], ncp_sh;
    if (threadIdx.x == 0) {
        // changepoint rows of this chunk: row q is the first row at or after changepoint j when
        // cprev <= j < c (the snapshot condition of eval_fg)
        int ncp = 0;
        for (int q = 0; q < cnt; ++q) {
            const unsigned cwv = cw[q * W + L];
            const int c = (int)(cwv & 0xffu), cprev = (int)(cwv >> 8);
            for (int j = cprev; j < c; ++j) {
                if (ncp < MT_MAXCP) { cp_j[ncp] = j; cp_q[ncp] = q; }
                ++ncp;
            }
        }
        if (ncp > MT_MAXCP) { atomicExch(overflow, 1); ncp = MT_MAXCP; }
        ncp_sh = ncp;
        for (int i = 0; i < 8; ++i) cpof[L * 8 + i] = (int8_t)(i < ncp ? cp_j[i] : -1);
    }
    __syncthreads();
    const int ncp = ncp_sh;
    for (int e = threadIdx.x; e < NG * KF * W; e += blockDim.x) {
        const int l = e % W, kk = (e / W) % KF, g = e / (W * KF);
        const int q = NTP - 1 - (16 * g + (l & 15)), col = 4 * kk + (l >> 4);
        XF[((size_t)(L * NG + g) * KF + kk) * W + l] = (q < cnt && col < KP) ? Xw[((size_t)q * KP + col) * W + L] : 0.0;
    }
    for (int e = threadIdx.x; e < NG * 4 * NCB * W; e += blockDim.x) {
        const int l = e % W, cb = (e / W) % NCB, rr = (e / (W * NCB)) % 4, g = e / (W * NCB * 4);
        const int q = NTP - 1 - (16 * g + 4 * rr + (l >> 4)), col = 16 * cb + (l & 15);
        XB[(((size_t)(L * NG + g) * 4 + rr) * NCB + cb) * W + l] = (q < cnt && col < KP) ? Xw[((size_t)q * KP + col) * W + L] : 0.0;
    }
    for (int e = threadIdx.x; e < NG * 4 * W; e += blockDim.x) {
        const int l = e % W, rr = (e / W) % 4, g = e / (W * 4);
        const int q = NTP - 1 - (16 * g + 4 * rr + (l >> 4)), col = l & 15;
        double v = 0.0;
        if (q < cnt) {
            const double t = tw[q * W + L];
            if (col == 0) v = t;
            else if (col == 1) v = 1.0;
            else {
                const int pi = (col - 2) >> 1;
                if (pi < ncp && q >= cp_q[pi]) v = (col & 1) ? 1.0 : t;
            }
        }
        XT[((size_t)(L * NG + g) * 4 + rr) * W + l] = v;
    }
    for (int e = threadIdx.x; e < NG * 16; e += blockDim.x) {
        const int rr = e & 3, k = (e >> 2) & 3, g = e >> 4;
        const int q = NTP - 1 - (16 * g + 4 * rr + k);
        const size_t o = (size_t)(L * NG + g) * 16 + k * 4 + rr;
        tq[o] = (q < cnt) ? tw[q * W + L] : 0.0;
        cq[o] = (q < cnt) ? (uint16_t)(cw[q * W + L] & 0xffu) : (uint16_t)0xFFFF;
    }
}

// scaled y of every series in the slot order of the tables above: one block per series
__global__ void mfma_y_kernel(const GridTab *__restrict__ gtab, const double *__restrict__ yw, int NTmax,
                              int NG, double *__restrict__ yq)
{
    const int64_t n = blockIdx.x;
    const int T = gtab[0].info.T, NT = gtab[0].info.NT, NTP = 16 * NG;
    const double *ys = yw + (size_t)n * NTmax * W;
    double *out = yq + (size_t)n * W * NG * 16;
    for (int e = threadIdx.x; e < W * NG * 16; e += blockDim.x) {
        const int rr = e & 3, k = (e >> 2) & 3, g = (e >> 4) % NG, L = e / (16 * NG);
        const int q = NTP - 1 - (16 * g + 4 * rr + k);
        int cnt = T - L * NT;
        cnt = cnt < 0 ? 0 : (cnt > NT ? NT : cnt);
        out[e] = (q < cnt) ? ys[q * W + L] : 0.0;
    }
}

// ---------------------------------------------------------------------------------------
// LDS carve-up
// ---------------------------------------------------------------------------------------
template <int KP>
struct MtSlot {                 // evaluation tables and time-axis sums of one series slot
    double ks[MT_SP + 1], mc[MT_SP + 1];
    double tp1[MT_SP], tp2[MT_SP];
    double tot1[W + 1], tot2[W + 1];    // per-chunk trend sums (evaluation), then their suffix sums (owner)
    double sse[W];
    double accR[KP];
    double pad_[(((2 * (MT_SP + 1) + 2 * MT_SP + 2 * (W + 1) + W + KP) & 1) == 0) ? 1 : 2];   // odd stride: 16 slots, 16 banks
};

struct MtScratch {              // logistic reverse sweep (eval_tail); aliased over the staging area
    double d1[MT_SP + 1], d2[MT_SP + 1], rb[MT_SP + 1], ab[MT_SP + 1];
};

struct MtVars {                 // scalar L-BFGS / line-search state of one slot (fit_kernel's locals)
    double fk, fk1, alpha, gammak, dfp, c1dfp, c2dfp, alpha0, prevF, prevDFp;
    double alo, aloF, aloDFp, ahi, ahiF, ahiDFp, gp, sigma, inv_s2, cap;
    int n, itNum, ret, resetB, hist_len, hist_head, nits, lsRestarts, zoom, zit, stage, gp_valid,
        pk1_scaled, n_eval, waiting, active;
};

struct MtState {
    double xk[W], gk[W], pk[W], pk1[W];
    double rho[MAXH], alphas[MAXH];
    MtVars v;
};

template <int KP>
struct MtLayout {
    static constexpr size_t slots = 0;
    static constexpr size_t state = slots + sizeof(MtSlot<KP>) * MT_NS;
    static constexpr size_t bm = state + sizeof(MtState) * MT_NS;
    static constexpr size_t gm = bm + sizeof(double) * MT_NS * MT_BSTR;
    static constexpr size_t stage = gm + sizeof(double) * MT_NS * MT_BSTR;
    static constexpr size_t stage_bytes = sizeof(double) * MT_LEAVES * KP * MT_NS;
    static constexpr size_t misc = stage + stage_bytes;           // owner-phase slot counters
    static constexpr size_t total = misc + 64;
    static_assert(sizeof(MtScratch) * MT_NS <= stage_bytes, "scratch must fit the staging area");
};

enum { MS_FETCH = 0, MS_INIT, MS_START_ITER, MS_START_LS, MS_LS_PRE, MS_LS_EVAL };

// ---------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------
#ifdef TSF_MFMA_TIMING      // dev only: cycles per phase (s_memtime), summed per wave into mt.dbg[block][wave][4]
#define MT_LAP(k) do { const long long t_ = __builtin_readcyclecounter(); mt_acc[k] += t_ - mt_t0; mt_t0 = t_; } while (0)
#else
#define MT_LAP(k) do { } while (0)
#endif

template <int KP, int GROWTH, int MODE>
__global__ __launch_bounds__(MT_NW * W) void fit_mfma_kernel(FitArgs a, MfmaTabs mt)
{
#ifdef TSF_MFMA_TIMING
    long long mt_acc[6] = {0, 0, 0, 0, 0, 0}, mt_t0 = __builtin_readcyclecounter();
#endif
    constexpr int PPL = 1;
    constexpr int KF = KP / 4, NCB = (KP + 15) / 16;
    static_assert(KP % 4 == 0 && KP <= 32, "one parameter per lane, at most two column blocks");
    extern __shared__ __align__(16) unsigned char smem[];
    if ((*mt.overflow != 0) != (mt.run_if_overflow != 0)) return;
    MtSlot<KP> *slots = reinterpret_cast<MtSlot<KP> *>(smem + MtLayout<KP>::slots);
    MtState *states = reinterpret_cast<MtState *>(smem + MtLayout<KP>::state);
    double *Bm = reinterpret_cast<double *>(smem + MtLayout<KP>::bm);
    double *Gm = reinterpret_cast<double *>(smem + MtLayout<KP>::gm);
    double *stg = reinterpret_cast<double *>(smem + MtLayout<KP>::stage);
    const int lane = lane_id(), wid = (int)threadIdx.x >> 6;
    const DevSpec *sp = a.sp;
    const GridTab &gt = a.gtab[0];
    // the grid is shared by every series of the panel
    SeriesView sv;
    sv.T = gt.info.T; sv.NT = gt.info.NT; sv.S = gt.S_fit; sv.S_out = gt.info.S;
    sv.P = 3 + sv.S + sp->K;
    sv.cnt = 0; sv.tw = nullptr; sv.yw = nullptr; sv.Xw = nullptr; sv.uw = nullptr; sv.Xu = nullptr; sv.cw = nullptr;
    sv.Lj = gt.Lj; sv.t_change = gt.info.t_change; sv.cap = 0.0; sv.tau = sp->tau; sv.n_eval = 0;
    set_lane_tables<1>(sp, sv);
    const int S = sv.S, NG = mt.NG;

    // slot start: nothing fetched yet
#pragma unroll 1
    for (int so = 0; so < MT_SPW; ++so) {
        const int sid = wid * MT_SPW + so;
        for (int i = lane; i < MT_BSTR; i += W) { Bm[sid * MT_BSTR + i] = 0.0; Gm[sid * MT_BSTR + i] = 0.0; }
        if (lane == 0) {
            MtVars z;
            z.fk = z.fk1 = z.alpha = z.gammak = z.dfp = z.c1dfp = z.c2dfp = z.alpha0 = z.prevF = z.prevDFp = 0.0;
            z.alo = z.aloF = z.aloDFp = z.ahi = z.ahiF = z.ahiDFp = z.gp = z.sigma = z.inv_s2 = z.cap = 0.0;
            z.n = 0; z.itNum = 0; z.ret = 0; z.resetB = 0; z.hist_len = 0; z.hist_head = 0; z.nits = 0;
            z.lsRestarts = 0; z.zoom = 0; z.zit = 0; z.stage = MS_FETCH; z.gp_valid = 0; z.pk1_scaled = 0;
            z.n_eval = 0; z.waiting = 0; z.active = 1;
            states[sid].v = z;
        }
    }
    wave_sync();

    const int H = a.opt.history > MAXH ? MAXH : a.opt.history;
    const double c1 = 1e-4, c2 = 0.9, minAlpha = 1e-12, min_range = 1e-16;
    const int maxLSIts = 20, maxLSRestarts = 10;

    for (;;) {
        // =====================================================================================
        // owner phase: this wave advances the series of its slots to their next evaluation request
        // =====================================================================================
#pragma unroll 1
        for (int so = 0; so < MT_SPW; ++so) {
        const int sid = wid * MT_SPW + so;
        MtSlot<KP> &slot = slots[sid];
        MtState &st = states[sid];
        MtScratch &scr = *reinterpret_cast<MtScratch *>(smem + MtLayout<KP>::stage + sizeof(MtScratch) * sid);
        double *hS = mt.hist + ((size_t)blockIdx.x * MT_NS + sid) * (2 * MAXH * W);
        double *hY = hS + MAXH * W;
        MtVars v = st.v;
        if (v.active) {
            double xk = st.xk[lane], gk = st.gk[lane], pk = st.pk[lane], pk1 = st.pk1[lane];
            double xk1 = Bm[sid * MT_BSTR + lane], gk1 = 0.0, f1 = 0.0;
            bool bad = false, resume = v.waiting != 0;
            if (resume) {
                // ---- f and the gradient of the point evaluated in the last round
                const double ssel = slot.sse[lane], r1l = slot.tot1[lane], r2l = slot.tot2[lane];
                const double sse_t = bfly_sum(ssel);
                const double s1 = suffix_scan(r1l), s2v = suffix_scan(r2l);
                wave_sync();
                slot.tot1[lane] = s1; slot.tot2[lane] = s2v;
                if (lane == 0) { slot.tot1[W] = 0.0; slot.tot2[W] = 0.0; }
                wave_sync();
                double th[1] = {xk1}, gg[1];
                sv.cap = v.cap;
                bad = eval_tail<GROWTH, PPL>(sp, sv, slot, scr, th, v.sigma, v.inv_s2, sse_t, f1, gg);
                gk1 = gg[0];
            }
            bool request = false;
            for (;;) {
                if (!resume && v.stage == MS_FETCH) {
                    // Every lane takes part in the queue fetch (lane 0 adds 1, the others 0): see the
                    // compiler note at the fetch of fit_quad_kernel
                    int n32 = atomicAdd(mt.counter, lane == 0 ? 1 : 0);
                    n32 = __builtin_amdgcn_readfirstlane(n32);
                    if (n32 >= a.N) { v.active = 0; break; }
                    v.n = n32;
                    const SeriesTab stb = a.stab[n32];
                    if (lane == 0) {
                        a.y_scale[n32] = stb.y_scale;
                        if (n32 == 0) a.grid_out[0] = gt.info;
                    }
                    xk = (lane == 0) ? stb.k0 : (lane == 1 ? stb.m0 : 0.0);
                    gk = 0.0; pk = 0.0; xk1 = xk; gk1 = 0.0; pk1 = 0.0;
                    if (stb.status0 != 0) {
                        if (stb.status0 == TSF_ST_CONSTANT && lane == 2) xk = -20.72326583694641;
                        double xo[1] = {xk};
                        store_theta<PPL>(a, sv, n32, xo, a.theta);
                        if (lane == 0) { a.status[n32] = stb.status0; a.n_iter[n32] = 0; a.n_eval[n32] = 0; a.fval[n32] = 0.0; }
                        continue;
                    }
                    v.cap = stb.cap;
                    v.fk = 0.0; v.fk1 = 0.0; v.alpha = a.opt.init_alpha; v.gammak = 1.0;
                    v.itNum = 0; v.ret = 0; v.resetB = 0; v.hist_len = 0; v.hist_head = 0;
                    v.dfp = v.c1dfp = v.c2dfp = v.alpha0 = v.prevF = v.prevDFp = 0.0;
                    v.alo = v.aloF = v.aloDFp = v.ahi = v.ahiF = v.ahiDFp = 0.0;
                    v.nits = 0; v.lsRestarts = 0; v.zoom = 0; v.zit = 0;
                    v.gp = 0.0; v.gp_valid = 0; v.pk1_scaled = 0; v.n_eval = 0;
                    v.stage = MS_INIT;
                    request = true;
                    break;
                }
                bool ls_fail = false;
                if (!resume) {
                    if (v.stage == MS_START_ITER) {
                        v.itNum++;
                        v.resetB = (v.itNum == 1) ? 1 : 0;
                        v.stage = MS_START_LS;
                    }
                    if (v.stage == MS_START_LS) {
                        if (v.resetB) { pk = -gk; v.gp_valid = 0; }
                        if (!v.gp_valid) { double x_[1] = {gk}, y_[1] = {pk}; v.gp = pdot<PPL>(x_, y_); }
                        v.gp_valid = 0;
                        if (v.itNum > 1 && v.resetB != 2) {
                            double x_[1] = {gk1}, y_[1] = {pk1};
                            const double gp1 = v.pk1_scaled ? pdot<PPL>(x_, y_) : v.dfp;
                            const double ci = cubic_interp6(gp1, v.alpha, v.fk - v.fk1, v.gp, minAlpha, 1.0);
                            v.alpha = __builtin_fmin(1.0, 1.01 * ci);
                        } else {
                            v.alpha = a.opt.init_alpha;
                        }
                        v.dfp = v.gp;
                        v.c1dfp = c1 * v.dfp; v.c2dfp = c2 * v.dfp;
                        v.alpha0 = minAlpha; v.prevF = v.fk; v.prevDFp = v.dfp;
                        v.nits = 0; v.lsRestarts = 0; v.zoom = 0; v.zit = 0;
                        v.stage = MS_LS_PRE;
                    }
                    if (v.stage == MS_LS_PRE) {
                        if (!v.zoom) {
                            if (v.nits >= maxLSIts) ls_fail = true;
                        } else {
                            v.zit++;
                            if (__builtin_fabs(v.alo - v.ahi) < min_range) {
                                ls_fail = true;
                            } else if (v.zit % 5 == 0) {
                                v.alpha = 0.5 * (v.alo + v.ahi);
                            } else {
                                const double d1 = v.aloDFp + v.ahiDFp - 3.0 * (v.aloF - v.ahiF) / (v.alo - v.ahi);
                                double d2 = __builtin_sqrt(d1 * d1 - v.aloDFp * v.ahiDFp);
                                if (v.ahi < v.alo) d2 = -d2;
                                v.alpha = v.ahi - (v.ahi - v.alo) * (v.ahiDFp + d2 - d1) / (v.ahiDFp - v.aloDFp + 2.0 * d2);
                                const double lo = __builtin_fmin(v.alo, v.ahi), hi = __builtin_fmax(v.alo, v.ahi),
                                             wd = __builtin_fabs(v.alo - v.ahi);
                                if (!finite_f64(v.alpha) || v.alpha < lo + 0.01 * wd || v.alpha > hi - 0.01 * wd)
                                    v.alpha = 0.5 * (v.alo + v.ahi);
                            }
                        }
                        if (!ls_fail) v.stage = MS_LS_EVAL;
                    }
                    if (!ls_fail) {
                        // guard against a line search that never settles (oracle cn_lbfgs eval_limit)
                        if (v.n_eval >= 64 * a.opt.max_iter + 1024) {
                            v.ret = TSF_ST_EVAL_LIMIT;
                        } else {
                            xk1 = __builtin_fma(v.alpha, pk, xk);
                            request = true;
                            break;
                        }
                    }
                } else {
                    resume = false;
                    bool finished = false;
                    if (v.stage == MS_INIT) {
                        v.fk = f1;
                        if (bad) { v.ret = TSF_ST_INIT_NONFINITE; finished = true; }
                        else {
                            gk = gk1; pk = -gk; gk1 = 0.0; xk1 = 0.0;
                            v.stage = MS_START_ITER;
                            continue;
                        }
                    }
                    if (!finished && bad) {
                        if (!v.zoom) {
                            if (v.lsRestarts >= maxLSRestarts) ls_fail = true;
                            else { v.alpha = 0.5 * (v.alpha0 + v.alpha); v.lsRestarts++; }
                        } else {
                            v.alpha = 0.5 * (v.alpha + __builtin_fmin(v.alo, v.ahi));
                            if (__builtin_fabs(__builtin_fmin(v.alo, v.ahi) - v.alpha) < min_range) ls_fail = true;
                        }
                        if (!ls_fail) continue;            // re-evaluate at the shortened step
                    }
                    if (!finished && !ls_fail) {
                        double x_[1] = {gk1}, y_[1] = {pk};
                        const double newDFp = pdot<PPL>(x_, y_);
                        bool ls_ok = false;
                        if (!v.zoom) {
                            v.lsRestarts = 0;
                            if (f1 > v.fk + v.alpha * v.c1dfp || (f1 >= v.prevF && v.nits > 0)) {
                                v.zoom = 1; v.alo = v.alpha0; v.aloF = v.prevF; v.aloDFp = v.prevDFp;
                                v.ahi = v.alpha; v.ahiF = f1; v.ahiDFp = newDFp;
                            } else if (__builtin_fabs(newDFp) <= -v.c2dfp) {
                                ls_ok = true;
                            } else if (newDFp >= 0) {
                                v.zoom = 1; v.alo = v.alpha; v.aloF = f1; v.aloDFp = newDFp;
                                v.ahi = v.alpha0; v.ahiF = v.prevF; v.ahiDFp = v.prevDFp;
                            } else {
                                v.alpha0 = v.alpha; v.prevF = f1; v.prevDFp = newDFp;
                                v.alpha *= 10.0;
                                v.nits++;
                            }
                        } else {
                            if (f1 > (v.fk + v.alpha * v.c1dfp) || f1 >= v.aloF) {
                                v.ahi = v.alpha; v.ahiF = f1; v.ahiDFp = newDFp;
                            } else if (__builtin_fabs(newDFp) <= -v.c2dfp) {
                                ls_ok = true;
                            } else {
                                if (newDFp * (v.ahi - v.alo) >= 0) { v.ahi = v.alo; v.ahiF = v.aloF; v.ahiDFp = v.aloDFp; }
                                v.alo = v.alpha; v.aloF = f1; v.aloDFp = newDFp;
                            }
                        }
                        if (!ls_ok) { v.stage = MS_LS_PRE; continue; }
                        v.fk1 = f1;
                        // ---- accepted step: k becomes the most recent iterate ----
                        { const double tf = v.fk; v.fk = v.fk1; v.fk1 = tf; }
                        { const double tx = xk; xk = xk1; xk1 = tx; }
                        { const double tg = gk; gk = gk1; gk1 = tg; }
                        { const double tp = pk; pk = pk1; pk1 = tp; }
                        double sk[1] = {xk - xk1}, yk[1] = {gk - gk1}, gkv[1] = {gk};
                        // g.g, s.s, y.s, y.y: one four-fold butterfly (lanes 0, 2, 1, 3), square roots and
                        // quotients one lane each (see fit_one_quad)
                        const double dots = bfly_sum4_lanes(pdot_part<PPL>(gkv, gkv), pdot_part<PPL>(sk, sk),
                                                            pdot_part<PPL>(yk, sk), pdot_part<PPL>(yk, yk));
                        const double nrm = __builtin_sqrt(dots);
                        const double gradNorm = readlane_f64(nrm, 0), stepNorm = readlane_f64(nrm, 2);
                        double qnum = dpp_mov<0x07>(dots);          // quad_perm [3,1,0,0]: y.y, y.s, -, -
                        if ((lane & 3) >= 2) qnum = 1.0;
                        const double qden = dpp_mov<0x5D>(dots);    // quad_perm [1,3,1,1]: y.s, y.y, y.s, y.s
                        const double qv = qnum / qden;
                        if (v.resetB) {
                            const double B0fact = readlane_f64(qv, 0);
                            v.hist_len = 0; v.hist_head = 0;
                            pk1 = pk1 / B0fact;
                            v.alpha = v.alpha * B0fact;
                            v.pk1_scaled = 1;
                        } else {
                            v.pk1_scaled = 0;
                        }
                        v.gammak = readlane_f64(qv, 1);
                        const double rho_new = readlane_f64(qv, 2);
                        {
                            int hslot;
                            if (v.hist_len < H) { hslot = (v.hist_head + v.hist_len) % H; v.hist_len++; }
                            else { hslot = v.hist_head; v.hist_head = (v.hist_head + 1) % H; }
                            if (lane == 0) st.rho[hslot] = rho_new;
                            hS[hslot * W + lane] = sk[0];
                            hY[hslot * W + lane] = yk[0];
                        }
                        wave_sync();
                        // history vectors: each lane reads back only what it wrote itself; all loads of
                        // the recursion are issued together
                        double si[MAXH], yi[MAXH];
#pragma unroll
                        for (int h = 0; h < MAXH; ++h) {
                            si[h] = 0.0; yi[h] = 0.0;
                            if (h < v.hist_len) {
                                const int hs = (v.hist_head + h) % H;
                                si[h] = hS[hs * W + lane]; yi[h] = hY[hs * W + lane];
                            }
                        }
                        pk = -gk;
#pragma unroll
                        for (int h = MAXH - 1; h >= 0; --h) {
                            if (h < v.hist_len) {
                                const int hs = (v.hist_head + h) % H;
                                double x2[1] = {si[h]}, y2[1] = {pk};
                                const double aa = lane63(st.rho[hs] * pdot_l63<PPL>(x2, y2));
                                pk = __builtin_fma(-aa, yi[h], pk);
                                if (lane == 0) st.alphas[h] = aa;
                            }
                        }
                        wave_sync();
                        pk = pk * v.gammak;
#pragma unroll
                        for (int h = 0; h < MAXH; ++h) {
                            if (h < v.hist_len) {
                                const int hs = (v.hist_head + h) % H;
                                double x2[1] = {yi[h]}, y2[1] = {pk};
                                const double cc = lane63(st.alphas[h] - st.rho[hs] * pdot_l63<PPL>(x2, y2));
                                pk = __builtin_fma(cc, si[h], pk);
                            }
                        }
                        wave_sync();
                        const double dF = __builtin_fabs(v.fk1 - v.fk);
                        const double fmaxv = __builtin_fmax(__builtin_fabs(v.fk1),
                                                            __builtin_fmax(__builtin_fabs(v.fk), 1.0));
                        { double x2[1] = {gk}, y2[1] = {pk}; v.gp = pdot<PPL>(x2, y2); }
                        v.gp_valid = 1;
                        if (dF < a.opt.tol_obj) v.ret = TSF_ST_ABSF;
                        else if (dF < a.opt.tol_rel_obj_eps * fmaxv) v.ret = TSF_ST_RELF;
                        else if (gradNorm < a.opt.tol_grad) v.ret = TSF_ST_ABSGRAD;
                        else if (-v.gp / __builtin_fmax(__builtin_fabs(v.fk), 1.0) < a.opt.tol_rel_grad_eps) v.ret = TSF_ST_RELGRAD;
                        else if (stepNorm < a.opt.tol_param) v.ret = TSF_ST_ABSX;
                        else if (v.itNum >= a.opt.max_iter) v.ret = TSF_ST_MAXIT;
                        else v.ret = 0;
                        if (v.ret == 0) { v.stage = MS_START_ITER; continue; }
                        finished = true;
                    }
                    if (finished) {
                        double xo[1] = {xk};
                        store_theta<PPL>(a, sv, v.n, xo, a.theta);
                        if (lane == 0) { a.status[v.n] = v.ret; a.n_iter[v.n] = v.itNum; a.n_eval[v.n] = v.n_eval; a.fval[v.n] = v.fk; }
                        v.stage = MS_FETCH;
                        continue;
                    }
                }
                if (v.ret == TSF_ST_EVAL_LIMIT) {
                    double xo[1] = {xk};
                    store_theta<PPL>(a, sv, v.n, xo, a.theta);
                    if (lane == 0) { a.status[v.n] = v.ret; a.n_iter[v.n] = v.itNum; a.n_eval[v.n] = v.n_eval; a.fval[v.n] = v.fk; }
                    v.stage = MS_FETCH;
                    continue;
                }
                // line search failed
                if (v.resetB) {
                    v.ret = TSF_ST_LSFAIL;
                    double xo[1] = {xk};
                    store_theta<PPL>(a, sv, v.n, xo, a.theta);
                    if (lane == 0) { a.status[v.n] = v.ret; a.n_iter[v.n] = v.itNum; a.n_eval[v.n] = v.n_eval; a.fval[v.n] = v.fk; }
                    v.stage = MS_FETCH;
                    continue;
                }
                v.resetB = 2;
                v.stage = MS_START_LS;
            }
            v.waiting = request ? 1 : 0;
            if (request) {
                // the point to evaluate, its segment tables and sigma terms
                v.n_eval++;
                double th[1] = {xk1};
                const double ls = readlane_f64(xk1, 2);
                v.sigma = dm_exp_sel(ls);
                v.inv_s2 = 1.0 / (v.sigma * v.sigma);
                segment_tables<GROWTH, PPL>(sv, slot, th);
            }
            Bm[sid * MT_BSTR + lane] = xk1;
            Gm[sid * MT_BSTR + lane] = gk1;
            st.xk[lane] = xk; st.gk[lane] = gk; st.pk[lane] = pk; st.pk1[lane] = pk1;
            if (lane == 0) st.v = v;
        }
        }
        MT_LAP(0);
        __syncthreads();
        MT_LAP(1);
        // anything left to evaluate in this workgroup?
        int any = 0;
#pragma unroll
        for (int s2 = 0; s2 < MT_NS; ++s2) any |= states[s2].v.waiting;
        if (!any) break;
#ifdef TSF_MFMA_TIMING
        mt_acc[5]++;
#endif

        // =====================================================================================
        // few requests (the tail of a launch: one long series keeps its workgroup alive): a tile
        // evaluation would do 16 series' work for one.  Instead the 8 waves share ONE series'
        // evaluation the way eval_fg orders it -- rows by (chunk = lane, step q): the steps are
        // dealt to the waves for the forward pass (r, r g, v per row -> LDS), then the design
        // columns are dealt to the waves for the per-chunk fma chains and the butterflies over the
        // chunks, and one wave runs the sse / trend chains.  Same operations, same order, same bits.
        // =====================================================================================
        int n_act = 0;
#pragma unroll
        for (int s2 = 0; s2 < MT_NS; ++s2) n_act += states[s2].v.waiting;
        const bool solo = n_act <= MT_SOLO_MAX && (size_t)sv.NT * 3 * W * sizeof(double) <= MtLayout<KP>::stage_bytes;
        if (solo) {
#pragma unroll 1
            for (int sj = 0; sj < MT_NS; ++sj) {
                if (!states[sj].v.waiting) continue;
                MtSlot<KP> &so = slots[sj];
                const MtVars &vo = states[sj].v;
                const int NT = sv.NT;
                int cnt = sv.T - lane * NT;
                cnt = cnt < 0 ? 0 : (cnt > NT ? NT : cnt);
                double *rbR = stg, *rbU = stg + (size_t)NT * W, *rbV = stg + (size_t)2 * NT * W;   // r, r or r g, v
                const double *ywn = a.yw + (size_t)vo.n * a.NTmax * W;
                // ---- forward: wave w takes steps q = w, w + 8, ...
                {
                    double bs[KP];
#pragma unroll
                    for (int jc = 0; jc < KP; ++jc) bs[jc] = Bm[sj * MT_BSTR + 3 + S + jc];
#pragma unroll 2
                    for (int q = wid; q < NT; q += MT_NW) {
                        double r0 = 0.0, ru = 0.0, vt = 0.0;
                        if (q < cnt) {
                            const int idx = q * W + lane;
                            const int c = (int)(a.cw[idx] & 0xffu);
                            const double ti = a.tw[idx], yi = ywn[idx];
                            const double *xp = a.Xw + (size_t)q * KP * W + lane;
                            double xa = 0.0, xm = 0.0;
#pragma unroll
                            for (int jc = 0; jc < KP; ++jc) {
                                if (MODE == 0) xa = __builtin_fma(xp[jc * W], bs[jc], xa);
                                else xm = __builtin_fma(xp[jc * W], bs[jc], xm);
                            }
                            const double ksc = so.ks[c], mcc = so.mc[c];
                            double gtr, qv = 0.0;
                            if (GROWTH == 0) {
                                gtr = __builtin_fma(ksc, ti, mcc);
                            } else {
                                const double z = ksc * (ti - mcc);
                                const double e = dm_exp_sel(-z);
                                const double sg = 1.0 / (1.0 + e);
                                gtr = vo.cap * sg;
                                qv = gtr * (1.0 - sg);
                            }
                            const double opm = 1.0 + xm;
                            const double mu = __builtin_fma(gtr, opm, xa);
                            r0 = yi - mu;
                            ru = (MODE == 0) ? r0 : r0 * gtr;
                            vt = r0 * opm;
                            if (GROWTH == 1) vt = vt * qv;
                        }
                        rbR[q * W + lane] = r0; rbU[q * W + lane] = ru; rbV[q * W + lane] = vt;
                    }
                }
                __syncthreads();
                // ---- backward: wave w takes the design columns w, w + 8, w + 16, w + 24; lane = chunk
                {
                    double acc[4] = {0.0, 0.0, 0.0, 0.0};
                    // four steps' loads in flight at a time (a step-by-step loop would wait for L2 at
                    // every step); steps past NT - 1 rounded up read row 0 again with weight 0
                    const int NT4 = (NT + 3) & ~3;
#pragma unroll 1
                    for (int q0 = NT4 - 4; q0 >= 0; q0 -= 4) {
                        double xv[4][4], rv4[4];
#pragma unroll
                        for (int k = 3; k >= 0; --k) {
                            const int q = q0 + k, qc = q < NT ? q : 0;
                            rv4[k] = q < NT ? rbU[qc * W + lane] : 0.0;
                            const double *xp = a.Xw + (size_t)qc * KP * W + lane;
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int jc = wid + MT_NW * u;
                                xv[k][u] = (jc < KP) ? xp[jc * W] : 0.0;
                            }
                        }
#pragma unroll
                        for (int k = 3; k >= 0; --k)
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (q0 + k < NT) acc[u] = __builtin_fma(xv[k][u], rv4[k], acc[u]);   // rows past the end of a chunk: X = 0, r = 0
                    }
                    // butterfly 32, 16, 1, 2, 4, 8 over the chunks (column_sums, one group of 4 columns)
                    double c2[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        double x0 = acc[2 * i], x1 = acc[2 * i + 1];
                        swap32(x0, x1);
                        c2[i] = x0 + x1;
                    }
                    double x0 = c2[0], x1 = c2[1];
                    swap16(x0, x1);
                    const double dsum = row_bfly_sum(x0 + x1);
                    if ((lane & 15) == 0) {
                        const int r = lane >> 4;
                        const int sub = (r == 0) ? 0 : (r == 1 ? 2 : (r == 2 ? 1 : 3));
                        const int jc = wid + MT_NW * sub;
                        if (jc < KP) so.accR[jc] = dsum;
                    }
                }
                if (wid == MT_NW - 1) {
                    // ---- per-chunk sum of squares and trend sums with their changepoint-row values
                    double sse = 0.0, rt1 = 0.0, rt2 = 0.0;
#pragma unroll 4
                    for (int q = NT - 1; q >= 0; --q) {
                        const int idx = q * W + lane;
                        const unsigned cwv = (unsigned)a.cw[idx];       // (loads for every lane: no divergent wait)
                        const double ti = a.tw[idx];
                        if (q < cnt) {
                            const int c = (int)(cwv & 0xffu), cprev = (int)(cwv >> 8);
                            const double r0 = rbR[idx], vt = rbV[idx];
                            sse = __builtin_fma(r0, r0, sse);
                            rt1 = __builtin_fma(vt, ti, rt1);
                            rt2 = rt2 + vt;
                            for (int jj = cprev; jj < c; ++jj) { so.tp1[jj] = rt1; so.tp2[jj] = rt2; }
                        }
                    }
                    so.sse[lane] = sse; so.tot1[lane] = rt1; so.tot2[lane] = rt2;
                }
                __syncthreads();        // the row buffers are free for the next requested slot
            }
        } else {
            const int j = lane & 15, kq = lane >> 4;
            MtSlot<KP> &sj = slots[j];
            const MtVars &vj = states[j].v;
            const double capj = vj.cap;
            const double *yj = mt.yq + (size_t)vj.n * W * NG * 16;
            double bop[KF];
#pragma unroll
            for (int kk = 0; kk < KF; ++kk) bop[kk] = Bm[j * MT_BSTR + 3 + S + 4 * kk + kq];
            // The wave walks its 2 x 4 chunks x NG row groups as ONE sequence of tiles; the operands
            // of tile i+1 (design tiles, t, segment indices: L2) are requested while tile i is
            // computed, y (HBM / Infinity Cache) two tiles ahead.  The loop body is kept free of
            // branches (tile positions advance by selects, requests past the end wrap to tile 0, zero
            // padding rows are multiplied through) so that the compiler's s_waitcnt placement can
            // count the loads in flight instead of draining them at the loop head.
            const int n_tiles = (MT_LEAVES / MT_NW) * 4 * NG;
            struct TilePos { int g, ci, lf; };
            auto pos_next = [&](TilePos p) -> TilePos {
                p.g++;
                const bool cg = p.g == NG;
                p.g = cg ? 0 : p.g;
                p.ci += cg ? 1 : 0;
                const bool cc = p.ci == 4;
                p.ci = cc ? 0 : p.ci;
                p.lf += cc ? 1 : 0;
                p.lf = (p.lf == MT_LEAVES / MT_NW) ? 0 : p.lf;     // past the end: wrap (harmless re-request)
                return p;
            };
            auto pos_chunk = [&](const TilePos &p) -> int {
                return wid + MT_NW * p.lf + 16 * (((p.ci & 1) << 1) | (p.ci >> 1));     // chunk order 0, 2, 1, 3
            };
            auto pos_tile = [&](const TilePos &p) -> size_t { return (size_t)pos_chunk(p) * NG + p.g; };
            double p_xf[KF], p_xb[4 * NCB], p_xt[4];
            d4_t p_t4, p_yA, p_yB;             // y: requested two tiles ahead (even tiles yA, odd tiles yB)
            uint2 p_c4;
            unsigned long long p_cp = 0;       // cpof row of the tile's chunk
            // three operand groups, each re-requested for the NEXT tile right after this tile has
            // consumed it (the registers rotate: no second buffer)
            auto request_fwd = [&](const TilePos &p) {
                const double *xf = mt.XF + pos_tile(p) * KF * W + lane;
#pragma unroll
                for (int kk = 0; kk < KF; ++kk) p_xf[kk] = xf[kk * W];
            };
            auto request_row = [&](const TilePos &p) {
                const size_t eo = pos_tile(p) * 16 + kq * 4;
                p_t4 = *reinterpret_cast<const d4_t *>(mt.tq + eo);
                p_c4 = *reinterpret_cast<const uint2 *>(mt.cq + eo);
            };
            auto load_y = [&](const TilePos &p) -> d4_t {
                return *reinterpret_cast<const d4_t *>(yj + pos_tile(p) * 16 + kq * 4);
            };
            auto request_bwd = [&](const TilePos &p) {
                const size_t tile = pos_tile(p);
                const double *xb = mt.XB + tile * 4 * NCB * W + lane;
#pragma unroll
                for (int u = 0; u < 4 * NCB; ++u) p_xb[u] = xb[u * W];
                const double *xt = mt.XT + tile * 4 * W + lane;
#pragma unroll
                for (int u = 0; u < 4; ++u) p_xt[u] = xt[u * W];
                p_cp = *reinterpret_cast<const unsigned long long *>(mt.cpof + pos_chunk(p) * 8);
            };
            TilePos cur = {0, 0, 0};
            TilePos nxt = pos_next(cur);
            TilePos ypos = pos_next(nxt);
            request_fwd(cur); request_row(cur); request_bwd(cur);
            p_yA = load_y(cur); p_yB = load_y(nxt);
            d4_t hold0[NCB], aX[NCB], aT = d4_t{0.0, 0.0, 0.0, 0.0}, aS = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) { hold0[cb] = d4_t{0.0, 0.0, 0.0, 0.0}; aX[cb] = hold0[cb]; }
            // one tile; y_buf is consumed and then re-requested for the tile two ahead (no register is
            // copied while a load into it is in flight: a copy would have to wait for the load)
            // CI >= 0: chunks of ONE row group (NG == 1, every series up to 1024 rows), position in the
            // chunk order known at compile time: a tile is then one basic block.  CI < 0: general case.
            auto tile_body = [&](d4_t &y_buf, auto ci_tag) {
                constexpr int CI = decltype(ci_tag)::value;
                const int L = pos_chunk(cur), ci = (CI >= 0) ? CI : cur.ci, leaf = wid + MT_NW * cur.lf;
                const int g = (CI >= 0) ? 0 : cur.g;
                {
                    const bool first = g == 0;
                    const d4_t z4 = d4_t{0.0, 0.0, 0.0, 0.0};
                    aT = first ? z4 : aT; aS = first ? z4 : aS;
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) aX[cb] = first ? z4 : aX[cb];
                }
                d4_t D = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < KF; ++kk)
                    D = __builtin_amdgcn_mfma_f64_16x16x4f64(p_xf[kk], bop[kk], D, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                request_fwd(nxt);
                const d4_t t4 = p_t4, y4 = y_buf;
                const uint2 c4 = p_c4;
                double rv[4], bv[4], vv[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const unsigned cu = (rr == 0) ? (c4.x & 0xffffu) : (rr == 1 ? (c4.x >> 16) : (rr == 2 ? (c4.y & 0xffffu) : (c4.y >> 16)));
                    const bool valid = cu != 0xffffu;
                    const int c = valid ? (int)cu : 0;
                    const double ti = t4[rr], yi = y4[rr];
                    const double ksc = sj.ks[c], mcc = sj.mc[c];
                    const double xa = (MODE == 0) ? D[rr] : 0.0, xm = (MODE == 1) ? D[rr] : 0.0;
                    double gtr, qv = 0.0;
                    if (GROWTH == 0) {
                        gtr = __builtin_fma(ksc, ti, mcc);
                    } else {
                        const double z = ksc * (ti - mcc);
                        const double e = dm_exp_sel(-z);
                        const double sg = 1.0 / (1.0 + e);
                        gtr = capj * sg;
                        qv = gtr * (1.0 - sg);
                    }
                    const double opm = 1.0 + xm;
                    const double mu = __builtin_fma(gtr, opm, xa);
                    const double r0 = yi - mu;
                    const double rg = r0 * gtr;
                    double vt = r0 * opm;
                    if (GROWTH == 1) vt = vt * qv;
                    rv[rr] = valid ? r0 : 0.0;
                    bv[rr] = valid ? ((MODE == 0) ? r0 : rg) : 0.0;
                    vv[rr] = valid ? vt : 0.0;
                    // two rows at a time: four interleaved exp / division chains run out of registers
                    if (rr == 1) __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
                request_row(nxt);
                y_buf = load_y(ypos);
                // (4-row blocks that are all padding -- the first (16 NG - NT) / 4 of group 0 -- are
                // multiplied through: zero rows of X and r = 0 leave every accumulator as it is)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
                        aX[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(p_xb[rr * NCB + cb], bv[rr], aX[cb], 0, 0, 0);
                    aT = __builtin_amdgcn_mfma_f64_16x16x4f64(p_xt[rr], vv[rr], aT, 0, 0, 0);
                    aS = __builtin_amdgcn_mfma_f64_16x16x4f64(rv[rr], rv[rr], aS, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long cp_row = p_cp;
                request_bwd(nxt);
                cur = nxt; nxt = pos_next(nxt); ypos = pos_next(ypos);
                if (CI < 0 && g != NG - 1) return;
                // ---- chunk L done: its trend sums, changepoint-row partials and sum of squares
#pragma unroll
                for (int r2 = 0; r2 < 4; ++r2) {
                    const int col = 4 * r2 + kq;
                    if (col == 0) sj.tot1[L] = aT[r2];
                    else if (col == 1) sj.tot2[L] = aT[r2];
                    else {
                        const int jj = (int)(signed char)(cp_row >> (8 * ((col - 2) >> 1)));
                        if (jj >= 0) { if (col & 1) sj.tp2[jj] = aT[r2]; else sj.tp1[jj] = aT[r2]; }
                    }
                }
                if (kq == (j & 3)) {
                    const int q2 = j >> 2;
                    sj.sse[L] = (q2 == 0) ? aS[0] : (q2 == 1 ? aS[1] : (q2 == 2 ? aS[2] : aS[3]));
                }
                // ---- stages 32 and 16 of the column butterfly: (p[l] + p[l+32]) + (p[l+16] + p[l+48]).
                // The first pair sum waits in this wave's own staging rows (nobody else touches them
                // before the barrier) while the registers take the second pair.
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    if (ci == 0 || ci == 2) hold0[cb] = aX[cb];
                    else hold0[cb] = hold0[cb] + aX[cb];
                }
                if (ci == 1 || ci == 3) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                        for (int r2 = 0; r2 < 4; ++r2) {
                            const int col = 16 * cb + 4 * r2 + kq;
                            if (col < KP) {
                                double *dst = &stg[((size_t)leaf * KP + col) * MT_NS + j];
                                *dst = (ci == 1) ? hold0[cb][r2] : *dst + hold0[cb][r2];
                            }
                        }
                }
            };
#pragma unroll 1
            for (int it = 0; it < n_tiles; it += 4) {       // n_tiles = 8 NG
                tile_body(p_yA, std::integral_constant<int, -1>());
                tile_body(p_yB, std::integral_constant<int, -1>());
                tile_body(p_yA, std::integral_constant<int, -1>());
                tile_body(p_yB, std::integral_constant<int, -1>());
            }
        }
        MT_LAP(2);
        __syncthreads();
        MT_LAP(3);
        // ---- the remaining 16-leaf balanced tree (butterfly stages 1, 2, 4, 8) over the chunk classes
        if (!solo)
        for (int o = (int)threadIdx.x; o < KP * MT_NS; o += MT_NW * W) {
            const int col = o / MT_NS, jx = o % MT_NS;
            double x[MT_LEAVES];
#pragma unroll
            for (int w2 = 0; w2 < MT_LEAVES; ++w2) x[w2] = stg[((size_t)w2 * KP + col) * MT_NS + jx];
#pragma unroll
            for (int w2 = 0; w2 < MT_LEAVES; w2 += 2) x[w2] = x[w2] + x[w2 + 1];
#pragma unroll
            for (int w2 = 0; w2 < MT_LEAVES; w2 += 4) x[w2] = x[w2] + x[w2 + 2];
#pragma unroll
            for (int w2 = 0; w2 < MT_LEAVES; w2 += 8) x[w2] = x[w2] + x[w2 + 4];
            slots[jx].accR[col] = x[0] + x[8];
        }
        __syncthreads();
        MT_LAP(4);
    }
#ifdef TSF_MFMA_TIMING
    if (mt.dbg && lane == 0)
        for (int k_ = 0; k_ < 6; ++k_) mt.dbg[((size_t)blockIdx.x * MT_NW + wid) * 6 + k_] = mt_acc[k_];
#endif
}

}  // namespace tsf
