// tsf_coop_kernels.h -- the cooperative tail of a residual-form L-BFGS launch.
//
// fit_kernel (tsf_fit_kernels.h) runs one wavefront per series; a launch of the reference's own
// model (/root/reference/src/jobs/prophet_modeler.py:65-66: logistic growth, multiplicative
// seasonality) ends with a few series that need 10-50 x the mean number of evaluations, each
// evaluated by ONE wave at ~40 k cycles per evaluation while 255 CUs idle.  Here a suspended
// series (checkpoint written by fit_kernel, see coop_checkpoint) is continued by a whole workgroup
// of NW waves on one CU:
//
//   wave 0, the OWNER, runs the same L-BFGS state machine as fit_kernel from the restored state;
//   an evaluation is split over the waves in the order eval_fg fixes (canonical arithmetic,
//   tsf_common.h) -- same operations, same operand order, same bits:
//     A  rows: step q of the 64 chunks (lane = chunk) -> wave 1 + q mod (NW-2): X.beta chain, trend,
//        r, r g, v of the row into LDS;
//     B  design column j -> wave 1 + j mod (NW-2): the chunk-partial fma chain over the steps (last
//        row first) and the 32,16,1,2,4,8 butterfly over the chunks; wave NW-1: the trend sums with
//        their changepoint-row snapshots and the suffix scans; the owner: the sum of squares;
//     C  the owner: eval_tail (priors, reverse sweep through the gamma recurrence, gradient).
//   Three workgroup barriers per evaluation; nothing but the design matrix (L2) is read from
//   global memory inside the loop.
#pragma once
#include "tsf_fit_kernels.h"

namespace tsf {

constexpr int COOP_NTB = 16;    // steps per chunk the register-resident loops hold (series of at most 1024 rows)

// ---- LDS of a cooperative workgroup -----------------------------------------------------------
template <int KP, int PPL>
struct CoopLds {
    WaveLds<KP, PPL> w;         // the owner's tables: theta, segment tables, time-axis sums, history
    int cmd, item;
    long long *dbg;             // (dev timing builds) per-series cycle counters, 16 per series
    // trend wave, NT <= COOP_NTB: where changepoint j's snapshot of the running trend sums is taken
    // (step and chunk of the first row at or after the changepoint), and the running sums of an
    // evaluation after every step
    int snap_q[NTAB], snap_l[NTAB];
    double run1[COOP_NTB * W], run2[COOP_NTB * W];
    // hand-overs inside an evaluation: 1 / sigma^2 (owner -> trend wave), the trend part of the gradient
    // without its prior terms (trend wave -> owner: [0] d/dk, [1] d/dm, [3 + j] d/d delta_j)
    double inv_s2;
    double gtr[W];
    // sparse indicator columns (fit_coop_kernel<..., SPARSE>; SP_* in tsf_common.h): the lanes' entry words, last row first
    // (filled by the whole workgroup when it takes a series), the slots the rows write, the columns' fold programs
    unsigned short sp_list[(SP_M + 1) * W];
    double sp_acc[SP_MAXC * SP_E];
    unsigned long long sp_prog[SP_MAXC];
};
enum { COOP_EVAL = 1, COOP_EXIT = 2 };
#ifndef COOP_NW
#define COOP_NW 8               // waves per cooperative workgroup (one workgroup per CU, 256 VGPRs per wave)
#endif

// LDS of the resident rows' first XL values: [step][XL][64]
// (table variant: 14 values of each of 12 x 64 rows; base-pair variant: the owner's 4 columns, then 2 of the 4 columns of each
// of the 6 row waves, 12 steps each)
__host__ __device__ constexpr size_t coop_xl_bytes(int KP, int NTmax) { return (KP == 28 && NTmax <= 12) ? sizeof(double) * 12 * 16 * W : 0; }
// row buffers r, r g, v: [rows][64] each, rows = max(NTmax, COOP_NTB) (coop_rb_rows)
__host__ __device__ constexpr int coop_rb_rows(int NTmax) { return NTmax > 16 ? NTmax : 16; }
// fit_coop_kernel<SP_DENSE, ..., PPL, ..., HARM, SPARSE>: the 64-column model's LDS tables, 12 rows of row buffers, the
// 28-column kernel's LDS columns
template <int PPL>
__host__ __device__ constexpr size_t coop_sparse_lds_bytes()
{
    return sizeof(CoopLds<64, PPL>) + sizeof(double) * 3 * 12 * W + coop_xl_bytes(SP_DENSE, 12);
}
template <int KP, int PPL>
__host__ __device__ constexpr size_t coop_lds_bytes(int NTmax)
{
    return sizeof(CoopLds<KP, PPL>) + sizeof(double) * 3 * (size_t)coop_rb_rows(NTmax) * W + coop_xl_bytes(KP, NTmax);
}

#ifdef TSF_COOP_TIMING      // dev only: owner-wave cycles per phase, summed per series
#define CT_DECL long long ct_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long ct_t0 = __builtin_readcyclecounter(), ct_start = ct_t0
#define CT_LAP(k) do { const long long t_ = __builtin_readcyclecounter(); ct_acc[k] += t_ - ct_t0; ct_t0 = t_; } while (0)
#define CT_FLUSH(dst, n) do { if ((dst) && lane_id() == 0) { ct_acc[7] = __builtin_readcyclecounter() - ct_start; for (int k_ = 0; k_ < 8; ++k_) ((long long *)(dst))[(size_t)(n) * 64 + k_] = ct_acc[k_]; for (int k_ = 8; k_ < 12; ++k_) ((long long *)(dst))[(size_t)(n) * 64 + 40 + k_] = ct_acc[k_]; } } while (0)
#define CT_ARGS , long long (&ct_acc)[12], long long &ct_t0
#define CT_PASS , ct_acc, ct_t0
#else
#define CT_ARGS
#define CT_PASS
#define CT_DECL do { } while (0)
#define CT_LAP(k) do { } while (0)
#define CT_FLUSH(dst, n) do { } while (0)
#endif

// ---- phase A: the rows of steps q0, q0 + qstep, ... -------------------------------------------
// eval_fg's row body (X.beta chain over the columns from 0, trend, mu, r) with beta read from the
// owner's LDS copy of theta; r, r g and v of every (step, chunk) go to rbR / rbU / rbV (0 for the
// rows past the end of a chunk: fma(x, 0, acc) leaves the chains of phase B unchanged).
template <int KP, int GROWTH, int MODE, int PPL, bool XIDX, class L>
__device__ __forceinline__ void coop_rows(const DevSpec *__restrict__ sp, const SeriesView &sv, const L &w,
                                          double *rbR, double *rbU, double *rbV, int q0, int qstep)
{
    const int lane = lane_id();
    const int S = sv.S, NT = sv.NT, K = sp->K;
    const int Ka = (MODE == 0) ? K : (MODE == 1 ? 0 : sp->Ka);
    const double *beta = &w.th[3 + S];
    for (int q = q0; q < NT; q += qstep) {
        const int idx = q * W + lane;
        double r = 0.0, rg = 0.0, v = 0.0;
        if (q < sv.cnt) {
            const unsigned cwv = (unsigned)sv.cw[idx];
            const int c = (int)(cwv & 0xffu);
            const double ti = sv.tw[idx];
            const double yi = sv.yw[idx];
            constexpr int XS = XIDX ? 1 : W;
            const double *xp = XIDX ? sv.Xu + (size_t)sv.uw[idx] * KP : sv.Xw + (size_t)q * KP * W + lane;
            double xa = 0.0, xm = 0.0;
            // columns beyond K are zero columns with zero coefficients: fma(0, 0, x) = x
            if (MODE == 0) {
#pragma unroll 8
                for (int j = 0; j < K; ++j) xa = __builtin_fma(xp[j * XS], beta[j], xa);
            } else if (MODE == 1) {
#pragma unroll 8
                for (int j = 0; j < K; ++j) xm = __builtin_fma(xp[j * XS], beta[j], xm);
            } else {
#pragma unroll 4
                for (int j = 0; j < Ka; ++j) xa = __builtin_fma(xp[j * XS], beta[j], xa);
#pragma unroll 4
                for (int j = Ka; j < K; ++j) xm = __builtin_fma(xp[j * XS], beta[j], xm);
            }
            const double ksc = w.ks[c], mcc = w.mc[c];
            double gtr, qv = 0.0;
            if (GROWTH == 0) {
                gtr = __builtin_fma(ksc, ti, mcc);
            } else {
                const double z = ksc * (ti - mcc);
                const double e = dm_exp_sel(-z);
                const double sg = 1.0 / (1.0 + e);
                gtr = sv.cap * sg;
                qv = gtr * (1.0 - sg);
            }
            const double opm = 1.0 + xm;
            const double mu = __builtin_fma(gtr, opm, xa);
            r = yi - mu;
            rg = r * gtr;
            v = r * opm;
            if (GROWTH == 1) v = v * qv;
        }
        rbR[idx] = r; rbU[idx] = rg; rbV[idx] = v;
    }
}

// sum over the 64 chunk partials of ONE column, butterfly offsets 32, 16, 1, 2, 4, 8 (column_sums
// for a single register: every lane ends with the sum)
__device__ __forceinline__ double chunk_sum_1(double v)
{
    { double x = v, y = v; swap32(x, y); v = x + y; }
    { double x = v, y = v; swap16(x, y); v = x + y; }
    return row_bfly_sum(v);
}

// ---- phase B -----------------------------------------------------------------------------------
// design columns j0, j0 + jstep, ...: acc_j = fma chain over the steps of a chunk, last row first,
// against r (additive column) or r g (multiplicative column), then the butterfly over the chunks
template <int KP, int MODE, bool XIDX, class L>
__device__ __forceinline__ void coop_columns(const DevSpec *__restrict__ sp, const SeriesView &sv, L &w,
                                             const double *rbR, const double *rbU, int j0, int jstep)
{
    const int lane = lane_id();
    const int NT = sv.NT, K = sp->K;
    const int Ka = (MODE == 0) ? K : (MODE == 1 ? 0 : sp->Ka);
    for (int j = j0; j < K; j += jstep) {
        const double *ru = (MODE == 0 || (MODE == 2 && j < Ka)) ? rbR : rbU;
        double acc = 0.0;
        int q = NT - 1;
        // four steps' loads in flight at a time
        for (; q >= 3; q -= 4) {
            double xv[4], rv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int qq = q - u, idx = qq * W + lane;
                rv[u] = ru[idx];
                if (XIDX) xv[u] = (qq < sv.cnt) ? sv.Xu[(size_t)sv.uw[idx] * KP + j] : 0.0;
                else xv[u] = sv.Xw[((size_t)qq * KP + j) * W + lane];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_fma(xv[u], rv[u], acc);
        }
        for (; q >= 0; --q) {
            const int idx = q * W + lane;
            double xv;
            if (XIDX) xv = (q < sv.cnt) ? sv.Xu[(size_t)sv.uw[idx] * KP + j] : 0.0;
            else xv = sv.Xw[((size_t)q * KP + j) * W + lane];
            acc = __builtin_fma(xv, ru[idx], acc);
        }
        acc = chunk_sum_1(acc);
        if (lane == 0) w.accR[j] = acc;
    }
}

// per-chunk trend sums with their values at the changepoint rows, then the suffix sums over the
// chunks (eval_fg: rt1, rt2, tp1, tp2, tot1, tot2)
template <class L>
__device__ __forceinline__ void coop_trend(const SeriesView &sv, L &w, const double *rbV)
{
    const int lane = lane_id();
    double rt1 = 0.0, rt2 = 0.0;
    for (int q = sv.NT - 1; q >= 0; --q) {
        if (q < sv.cnt) {
            const int idx = q * W + lane;
            const unsigned cwv = (unsigned)sv.cw[idx];
            const int c = (int)(cwv & 0xffu), cprev = (int)(cwv >> 8);
            const double ti = sv.tw[idx];
            const double v = rbV[idx];
            rt1 = __builtin_fma(v, ti, rt1);
            rt2 = rt2 + v;
            for (int j = cprev; j < c; ++j) { w.tp1[j] = rt1; w.tp2[j] = rt2; }
        }
    }
    const double s1 = suffix_scan(rt1), s2v = suffix_scan(rt2);
    w.tot1[lane] = s1; w.tot2[lane] = s2v;
    if (lane == 0) { w.tot1[W] = 0.0; w.tot2[W] = 0.0; }
}

__device__ __forceinline__ double coop_sse(const SeriesView &sv, const double *rbR)
{
    const int lane = lane_id();
    double sse = 0.0;
    for (int q = sv.NT - 1; q >= 0; --q) {
        if (q < sv.cnt) { const double r = rbR[q * W + lane]; sse = __builtin_fma(r, r, sse); }
    }
    return bfly_sum(sse);
}

// ---- pieces of an evaluation that moved off the owner ---------------------------------------------
// Segment tables ks[c], mc[c] (segment_tables of tsf_fit_kernels.h: the sequential recurrences over
// the changepoints) from the owner's LDS copy of theta, by ONE lane of the trend wave: operands come
// from LDS instead of v_readlane and every step stores its value, so a step is its dependent adds /
// multiplies and nothing else (the lane-parallel form costs ~10 vector instructions per step: the
// owner spent 4.4 k cycles here per evaluation).  Same operations, same operands, same order.
// Scratch: w.d1, w.d2 (free until the tail).
template <int GROWTH, class L>
__device__ __forceinline__ void coop_segment_tables(const SeriesView &sv, L &w)
{
    const int lane = lane_id();
    const int S = sv.S;
    // The chains run in blocks of 8 steps: the 8 operands of a block are read together (one LDS
    // latency per block instead of one per step -- a rolled loop waits for every read), then the 8
    // dependent steps and their stores.  Steps past S (the last block) work on whatever follows the
    // deltas in LDS and write entries ks / mc [S+1 .. ] that nothing reads.
    if (GROWTH == 0) {
        // the product of step j, lane-parallel: (-t_change[j]) * delta[j]
        w.d1[lane] = (lane < S) ? (-sv.tc_l) * w.th[3 + lane] : 0.0;
        wave_sync();
        if (lane == 0) {
            double ksv = w.th[0], mcv = w.th[1];
            w.ks[0] = ksv; w.mc[0] = mcv;
            for (int j0 = 0; j0 < S; j0 += 8) {
                double dj[8], pj[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { dj[u] = w.th[3 + j0 + u]; pj[u] = w.d1[j0 + u]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    ksv = ksv + dj[u];
                    mcv = mcv + pj[u];
                    w.ks[j0 + u + 1] = ksv; w.mc[j0 + u + 1] = mcv;
                }
            }
        }
    } else {
        // logistic (round 5): the chains as scans over the parameter lanes (logistic_tables_lanes, tsf_fit_kernels.h) --
        // theta from the owner's LDS copy, lane p = parameter p
        double ksn, mcn;
        logistic_tables_lanes(w.th[0], w.th[1], w.th[lane], sv.tcp_l[0], S, ksn, mcn);
        if (lane >= 3 && lane < 3 + S) { w.ks[lane - 2] = ksn; w.mc[lane - 2] = mcn; }
        if (lane == 0) { w.ks[0] = w.th[0]; w.mc[0] = w.th[1]; }
    }
    wave_sync();
}

// The trend part of the gradient (eval_tail's GROWTH block and its delta / k / m entries, WITHOUT the
// prior terms) from the time-axis sums the trend wave has just produced: cl.gtr.  Runs on the trend
// wave while the other waves finish their columns.  Scratch: w.d1, w.d2, w.rb, w.ab.
template <int GROWTH, int KP, int PPL>
__device__ __forceinline__ void coop_tail_trend(const SeriesView &sv, CoopLds<KP, PPL> &cl, const LogisticReversePre *pre = nullptr)
{
    const int lane = lane_id();
    const int S = sv.S;
    auto &lds = cl.w;
    const double nis = -cl.inv_s2;
    const double TA = lds.tot1[0], TB = lds.tot2[0];
    double gk, gm, gd = 0.0;
    if (GROWTH == 1) {
        // (round 5) the reverse sweep and the running sums as scans: logistic_reverse_lanes, tsf_fit_kernels.h
        double gd_l;
        // (pre: the quotients of the segment tables, formed by the caller while the rows were still running)
        if (pre) logistic_reverse_post(sv, lds, *pre, TA, TB, gk, gm, gd_l);
        else logistic_reverse_lanes(sv, lds, TA, TB, gk, gm, gd_l);
        gk = nis * gk; gm = nis * gm;
        if (lane >= 3 && lane < 3 + S) gd = nis * gd_l;
    } else {
        gk = nis * TA;
        gm = nis * TB;
        if (lane >= 3 && lane < 3 + S) {
            const int j = lane - 3, Lj = sv.Ljp_l[0];
            const double SA = lds.tp1[j] + lds.tot1[Lj + 1];
            const double SB = lds.tp2[j] + lds.tot2[Lj + 1];
            gd = nis * (SA - sv.tcp_l[0] * SB);
        }
    }
    cl.gtr[lane] = (lane == 0) ? gk : (lane == 1 ? gm : gd);
}

// ---- the helpers' loops ------------------------------------------------------------------------
// Workgroup barrier that waits for this wave's LDS traffic only: global loads issued before it (the
// design values of the next phase) stay in flight across it.  __syncthreads() would drain them
// (its release fence is s_waitcnt vmcnt(0)).  Everything the waves exchange goes through LDS.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// X.beta chains of R rows side by side over the Fourier columns (base pairs bp[i][se]; coefficients bl[j] in LDS):
// harm_columns' recurrence and eval_fg's fma order, harmonic by harmonic.  The coefficients come in batches of 8 columns,
// two batches ahead of their use (32 registers instead of 2 KF).  The empty asm statements tie the recurrence to the
// chain and the batches to their place: left free, the scheduler runs the recurrences of all rows ahead of the
// coefficient reads and holds every harmonic of every row in registers (measured: 165 registers spilled, 12 instead of
// 6 us per evaluation).
template <int O, int COL0, int R, int SE, int NSX, int KFX>
__device__ __forceinline__ void harm_chain_season(const double2 (&bp)[R][NSX], const double *bl, double (&bj)[KFX], double (&ch)[R])
{
    if constexpr (O > 0) {
        double c2[R], sp[R], cp[R], sc[R], cc[R];
#pragma unroll
        for (int i = 0; i < R; ++i) { c2[i] = 2.0 * bp[i][SE].y; sp[i] = 0.0; cp[i] = 1.0; sc[i] = bp[i][SE].x; cc[i] = bp[i][SE].y; }
#pragma unroll
        for (int h = 1; h <= O; ++h) {
            constexpr int B = 8;
            const int col = COL0 + 2 * (h - 1);
            if (col % B == 0 && col > 0 && col + B < KFX) {     // the batch after next
#pragma unroll
                for (int j = col + B; j < col + 2 * B; ++j) if (j < KFX) bj[j] = bl[j];
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                if (h > 1) {
                    const double sn = __builtin_fma(c2[i], sc[i], -sp[i]), cn = __builtin_fma(c2[i], cc[i], -cp[i]);
                    sp[i] = sc[i]; cp[i] = cc[i]; sc[i] = sn; cc[i] = cn;
                }
                ch[i] = __builtin_fma(sc[i], bj[col], ch[i]);
                ch[i] = __builtin_fma(cc[i], bj[col + 1], ch[i]);
            }
#pragma unroll
            for (int i = 0; i < R; ++i) asm volatile("" : "+v"(ch[i]), "+v"(sc[i]), "+v"(cc[i]));
        }
    }
}
template <int HARM, int R>
__device__ __forceinline__ void harm_chain_rows(const double2 (&bp)[R][harm_ns(HARM)], const double *bl, double (&ch)[R])
{
    constexpr int KF = harm_kf(HARM), NS = harm_ns(HARM), O0 = harm_order(HARM, 0), O1 = harm_order(HARM, 1), O2 = harm_order(HARM, 2);
    double bj[KF];
#pragma unroll
    for (int j = 0; j < 16; ++j) if (j < KF) bj[j] = bl[j];
    asm volatile("" ::: "memory");
    harm_chain_season<O0, 0, R, 0, NS, KF>(bp, bl, bj, ch);
    harm_chain_season<O1, 2 * O0, R, 1, NS, KF>(bp, bl, bj, ch);
    harm_chain_season<O2, 2 * (O0 + O1), R, 2 < NS ? 2 : 0, NS, KF>(bp, bl, bj, ch);
}

// generic loops (any NT <= COOP_MAX_NT): every load inside its phase
template <int KP, int GROWTH, int MODE, int PPL, int NW, bool XIDX>
__device__ __forceinline__ void coop_helper(const DevSpec *__restrict__ sp, const SeriesView &sv,
                                            CoopLds<KP, PPL> &cl, double *rbR, double *rbU, double *rbV, int wid)
{
    for (;;) {
        lds_barrier();                                      // A: the owner has published theta / cmd
        if (cl.cmd == COOP_EXIT) break;
        if (wid == NW - 1) coop_segment_tables<GROWTH>(sv, cl.w);
        lds_barrier();                                      // A2: segment tables, 1 / sigma^2
        if (wid < NW - 1) coop_rows<KP, GROWTH, MODE, PPL, XIDX>(sp, sv, cl.w, rbR, rbU, rbV, wid - 1, NW - 2);
        lds_barrier();                                      // B: rows complete
        if (wid == NW - 1) { coop_trend(sv, cl.w, rbV); wave_sync(); coop_tail_trend<GROWTH>(sv, cl); }
        else coop_columns<KP, MODE, XIDX>(sp, sv, cl.w, rbR, rbU, wid - 1, NW - 2);
        lds_barrier();                                      // C: sums complete
    }
}

// ---- helpers for series of at most NTB <= COOP_NTB steps per chunk ------------------------------
// Roles: waves 1 .. NW-2 take the rows (row q -> wave 1 + q mod (NW-2), RPW rows of a wave side by
// side: independent chains); TREND = wave NW-1 runs the trend sums.  Design columns:
//   RES (resident): the design values a wave needs -- its rows for phase A, its columns x all steps
//     for phase B -- are loaded ONCE per series and stay in registers (8 waves x 256 registers hold the
//     panel's design matrix twice: 2 x 152 KB for 730 x 26; a CU's L1 fills at 64 B per cycle, so
//     streaming both copies from L2 costs ~5 k cycles per evaluation); column j -> wave j mod (NW-1): the
//     OWNER (wave 0, idle while the sums are formed) takes a share, the trend wave none (it forms the trend
//     part of the gradient meanwhile).  An evaluation touches LDS only.
//   otherwise (wider models / longer series): the rows of phase A are requested before barrier A (they
//     arrive while the owner runs the optimiser), the columns of phase B CB at a time, the first batch
//     inside phase A; column j -> wave 1 + j mod (NW-2).
// Straight-line code: requests that have no row / column / step behind them are clamped to row 0 /
// column 0 / step 0 and their results dropped or multiplied by the zero rows [NT, COOP_NTB) of the row
// buffers (cleared once per series by the caller).
template <int KP, int NW, int NTB, int HARM = 0>
struct CoopShape {
    static constexpr int RPW = 2;                               // rows of a wave held at a time
    static constexpr int XB = KP <= 32 ? KP : 32;               // design values of a row held at a time
    static constexpr int NXB = KP / XB;                         // KP = 8, 16, 28: 1;  64: 2
    // columns per wave, resident mode: owner + row waves; HARM: the row waves alone (the registers the rows' design values
    // took hold a fifth column per wave, and the owner's evaluation has no loads left: they made it the last to arrive at
    // a barrier, measured)
    static constexpr bool OWNER_COLS = HARM == 0;
    // HARM, 28 columns: four per row wave (24), the last four summed by the owner from an LDS copy (OWN_L; the staging
    // region of the table variant, unused here) -- a fifth column per row wave spilled 25 registers into its loop
    static constexpr int OWN_L = (HARM != 0 && KP == 28) ? KP - 4 * (NW - 2) : 0;
    static constexpr int CPW_RES = OWNER_COLS ? (KP + NW - 2) / (NW - 1) : (OWN_L > 0 ? 4 : (KP + NW - 3) / (NW - 2));
    // resident mode, 28 columns x 12 steps: the first XL design values of every row live in LDS instead
    // of registers (with all of 2 rows x 28 + 4 columns x 12 values in registers the row waves spill
    // ~30 of them, and every reload is a dependent scratch round trip inside the row chain: 7.4 k
    // instead of ~2 k cycles per phase A)
    // HARM (round 5): a row is its base pairs (FitArgs::Bw; NS double2 per row, resident) and the Fourier columns are
    // expanded in registers inside the X.beta chain -- no design values of the rows in registers or LDS at all, and the
    // coefficients of an evaluation fit the registers that frees (one batch of LDS reads instead of one per fma)
    static constexpr int XL = (HARM == 0 && KP == 28 && NTB == 12) ? 14 : 0;
    static constexpr bool RES = NXB == 1 && (HARM != 0 ? 0 : RPW * (XB - XL)) + CPW_RES * NTB <= 76;
    static constexpr int NCW = (RES && OWNER_COLS) ? NW - 1 : NW - 2;           // waves that take columns
    static constexpr int CPW = (KP + NCW - 1) / NCW;
    static constexpr int CB = RES ? CPW : 3;                    // columns held at a time
    static constexpr int NCB = (CPW + CB - 1) / CB;
};
// SPARSE (with HARM, 28-column registers on the 64-column model's tables: BASELINE cfg4's stragglers): the columns from
// the 29th on are 0 / 1 indicators kept as entry words per lane (eval_fg<..., SPARSE>): a row adds the coefficients of its
// ones to its chain and writes its r (or r g) into the slot of each; the owner folds the slots per column after barrier B.
template <int KP, int GROWTH, int MODE, int PPL, int NW, bool XIDX, int NTB, bool TREND, int HARM = 0, bool SPARSE = false, class CLT>
__device__ __forceinline__ void coop_helper_pf(const DevSpec *__restrict__ sp, const SeriesView &sv,
                                               CLT &cl, double *rbR, double *rbU, double *rbV, double *xl, int wid)
{
    constexpr int XC = SPARSE ? 64 : KP;                // columns per row of the design table
    static_assert(!SPARSE || (HARM != 0 && KP == SP_DENSE && NTB == 12 && MODE != 2), "sparse columns: base-pair rows of the 28-column kernel");
#ifdef TSF_COOP_TIMING
    long long ht_a = 0, ht_b = 0, ht_t = 0;     // busy cycles of this wave in phase A / phase B
    long long htx[8] = {0, 0, 0, 0, 0, 0, 0, 0}, htx_t = 0;    // finer laps of the same wave (dbg[16 ..] row wave 1, dbg[24 ..] trend wave)
#define HT_START() ht_t = __builtin_readcyclecounter()
#define HT_STOP(acc) acc += __builtin_readcyclecounter() - ht_t
#define HX_START() htx_t = __builtin_readcyclecounter()
#define HX_LAP(k) do { const long long t_ = __builtin_readcyclecounter(); htx[k] += t_ - htx_t; htx_t = t_; } while (0)
#else
#define HT_START() do { } while (0)
#define HT_STOP(acc) do { } while (0)
#define HX_START() do { } while (0)
#define HX_LAP(k) do { } while (0)
#endif
    using SH = CoopShape<KP, NW, NTB, HARM>;
    constexpr int NRW = NW - 2, RPW = SH::RPW, XB = SH::XB, NXB = SH::NXB, CB = SH::CB, NCB = SH::NCB, NCW = SH::NCW;
    constexpr bool RES = SH::RES;
    static_assert(HARM == 0 || (RES && MODE != 2 && NXB == 1), "base-pair rows: resident mode, one column mode");
    static_assert(!(HARM != 0 && XIDX) || !SPARSE, "base pairs of a timestamp lattice: no sparse columns");
    // base-pair rows, 28 columns: two of the wave's four columns live in LDS (behind the owner's columns in xl) instead of
    // registers -- with all four in registers the per-series constants of the rows were reloaded from scratch in every
    // evaluation (22 scratch reads per evaluation of a row wave)
    constexpr int CL = (HARM != 0 && SH::OWN_L > 0 && !TREND) ? 2 : 0, CBR = CB - CL;
    double *const xcl = xl + (size_t)12 * SH::OWN_L * W + (size_t)(wid > 0 ? wid - 1 : 0) * CL * 12 * W;
    constexpr int HKF = HARM != 0 ? harm_kf(HARM) : 1, HNS = HARM != 0 ? harm_ns(HARM) : 1;
    constexpr int XL = RES ? SH::XL : 0;                // design values of a row kept in LDS (xl)
    constexpr bool ROWS = !TREND, COLS = !TREND;
    const int cw0 = (RES && SH::OWNER_COLS) ? wid : wid - 1;    // first column of this wave (resident mode without base-pair rows: wave 0 = the owner has one too)
    const int lane = lane_id();
    const int NT = sv.NT, S = sv.S, K = sp->K;
    const int Ka = (MODE == 0) ? K : (MODE == 1 ? 0 : sp->Ka);
    auto &w = cl.w;
    // per-series constants of the wave's rows
    bool row[RPW], valid[RPW];
    int idx[RPW], cq[RPW], qrow[RPW];
    double tq[RPW], yq[RPW];
    const double *xg[RPW];          // XIDX: the lane's gathered design row
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int q = (wid - 1) + i * NRW;
        row[i] = ROWS && q < NT;                        // this wave has such a row (wave-uniform)
        valid[i] = row[i] && q < sv.cnt;                // ... and it exists in this lane's chunk
        qrow[i] = row[i] ? q : 0;
        idx[i] = qrow[i] * W + lane;
        cq[i] = (int)((unsigned)sv.cw[idx[i]] & 0xffu);
        tq[i] = sv.tw[idx[i]]; yq[i] = sv.yw[idx[i]];
        xg[i] = XIDX ? sv.Xu + (size_t)(valid[i] ? sv.uw[idx[i]] : 0) * KP : nullptr;
    }
    // the trend wave's segment indices and times of its 64 chunks, all steps
    unsigned cwq[TREND ? NTB : 1];
    double twq[TREND ? NTB : 1];
    if constexpr (TREND) {
        // once per series: the times of the wave's rows (0 where a chunk has no such row: fma(0, 0, s) = s)
        // and, per changepoint, the (step, chunk) of the row whose running sums eval_fg snapshots
        // (`for j in [cprev, c): tp[j] = running sums` at the row where the segment index passes j)
        if (lane < NTAB) { cl.snap_q[lane] = 0; cl.snap_l[lane] = 0; }
        wave_sync();
#pragma unroll
        for (int q = 0; q < NTB; ++q) {
            const int qi = q < NT ? q : 0;
            const bool ok = q < NT && q < sv.cnt;
            cwq[q] = (unsigned)sv.cw[qi * W + lane];
            const double tv = sv.tw[qi * W + lane];
            twq[q] = ok ? tv : 0.0;
            if (ok) {
                const int c = (int)(cwq[q] & 0xffu), cprev = (int)(cwq[q] >> 8);
                for (int j = cprev; j < c; ++j) { cl.snap_q[j] = q; cl.snap_l[j] = lane; }
            }
        }
        wave_sync();
    }
    // Addresses: a wave-uniform base (scalar registers) plus the lane's offset.  In the streaming mode
    // `z` is an opaque zero that keeps the compiler from hoisting ~50 loop-invariant 64-bit addresses
    // into vector registers for the whole series (the base is recomputed with scalar adds instead).
    auto load_row = [&](int i, int h, int z, double (&x)[XB]) {
        if (XIDX) {
#pragma unroll
            for (int j = 0; j < XB; j += 2) {
                const double2 v2 = reinterpret_cast<const double2 *>(xg[i])[(h * XB + j) >> 1];
                x[j] = v2.x; x[j + 1 < XB ? j + 1 : j] = v2.y;
            }
        } else {
            const double *base = sv.Xw + ((size_t)(qrow[i] + z) * KP + h * XB) * W;
#pragma unroll
            for (int j = 0; j < XB; ++j) x[j] = base[j * W + lane];
        }
    };
    // column values of batch b: xc[u][q] = X[q][j_u][chunk]
    auto load_cols = [&](int b, int z, double (&xc)[CB][NTB]) {
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            int j = cw0 + (b * CB + u) * NCW + z;
            j = j < K ? j : 0;
#pragma unroll
            for (int q = 0; q < NTB; ++q) {
                const int qc = q < NT ? q : 0;
                if (XIDX) xc[u][q] = (qc < sv.cnt) ? sv.Xu[(size_t)sv.uw[qc * W + lane] * KP + j] : 0.0;
                else xc[u][q] = (sv.Xw + ((size_t)qc * XC + j) * W)[lane];
            }
        }
    };

    double x[(ROWS && NXB == 1 && HARM == 0) ? RPW : 1][XB];
    double2 bpr[(ROWS && HARM != 0) ? RPW : 1][HNS];           // HARM: the base pairs of the wave's rows
    double xc[COLS ? CB : 1][NTB];
    double xc2[(COLS && !RES && NCB > 1) ? CB : 1][NTB];       // streaming mode: the next batch of columns
    if constexpr (ROWS && HARM != 0) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            // (a row the lane's chunk does not have: whatever Bw holds there -- its r, r g, v are replaced by zeros below)
            // (XIDX: the pairs of the row's lattice point, FitArgs::Bu; point 0 where the chunk has no such row)
            const double2 *bq = XIDX ? reinterpret_cast<const double2 *>(sv.Bu) + (size_t)(valid[i] ? sv.uw[idx[i]] : 0) * HNS
                                     : reinterpret_cast<const double2 *>(sv.Bw) + (size_t)qrow[i] * HNS * W + lane;
#pragma unroll
            for (int se = 0; se < HNS; ++se) bpr[i][se] = bq[se * (XIDX ? 1 : W)];
        }
    }
    // SPARSE: the dense columns between the Fourier block and the sparse ones (two of them), and where the lane's entry
    // list holds the ones of each of the wave's rows (the list is sorted by row, last row first: a run per row)
    constexpr int NXDS = SPARSE ? SP_DENSE - HKF : 1;
    double xds[(ROWS && SPARSE) ? RPW : 1][NXDS];
    int sp_e0[(ROWS && SPARSE) ? RPW : 1], sp_n[(ROWS && SPARSE) ? RPW : 1];
    if constexpr (ROWS && SPARSE) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
#pragma unroll
            for (int j = 0; j < NXDS; ++j) xds[i][j] = valid[i] ? (sv.Xw + ((size_t)qrow[i] * XC + HKF + j) * W)[lane] : 0.0;
            sp_e0[i] = 0; sp_n[i] = 0;
            for (int e = SP_M - 1; e >= 0; --e) {
                const unsigned wd = cl.sp_list[e * W + lane];
                if (valid[i] && wd != SP_END && (int)(wd & 127u) == qrow[i]) { sp_e0[i] = e; sp_n[i]++; }
            }
        }
    }
    if constexpr (RES && COLS) {
        if constexpr (ROWS && HARM == 0) {
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                load_row(i, 0, 0, x[i]);
                if constexpr (XL > 0) {
                    if (row[i]) {
#pragma unroll
                        for (int j = 0; j < XL; ++j) xl[(qrow[i] * XL + j) * W + lane] = x[i][j];
                    }
                }
            }
        }
        load_cols(0, 0, xc);
        if constexpr (CL > 0) {
#pragma unroll
            for (int u = CBR; u < CB; ++u)
#pragma unroll
                for (int q = 0; q < NTB; ++q) xcl[((u - CBR) * NTB + q) * W + lane] = xc[u][q];
        }
    }
    // one half (XB values) of a row's X.beta chain
    auto chain_half = [&](int i, int h, const double (&xv_)[XB], double &xa_, double &xm_) {
#pragma unroll
        for (int j = 0; j < XB; ++j) {
            const int jj = h * XB + j;
            const double bj = w.th[3 + S + jj];             // zero beyond the model's parameters
            // (resident mode: the first XL values of the row come back from LDS)
            const double xv = (j < XL) ? xl[(qrow[i] * XL + j) * W + lane] : xv_[j];
            if (MODE == 0) xa_ = __builtin_fma(xv, bj, xa_);
            else if (MODE == 1) xm_ = __builtin_fma(xv, bj, xm_);
            else { if (jj < Ka) xa_ = __builtin_fma(xv, bj, xa_); else xm_ = __builtin_fma(xv, bj, xm_); }
            // (keeps the coefficient reads next to their use: hoisted to the top of the chain, the 28 /
            // 32 of them hold 56 / 64 registers beside the design values)
            if ((j & 3) == 3) asm volatile("" : "+v"(xa_), "+v"(xm_) :: "memory");
        }
    };
    double xw[(ROWS && NXB == 2) ? 3 : 1][XB];          // KP = 64: rotating half-row buffers
    for (;;) {
        int z = 0;
        // (an opaque zero per evaluation in the LDS addresses of the wave's columns: loop-invariant, the 24 addresses were
        // hoisted out of the evaluation loop into registers, spilled, and each read then waited for a scratch round trip)
        int zl = 0;
        if constexpr (CL > 0) asm volatile("v_mov_b32 %0, 0" : "=v"(zl));
        if constexpr (!RES) {
            asm volatile("s_mov_b32 %0, 0" : "=s"(z));
            if constexpr (ROWS && NXB == 1) {
#pragma unroll
                for (int i = 0; i < RPW; ++i) load_row(i, 0, z, x[i]);
            }
            if constexpr (ROWS && NXB == 2) { load_row(0, 0, z, xw[0]); load_row(0, 1, z, xw[1]); }
        }
        lds_barrier();                                      // A
        if (cl.cmd == COOP_EXIT) break;
        HT_START();
        HX_START();
        if constexpr (TREND) { coop_segment_tables<GROWTH>(sv, w); HX_LAP(0); lds_barrier(); HX_LAP(1); }    // A2 (trend wave)
        if constexpr (ROWS) {
            double xa[RPW], xm[RPW];
#pragma unroll
            for (int i = 0; i < RPW; ++i) { xa[i] = 0.0; xm[i] = 0.0; }
            if constexpr (HARM != 0) {
                // the chains of the wave's rows side by side: column j's value from the recurrence, fma'd in column order
                // (eval_fg HARM: same operands, same order)
                double ch[RPW];
#pragma unroll
                for (int i = 0; i < RPW; ++i) ch[i] = 0.0;
                harm_chain_rows<HARM, RPW>(bpr, &w.th[3 + S], ch);
                if constexpr (SPARSE) {
                    // eval_fg<..., SPARSE>: the dense columns behind the Fourier block, then the ones of the row in
                    // ascending column order: fma(1, b, chain) = chain + b
#pragma unroll
                    for (int i = 0; i < RPW; ++i) {
#pragma unroll
                        for (int j = 0; j < NXDS; ++j) ch[i] = __builtin_fma(xds[i][j], w.th[3 + S + HKF + j], ch[i]);
                        int k = 0;
                        while (__any(k < sp_n[i])) {
                            if (k < sp_n[i]) {
                                const unsigned wd = cl.sp_list[(sp_e0[i] + k) * W + lane];
                                ch[i] = ch[i] + w.th[3 + S + SP_DENSE + (int)((wd >> 7) & 63u)];
                            }
                            ++k;
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < RPW; ++i) { if (MODE == 0) xa[i] = ch[i]; else xm[i] = ch[i]; }
            } else if constexpr (NXB == 1) {
#pragma unroll
                for (int i = 0; i < RPW; ++i) chain_half(i, 0, x[i], xa[i], xm[i]);
            } else {
                // two rows x two halves of 32 values: the request of a half goes out while an earlier half
                // is being consumed (three buffers of 64 registers; the pins order request / chain / request)
                static_assert(NXB == 1 || RPW == 2, "the half-row rotation is written for two rows");
                load_row(1, 0, z, xw[2]);
                chain_half(0, 0, xw[0], xa[0], xm[0]);
                asm volatile("" : "+v"(xa[0]), "+v"(xm[0]) :: "memory");
                load_row(1, 1, z, xw[0]);
                chain_half(0, 1, xw[1], xa[0], xm[0]);
                asm volatile("" : "+v"(xa[0]), "+v"(xm[0]) :: "memory");
                chain_half(1, 0, xw[2], xa[1], xm[1]);
                chain_half(1, 1, xw[0], xa[1], xm[1]);
            }
            if constexpr (!RES) {
                // The column requests go out here: after the row values are consumed (the empty asm pins
                // the finished chains at this point -- otherwise the compiler sinks them below the requests
                // and both sets of design values hold registers together, the rest spills) and ahead of
                // the trend arithmetic that hides their latency.
#pragma unroll
                for (int i = 0; i < RPW; ++i) asm volatile("" : "+v"(xa[i]), "+v"(xm[i]));
                load_cols(0, z, xc);
            }
            HX_LAP(0);
            lds_barrier();                                  // A2: the trend wave's segment tables are in LDS
            HX_LAP(1);
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const double ksc = w.ks[cq[i]], mcc = w.mc[cq[i]];
                double gtr, qv = 0.0;
                if (GROWTH == 0) {
                    gtr = __builtin_fma(ksc, tq[i], mcc);
                } else {
                    const double z2 = ksc * (tq[i] - mcc);
                    const double e = dm_exp_sel(-z2);          // (dm_exp_sel_sc, literals as scalar operands: measured, 5.05 -> 5.3 us per evaluation)
                    const double sg = 1.0 / (1.0 + e);
                    gtr = sv.cap * sg;
                    qv = gtr * (1.0 - sg);
                }
                const double opm = 1.0 + xm[i];
                const double mu = __builtin_fma(gtr, opm, xa[i]);
                double r = yq[i] - mu;
                double rg = r * gtr;
                double v = r * opm;
                if (GROWTH == 1) v = v * qv;
                if constexpr (SPARSE) {
                    // fma(1, w, +0) = w: the lane's partial of each of the row's sparse columns, into its slot
                    int k = 0;
                    while (__any(k < sp_n[i])) {
                        if (k < sp_n[i]) {
                            const unsigned wd = cl.sp_list[(sp_e0[i] + k) * W + lane];
                            cl.sp_acc[((wd >> 7) & 63u) * SP_E + (wd >> 13)] = (MODE == 0) ? r : rg;
                        }
                        ++k;
                    }
                }
                // rows past the end of a chunk: zeros (fma(x, 0, acc) leaves the chains of phase B unchanged)
                r = valid[i] ? r : 0.0; rg = valid[i] ? rg : 0.0; v = valid[i] ? v : 0.0;
                if (row[i]) { rbR[idx[i]] = r; rbU[idx[i]] = rg; rbV[idx[i]] = v; }
            }
            // NT beyond RPW rows per wave: the remaining rows the plain way (none where RPW rows per wave cover NTB steps: the
            // loop would never run, but its registers -- a row of design values in flight -- were allocated across the
            // whole evaluation loop and pushed the rows' per-series constants into scratch)
            if constexpr (RPW * NRW < NTB)
                coop_rows<KP, GROWTH, MODE, PPL, XIDX>(sp, sv, w, rbR, rbU, rbV, (wid - 1) + RPW * NRW, NRW);
        }
        LogisticReversePre lrp;
        if constexpr (TREND && GROWTH == 1) logistic_reverse_pre(sv, w, lrp);      // while the rows run: what the reverse sweep needs of ks / mc alone
        HT_STOP(ht_a);
        HX_LAP(2);
        lds_barrier();                                      // B
        HX_LAP(3);
        HT_START();
        if constexpr (TREND) {
            // running trend sums of every chunk, last row first (rows a chunk does not have carry
            // v = 0 and t = 0: the sums pass through unchanged), kept per step; the snapshots at the
            // changepoint rows are then one gather
            double rt1 = 0.0, rt2 = 0.0;
            // (all reads of v ahead of the stores of the running sums: both are LDS, and a read behind a store that
            // may alias it waits for it -- twelve round trips, 1.1 k cycles per evaluation)
            double vq[NTB];
#pragma unroll
            for (int q = 0; q < NTB; ++q) vq[q] = rbV[q * W + lane];     // (zero rows beyond NT)
#pragma unroll
            for (int q = NTB - 1; q >= 0; --q) {
                rt1 = __builtin_fma(vq[q], twq[q], rt1);
                rt2 = rt2 + vq[q];
                cl.run1[q * W + lane] = rt1; cl.run2[q * W + lane] = rt2;
            }
            HX_LAP(4);
            const double s1 = suffix_scan(rt1), s2v = suffix_scan(rt2);
            w.tot1[lane] = s1; w.tot2[lane] = s2v;
            if (lane == 0) { w.tot1[W] = 0.0; w.tot2[W] = 0.0; }
            wave_sync();
            HX_LAP(5);
            if (lane < S) {
                const int at = cl.snap_q[lane] * W + cl.snap_l[lane];
                w.tp1[lane] = cl.run1[at]; w.tp2[lane] = cl.run2[at];
            }
            wave_sync();
            HX_LAP(6);
            coop_tail_trend<GROWTH>(sv, cl, GROWTH == 1 ? &lrp : nullptr);
            HX_LAP(7);
        }
        if constexpr (COLS) {
            const double *const xcb_l = xcl + lane + zl;        // (one base per evaluation; the reads below differ by immediates)
            auto col_batch = [&](int b, const double (&xcb)[CB][NTB]) {
                double acc[CB];
#pragma unroll
                for (int u = 0; u < CB; ++u) acc[u] = 0.0;
#pragma unroll
                for (int q = NTB - 1; q >= 0; --q) {
                    // (steps in [NT, NTB): zero rows of the buffers, fma(x, 0, acc) = acc)
                    const double r0 = (MODE == 1) ? 0.0 : rbR[q * W + lane];
                    const double r1 = (MODE == 0) ? 0.0 : rbU[q * W + lane];
#pragma unroll
                    for (int u = 0; u < CB; ++u) {
                        const int j = cw0 + (b * CB + u) * NCW;
                        const double ru = (MODE == 0) ? r0 : (MODE == 1 ? r1 : (j < Ka ? r0 : r1));
                        const double xv = (CL > 0 && u >= CBR) ? xcb_l[((u - CBR < 0 ? 0 : u - CBR) * NTB + q) * W] : xcb[u][q];
                        acc[u] = __builtin_fma(xv, ru, acc[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < CB; ++u) {
                    const int j = cw0 + (b * CB + u) * NCW;
                    const double sacc = chunk_sum_1(acc[u]);
                    if (j < K && lane == 0) w.accR[j] = sacc;
                }
            };
            if constexpr (RES || NCB == 1) {
                col_batch(0, xc);
            } else {
                // streaming: the request of batch b + 1 goes out before batch b is consumed (two buffers)
#pragma unroll
                for (int b = 0; b < NCB; ++b) {
                    if (b + 1 < NCB) { if ((b & 1) == 0) load_cols(b + 1, z, xc2); else load_cols(b + 1, z, xc); }
                    if ((b & 1) == 0) col_batch(b, xc); else col_batch(b, xc2);
                }
            }
        }
        HT_STOP(ht_b);
        if constexpr (!TREND) HX_LAP(4);
        lds_barrier();                                      // C
    }
#ifdef TSF_COOP_TIMING
    if (cl.dbg && lane == 0 && (wid == 1 || wid == NW - 1)) {
        cl.dbg[wid == 1 ? 8 : 10] = ht_a; cl.dbg[wid == 1 ? 9 : 11] = ht_b;
        for (int k_ = 0; k_ < 8; ++k_) cl.dbg[(wid == 1 ? 16 : 24) + k_] = htx[k_];
    }
    if (cl.dbg && lane == 0 && wid >= 1 && wid < NW - 1) { cl.dbg[32 + 2 * wid] = htx[0]; cl.dbg[33 + 2 * wid] = htx[2]; }
#endif
}

template <int KP, int GROWTH, int MODE, int PPL, int NW, bool XIDX, int NTB, int HARM = 0, bool SPARSE = false, class CLT>
__device__ __forceinline__ void coop_helper_ntb(const DevSpec *__restrict__ sp, const SeriesView &sv,
                                                CLT &cl, double *rbR, double *rbU, double *rbV, int wid)
{
    double *xl = rbV + (size_t)(SPARSE ? 12 : COOP_NTB) * W;    // (present when coop_xl_bytes() > 0: NTmax <= 12 < COOP_NTB rows of row buffers; SPARSE: 12 rows, coop_sparse_lds_bytes)
    // issue priority: the trend wave's chains (segment tables in phase A; running sums, scans and the reverse sweep in
    // phase B) are the longest of both phases and it shares its SIMD with a row wave (measured: 5.3 -> 5.05 us per evaluation)
    if (wid == NW - 1) __builtin_amdgcn_s_setprio(3);
    if (wid == NW - 1) coop_helper_pf<KP, GROWTH, MODE, PPL, NW, XIDX, NTB, true, HARM, SPARSE>(sp, sv, cl, rbR, rbU, rbV, xl, wid);
    else coop_helper_pf<KP, GROWTH, MODE, PPL, NW, XIDX, NTB, false, HARM, SPARSE>(sp, sv, cl, rbR, rbU, rbV, xl, wid);
}

// ---- one evaluation, the owner's side ----------------------------------------------------------
// columns the owner holds in resident mode (column j of wave-slot 0: j = 0, NW-1, 2 (NW-1), ...)
template <int KP, int NW, int HARM = 0>
struct CoopOwnerCols {
    // (the base-pair rows exist for series of at most 12 steps per chunk: the 16-step helpers keep reading the tables)
    static constexpr bool ANY = (CoopShape<KP, NW, 12, HARM>::RES && CoopShape<KP, NW, 12, HARM>::OWNER_COLS) || CoopShape<KP, NW, COOP_NTB, 0>::RES;
    static constexpr int OC = ANY ? CoopShape<KP, NW, 12, 0>::CPW_RES : 1;
    static constexpr int ONT = CoopShape<KP, NW, COOP_NTB, 0>::RES ? COOP_NTB : 12;     // steps the owner's columns span
};

template <int KP, int GROWTH, int MODE, int PPL, int NW, bool XIDX, int HARM = 0, bool SPARSE = false, class CLT>
__device__ __forceinline__ bool coop_eval_owner(const DevSpec *__restrict__ sp, SeriesView &sv,
                                                CLT &cl, double *rbR, double *rbU, double *rbV,
                                                const double (&th)[PPL], double &f_out, double (&g)[PPL],
                                                bool res, const double *xol CT_ARGS)
{
    constexpr int OC = CoopOwnerCols<KP, NW, HARM>::OC, ONT = CoopOwnerCols<KP, NW, HARM>::ONT;
    constexpr int OWN_L = CoopShape<KP, NW, 12, HARM>::OWN_L;      // columns the owner sums from its LDS copy xol[q][c][lane]
    const int lane = lane_id();
    const int S = sv.S, T = sv.T, K = sp->K;
    const int Ka = (MODE == 0) ? K : (MODE == 1 ? 0 : sp->Ka);
    auto &lds = cl.w;
    sv.n_eval++;
#pragma unroll
    for (int s = 0; s < PPL; ++s) lds.th[lane + s * W] = th[s];
    if (lane == 0) cl.cmd = COOP_EVAL;
    CT_LAP(1);
    lds_barrier();                                          // A: theta is out
    // Nothing the owner forms is needed before barrier B (1 / sigma^2: the trend wave, after B), so it passes A2 at once --
    // the row waves' chains and the trend wave's segment tables decide when A2 happens (measured: with its arithmetic
    // between A and A2 the owner arrived last, 2.7 k cycles after A) -- and does its own work (sigma terms, the two prior
    // sums, the prior terms of the gradient) while the rows run.  Its share of the design columns (table variant: column 0,
    // NW-1, ...) is requested before A2, every evaluation -- the loads stay in flight across the barrier and arrive during
    // the rows; requested after A2 they made the owner the last at B -- instead of holding ~100 registers across the
    // whole optimiser loop.
    double xo[OC][ONT];
#pragma unroll
    for (int u = 0; u < OC; ++u) {
        int j = u * (NW - 1);
        j = j < K ? j : 0;
#pragma unroll
        for (int q = 0; q < ONT; ++q) {
            const int qc = q < sv.NT ? q : 0;
            double xv = 0.0;
            if (res) {
                if (XIDX) xv = (qc < sv.cnt) ? sv.Xu[(size_t)sv.uw[qc * W + lane] * KP + j] : 0.0;
                else xv = (sv.Xw + ((size_t)qc * KP + j) * W)[lane];
            }
            xo[u][q] = xv;
        }
    }
    CT_LAP(8);
    lds_barrier();                                          // A2
    CT_LAP(9);
    // while the trend wave walks the changepoint recurrences and the row waves their X.beta chains:
    // sigma terms and the two prior sums (eval_tail's first block)
    const double k = theta_at<PPL>(th, 0), m = theta_at<PPL>(th, 1), ls = theta_at<PPL>(th, 2);
    const double sigma = dm_exp_sel(ls);
    const double inv_s2 = 1.0 / (sigma * sigma);
    if (lane == 0) cl.inv_s2 = inv_s2;
    double pa = 0.0, pb = 0.0;
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const int p = lane + s * W;
        if (p >= 3 && p < 3 + S) pa = pa + __builtin_fabs(th[s]);
        if (p >= 3 + S && p < sv.P) { const double qq = th[s] / sv.prior_l[s]; pb = __builtin_fma(qq, qq, pb); }
    }
    const double sabs = bfly_sum(pa), sb = bfly_sum(pb);
    const double s2 = sigma * sigma;
    double f = ((0.5 * k) * k) / 25.0 + ((0.5 * m) * m) / 25.0;
    f = f + sabs / sv.tau;
    f = f + 2.0 * s2;
    f = f + 0.5 * sb;
    f = f + (double)T * ls;
    // the prior terms of the gradient (divisions by constants): formed here, while the rows run, instead of in the tail
    double gpr[PPL];
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const int p = lane + s * W;
        double gv = 0.0;
        if (p == 0) gv = k / 25.0;
        else if (p == 1) gv = m / 25.0;
        else if (p == 2) gv = 4.0 * s2;
        else if (p < 3 + S) {
            const double dj = th[s];
            const double sgn = (double)((dj > 0.0) - (dj < 0.0));
            gv = sgn / sv.tau;
        } else if (p < sv.P) {
            const double pr = sv.prior_l[s];
            gv = th[s] / (pr * pr);
        }
        gpr[s] = gv;
    }
    CT_LAP(10);
    lds_barrier();                                          // B: rows complete
    CT_LAP(2);
    const double sse_t = coop_sse(sv, rbR);
    if constexpr (OWN_L > 0) {
        if (xol) {
            // base-pair rows: the columns behind the row waves' 24, from the owner's LDS copy (coop_owner), as coop_helper_pf sums its own
            const double *ru = (MODE == 0) ? rbR : rbU;
            double acc[OWN_L];
#pragma unroll
            for (int c = 0; c < OWN_L; ++c) acc[c] = 0.0;
#pragma unroll
            for (int q = 11; q >= 0; --q) {
                const double rv = ru[q * W + lane];         // (zero rows beyond NT)
#pragma unroll
                for (int c = 0; c < OWN_L; ++c) acc[c] = __builtin_fma(xol[(q * OWN_L + c) * W + lane], rv, acc[c]);
            }
#pragma unroll
            for (int c = 0; c < OWN_L; ++c) {
                const int j = KP - OWN_L + c;
                const double sacc = chunk_sum_1(acc[c]);
                if (j < K && lane == 0) lds.accR[j] = sacc;
            }
        }
    }
    if (res) {
        // the owner's share of the design columns (resident mode), as in coop_helper_pf
        double acc[OC];
#pragma unroll
        for (int u = 0; u < OC; ++u) acc[u] = 0.0;
#pragma unroll
        for (int q = ONT - 1; q >= 0; --q) {
            const double r0 = (MODE == 1) ? 0.0 : rbR[q * W + lane];
            const double r1 = (MODE == 0) ? 0.0 : rbU[q * W + lane];
#pragma unroll
            for (int u = 0; u < OC; ++u) {
                const int j = u * (NW - 1);
                const double ru = (MODE == 0) ? r0 : (MODE == 1 ? r1 : (j < Ka ? r0 : r1));
                acc[u] = __builtin_fma(xo[u][q], ru, acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < OC; ++u) {
            const int j = u * (NW - 1);
            const double sacc = chunk_sum_1(acc[u]);
            if (j < K && lane == 0) lds.accR[j] = sacc;
        }
    }
    if constexpr (SPARSE) {
        // the sparse columns' sums: lane c folds the slots of column c in the reduction network's order (eval_fg<..., SPARSE>)
        if (lane < sv.P - 3 - S - SP_DENSE) {
            const unsigned long long pg = cl.sp_prog[lane];
            const int nm = (int)(pg & 7u), root = (int)((pg >> 3) & 7u), kl = (int)((pg >> 6) & 15u);
            double *wsl = cl.sp_acc + lane * SP_E;
            for (int i = 0; i < nm; ++i) {
                const int d = (int)((pg >> (10 + 6 * i)) & 7u), r2 = (int)((pg >> (13 + 6 * i)) & 7u);
                wsl[d] = wsl[d] + wsl[r2];
            }
            lds.accR[SP_DENSE + lane] = (kl > 0 ? wsl[root] : 0.0) + 0.0;
        }
    }
    lds_barrier();                                          // C: all sums, and the trend part of the gradient
    CT_LAP(3);
    // the rest of eval_tail: f, and the gradient from cl.gtr (k, m, delta: trend wave) and accR (beta)
    f = f + (0.5 * sse_t) * inv_s2;
    const double nis = -inv_s2;
    bool bad = !finite_f64(f);
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const int p = lane + s * W;
        double gv = 0.0;
        if (p == 0) gv = cl.gtr[0] + gpr[s];
        else if (p == 1) gv = cl.gtr[1] + gpr[s];
        else if (p == 2) gv = ((double)T - sse_t * inv_s2) + gpr[s];
        else if (p < 3 + S) gv = cl.gtr[p] + gpr[s];
        else if (p < sv.P) gv = nis * lds.accR[p - 3 - S] + gpr[s];
        g[s] = gv;
        bad = bad || !finite_f64(gv);
    }
    f_out = f;
    const bool bad_ = __any(bad);
    CT_LAP(4);
    return bad_;
}

// ---- the owner: fit_kernel's L-BFGS loop, resumed at a line-search evaluation -----------------
// (Stan's BFGSMinimizer<LBFGSUpdate>::step / WolfeLineSearch / WolfLSZoom as restated in fit_kernel;
// the text below is that loop with the state restored from the checkpoint and eval_fg replaced by
// coop_eval_owner -- keep the two in step.)
template <int KP, int GROWTH, int MODE, int PPL, int NW, bool XIDX, int HARM = 0, bool SPARSE = false, class CLT>
__device__ __forceinline__ void coop_owner(const FitArgs &a, SeriesView &sv, int64_t n, const double *slot,
                                           CLT &cl, double *rbR, double *rbU, double *rbV, double *xl = nullptr)
{
    constexpr int XC = SPARSE ? 64 : KP;                // columns per row of the design table
    const int lane = lane_id();
    const DevSpec *sp = a.sp;
    auto &lds = cl.w;
    // slot == nullptr: the whole fit runs here (fit_coop_kernel in direct mode: no one-wave phase, no
    // checkpoint) from fbprophet's initial values, as fit_kernel starts it
    const bool scratch = slot == nullptr;
    CoopVars cv;
    double xk[PPL], gk[PPL], pk[PPL], xk1[PPL], gk1[PPL], pk1[PPL];
    const int H = a.opt.history > MAXH ? MAXH : a.opt.history;
    if (scratch) {
        const SeriesTab st = a.stab[n];
        if (lane == 0) {
            a.y_scale[n] = st.y_scale;
            if (!a.aligned || n == 0) a.grid_out[a.aligned ? 0 : n] = a.gtab[grid_index(a, n)].info;
        }
#pragma unroll
        for (int s = 0; s < PPL; ++s) {
            const int p = lane + s * W;
            xk[s] = (p == 0) ? st.k0 : (p == 1 ? st.m0 : 0.0);
            gk[s] = 0.0; pk[s] = 0.0; xk1[s] = xk[s]; gk1[s] = 0.0; pk1[s] = 0.0;
        }
        // (series that never reach the optimiser are reported by fit_coop_kernel itself: coop_report_unfitted)
        memset(&cv, 0, sizeof(cv));
        cv.alpha = a.opt.init_alpha; cv.gammak = 1.0;
    } else {
        cv = *reinterpret_cast<const CoopVars *>(slot);
        coop_get_vec<PPL>(slot, 0, xk); coop_get_vec<PPL>(slot, 1, gk); coop_get_vec<PPL>(slot, 2, pk);
#pragma unroll
        for (int s = 0; s < PPL; ++s) { xk1[s] = 0.0; gk1[s] = 0.0; pk1[s] = 0.0; }     // dead at a line-search evaluation
        for (int h = 0; h < H; ++h) {
            double sv_[PPL], yv_[PPL];
            coop_get_vec<PPL>(slot, 6 + h, sv_); coop_get_vec<PPL>(slot, 6 + MAXH + h, yv_);
#pragma unroll
            for (int s = 0; s < PPL; ++s) { lds.SY[((2 * h) * PPL + s) * W + lane] = sv_[s]; lds.SY[((2 * h + 1) * PPL + s) * W + lane] = yv_[s]; }
        }
        if (lane < MAXH) lds.rho[lane] = slot[COOP_VARS_D + lane];
        TSF_WAVE_SYNC();
    }

    const double c1 = 1e-4, c2 = 0.9, minAlpha = 1e-12, min_range = 1e-16;
    const int maxLSIts = 20, maxLSRestarts = 10;
    (void)c1; (void)c2;

    double fk = cv.fk, fk1 = cv.fk1, alpha = cv.alpha, gammak = cv.gammak;
    int itNum = cv.itNum, ret = 0, resetB = cv.resetB, hist_len = cv.hist_len, hist_head = cv.hist_head;
    double dfp = cv.dfp, c1dfp = cv.c1dfp, c2dfp = cv.c2dfp, alpha0 = cv.alpha0, prevF = cv.prevF, prevDFp = cv.prevDFp;
    double alo = cv.alo, aloF = cv.aloF, aloDFp = cv.aloDFp, ahi = cv.ahi, ahiF = cv.ahiF, ahiDFp = cv.ahiDFp;
    int nits = cv.nits, lsRestarts = cv.lsRestarts, zoom = cv.zoom, zit = cv.zit;
    double gp = cv.gp;
    bool gp_valid = cv.gp_valid != 0, pk1_scaled = cv.pk1_scaled != 0;
    sv.n_eval = cv.n_eval;

    // resident mode (as the helpers decide it, fit_coop_kernel): the owner then has a share of the columns
    // (base-pair rows, series of <= 12 steps: the row waves take every column, the owner none)
    const bool res = sv.NT <= COOP_NTB && (a.NTmax > 12 ? CoopShape<KP, NW, COOP_NTB, 0>::RES
                                                       : (CoopShape<KP, NW, 12, HARM>::RES && CoopShape<KP, NW, 12, HARM>::OWNER_COLS));

    // base-pair rows (series of <= 12 steps): the owner's columns, once per series, into the LDS region the table variant
    // stages row values in -- [step][column][lane]; rows a chunk does not have and columns beyond K are zeros in Xw
    const double *xol = nullptr;
    if constexpr (CoopShape<KP, NW, 12, HARM>::OWN_L > 0) {
        constexpr int OWN_L = CoopShape<KP, NW, 12, HARM>::OWN_L;
        if (xl && a.NTmax <= 12) {
            for (int q = 0; q < 12; ++q) {
                const int qc = q < sv.NT ? q : 0;
#pragma unroll
                for (int c = 0; c < OWN_L; ++c)
                    xl[(q * OWN_L + c) * W + lane] = XIDX ? ((q < sv.NT && qc < sv.cnt) ? sv.Xu[(size_t)sv.uw[qc * W + lane] * KP + (KP - OWN_L + c)] : 0.0)
                                                          : ((q < sv.NT) ? (sv.Xw + ((size_t)qc * XC + (KP - OWN_L + c)) * W)[lane] : 0.0);
            }
            TSF_WAVE_SYNC();
            xol = xl;
        }
    }

    enum { ST_INIT = 0, ST_START_ITER, ST_START_LS, ST_LS_PRE, ST_LS_EVAL };
    int stage = scratch ? ST_INIT : ST_LS_EVAL;
    CT_DECL;
    for (;;) {
        if (stage == ST_START_ITER) {
            itNum++;
            resetB = (itNum == 1) ? 1 : 0;
            stage = ST_START_LS;
        }
        if (stage == ST_START_LS) {
            if (resetB) {
#pragma unroll
                for (int s = 0; s < PPL; ++s) pk[s] = -gk[s];
                gp_valid = false;
            }
            if (!gp_valid) gp = pdot<PPL>(gk, pk);
            gp_valid = false;
            if (itNum > 1 && resetB != 2) {
                const double gp1 = pk1_scaled ? pdot<PPL>(gk1, pk1) : dfp;
                const double ci = cubic_interp6(gp1, alpha, fk - fk1, gp, minAlpha, 1.0);
                alpha = uniform_f64(__builtin_fmin(1.0, 1.01 * ci));
            } else {
                alpha = a.opt.init_alpha;
            }
            dfp = gp;
            c1dfp = uniform_f64(c1 * dfp); c2dfp = uniform_f64(c2 * dfp);
            alpha0 = minAlpha; prevF = fk; prevDFp = dfp;
            nits = 0; lsRestarts = 0; zoom = 0; zit = 0;
            stage = ST_LS_PRE;
        }
        bool ls_fail = false;
        if (stage == ST_LS_PRE) {
            if (!zoom) {
                if (nits >= maxLSIts) ls_fail = true;
            } else {
                zit++;
                if (__builtin_fabs(alo - ahi) < min_range) {
                    ls_fail = true;
                } else if (zit % 5 == 0) {
                    alpha = uniform_f64(0.5 * (alo + ahi));
                } else {
                    const double d1 = aloDFp + ahiDFp - 3.0 * (aloF - ahiF) / (alo - ahi);
                    double d2 = __builtin_sqrt(d1 * d1 - aloDFp * ahiDFp);
                    if (ahi < alo) d2 = -d2;
                    alpha = ahi - (ahi - alo) * (ahiDFp + d2 - d1) / (ahiDFp - aloDFp + 2.0 * d2);
                    const double lo = __builtin_fmin(alo, ahi), hi = __builtin_fmax(alo, ahi),
                                 w = __builtin_fabs(alo - ahi);
                    if (!finite_f64(alpha) || alpha < lo + 0.01 * w || alpha > hi - 0.01 * w)
                        alpha = 0.5 * (alo + ahi);
                    alpha = uniform_f64(alpha);
                }
            }
            if (!ls_fail) stage = ST_LS_EVAL;
        }
        if (!ls_fail) {
            if (stage == ST_LS_EVAL) {
                if (sv.n_eval >= 64 * a.opt.max_iter + 1024) { ret = TSF_ST_EVAL_LIMIT; break; }
#pragma unroll
                for (int s = 0; s < PPL; ++s) xk1[s] = __builtin_fma(alpha, pk[s], xk[s]);
            }
            double f1;
            CT_LAP(0);
            const bool bad = coop_eval_owner<KP, GROWTH, MODE, PPL, NW, XIDX, HARM, SPARSE>(sp, sv, cl, rbR, rbU, rbV, xk1, f1, gk1, res, xol CT_PASS);
            f1 = uniform_f64(f1);
            if (stage == ST_INIT) {         // (direct mode only) the initial point
                if (bad) { ret = TSF_ST_INIT_NONFINITE; fk = f1; break; }
                fk = f1;
#pragma unroll
                for (int s = 0; s < PPL; ++s) { gk[s] = gk1[s]; pk[s] = -gk[s]; gk1[s] = 0.0; xk1[s] = 0.0; }
                stage = ST_START_ITER;
                continue;
            }
            if (bad) {
                if (!zoom) {
                    if (lsRestarts >= maxLSRestarts) ls_fail = true;
                    else { alpha = uniform_f64(0.5 * (alpha0 + alpha)); lsRestarts++; }
                } else {
                    alpha = uniform_f64(0.5 * (alpha + __builtin_fmin(alo, ahi)));
                    if (__builtin_fabs(__builtin_fmin(alo, ahi) - alpha) < min_range) ls_fail = true;
                }
                if (!ls_fail) continue;            // re-evaluate at the shortened step
            }
            if (!ls_fail) {
                CT_LAP(0);
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
                        alpha = uniform_f64(alpha * 10.0);
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
                CT_LAP(6);
                if (!ls_ok) { stage = ST_LS_PRE; continue; }
                fk1 = f1;
                // ---- accepted step: k becomes the most recent iterate ----
                { const double tf = fk; fk = fk1; fk1 = tf; }
                double sk[PPL], yk[PPL];
#pragma unroll
                for (int s = 0; s < PPL; ++s) {
                    const double tx = xk[s]; xk[s] = xk1[s]; xk1[s] = tx;
                    const double tg = gk[s]; gk[s] = gk1[s]; gk1[s] = tg;
                    const double tp = pk[s]; pk[s] = pk1[s]; pk1[s] = tp;
                    sk[s] = xk[s] - xk1[s];
                    yk[s] = gk[s] - gk1[s];
                }
                const double dots = bfly_sum4_lanes(pdot_part<PPL>(gk, gk), pdot_part<PPL>(sk, sk),
                                                    pdot_part<PPL>(yk, sk), pdot_part<PPL>(yk, yk));
                const double nrm = __builtin_sqrt(dots);
                const double gradNorm = readlane_f64(nrm, 0), stepNorm = readlane_f64(nrm, 2);
                double qnum = dpp_mov<0x07>(dots);          // quad_perm [3,1,0,0]: y.y, y.s, -, -
                if ((lane & 3) >= 2) qnum = 1.0;
                const double qden = dpp_mov<0x5D>(dots);    // quad_perm [1,3,1,1]: y.s, y.y, y.s, y.s
                const double qv = qnum / qden;
                if (resetB) {
                    const double B0fact = readlane_f64(qv, 0);
                    hist_len = 0; hist_head = 0;
#pragma unroll
                    for (int s = 0; s < PPL; ++s) pk1[s] = pk1[s] / B0fact;
                    alpha = uniform_f64(alpha * B0fact);
                    pk1_scaled = true;
                } else {
                    pk1_scaled = false;
                }
                gammak = readlane_f64(qv, 1);
                const double rho_new = readlane_f64(qv, 2);
                {
                    int hs;
                    if (hist_len < H) { hs = hist_head + hist_len; if (hs >= H) hs -= H; hist_len++; }
                    else { hs = hist_head; hist_head = hist_head + 1; if (hist_head >= H) hist_head -= H; }
                    if (lane == 0) lds.rho[hs] = rho_new;
#pragma unroll
                    for (int s = 0; s < PPL; ++s) {
                        lds.SY[((2 * hs) * PPL + s) * W + lane] = sk[s];
                        lds.SY[((2 * hs + 1) * PPL + s) * W + lane] = yk[s];
                    }
                }
                TSF_WAVE_SYNC();
#pragma unroll
                for (int s = 0; s < PPL; ++s) pk[s] = -gk[s];
                // (the pair of the NEXT step is requested before the dot product of this one -- its LDS round trip was on the
                // chain ten times per iteration --, and the ring index wraps by comparison: `% H` with a run-time H is an
                // integer division, twenty per iteration)
                auto hist_load = [&](int hs_, double (&s_)[PPL], double (&y_)[PPL], double &r_) {
#pragma unroll
                    for (int s = 0; s < PPL; ++s) {
                        s_[s] = lds.SY[((2 * hs_) * PPL + s) * W + lane];
                        y_[s] = lds.SY[((2 * hs_ + 1) * PPL + s) * W + lane];
                    }
                    r_ = lds.rho[hs_];
                };
                if (hist_len > 0) {
                    double si[PPL], yi[PPL], rh;
                    int hs = hist_head + hist_len - 1;
                    if (hs >= H) hs -= H;
                    hist_load(hs, si, yi, rh);
                    for (int h = hist_len - 1; h >= 0; --h) {
                        double sn[PPL], yn[PPL], rhn = 0.0;
                        int hsn = hs - 1;
                        if (hsn < 0) hsn += H;
                        if (h > 0) hist_load(hsn, sn, yn, rhn);
                        const double aa = lane63(rh * pdot_l63<PPL>(si, pk));
#pragma unroll
                        for (int s = 0; s < PPL; ++s) pk[s] = __builtin_fma(-aa, yi[s], pk[s]);
                        if (lane == 0) lds.alphas[h] = aa;
#pragma unroll
                        for (int s = 0; s < PPL; ++s) { si[s] = sn[s]; yi[s] = yn[s]; }
                        rh = rhn; hs = hsn;
                    }
                }
                TSF_WAVE_SYNC();
#pragma unroll
                for (int s = 0; s < PPL; ++s) pk[s] = pk[s] * gammak;
                if (hist_len > 0) {
                    double si[PPL], yi[PPL], rh;
                    int hs = hist_head;
                    hist_load(hs, si, yi, rh);
                    double al = lds.alphas[0];
                    for (int h = 0; h < hist_len; ++h) {
                        double sn[PPL], yn[PPL], rhn = 0.0, aln = 0.0;
                        int hsn = hs + 1;
                        if (hsn >= H) hsn -= H;
                        if (h + 1 < hist_len) { hist_load(hsn, sn, yn, rhn); aln = lds.alphas[h + 1]; }
                        const double cc = lane63(al - rh * pdot_l63<PPL>(yi, pk));
#pragma unroll
                        for (int s = 0; s < PPL; ++s) pk[s] = __builtin_fma(cc, si[s], pk[s]);
#pragma unroll
                        for (int s = 0; s < PPL; ++s) { si[s] = sn[s]; yi[s] = yn[s]; }
                        rh = rhn; al = aln; hs = hsn;
                    }
                }
                TSF_WAVE_SYNC();
                const double dF = __builtin_fabs(fk1 - fk);
                const double fmaxv = __builtin_fmax(__builtin_fabs(fk1),
                                                    __builtin_fmax(__builtin_fabs(fk), 1.0));
                gp = pdot<PPL>(gk, pk);
                gp_valid = true;
                if (dF < a.opt.tol_obj) ret = TSF_ST_ABSF;
                else if (dF < a.opt.tol_rel_obj_eps * fmaxv) ret = TSF_ST_RELF;
                else if (gradNorm < a.opt.tol_grad) ret = TSF_ST_ABSGRAD;
                else if (-gp / __builtin_fmax(__builtin_fabs(fk), 1.0) < a.opt.tol_rel_grad_eps) ret = TSF_ST_RELGRAD;
                else if (stepNorm < a.opt.tol_param) ret = TSF_ST_ABSX;
                else if (itNum >= a.opt.max_iter) ret = TSF_ST_MAXIT;
                else ret = 0;
                CT_LAP(5);
                if (ret != 0) break;
                stage = ST_START_ITER;
                continue;
            }
        }
        // line search failed
        if (resetB) { ret = TSF_ST_LSFAIL; break; }
        resetB = 2;
        stage = ST_START_LS;
    }
    if (lane == 0) cl.cmd = COOP_EXIT;
    lds_barrier();                                          // A of the helpers' last round
    store_theta<PPL>(a, sv, n, xk, a.theta);
    if (lane == 0) { a.status[n] = ret; a.n_iter[n] = itNum; a.n_eval[n] = sv.n_eval; a.fval[n] = fk; }
    CT_LAP(0);
    CT_FLUSH(a.grad_out, n);
}

// direct mode: a series whose setup status says "no fit" (as fit_kernel reports it)
template <int KP, int PPL>
__device__ __forceinline__ void coop_report_unfitted(const FitArgs &a, const SeriesView &sv, int64_t n)
{
    const int lane = lane_id();
    const SeriesTab st = a.stab[n];
    if (lane == 0) {
        a.y_scale[n] = st.y_scale;
        if (!a.aligned || n == 0) a.grid_out[a.aligned ? 0 : n] = a.gtab[grid_index(a, n)].info;
    }
    double xk[PPL];
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const int p = lane + s * W;
        xk[s] = (p == 0) ? st.k0 : (p == 1 ? st.m0 : 0.0);
        if (st.status0 == TSF_ST_CONSTANT && p == 2) xk[s] = -20.72326583694641;
    }
    store_theta<PPL>(a, sv, n, xk, a.theta);
    if (lane == 0) { a.status[n] = st.status0; a.n_iter[n] = 0; a.n_eval[n] = 0; a.fval[n] = 0.0; }
}

// ---- the kernel: persistent workgroups over the checkpoint list ---------------------------------
// SPARSE: KP = SP_DENSE registers on the tables and the LDS of the 64-column model (KL), series of <= 12 steps only
template <int KP, int GROWTH, int MODE, int PPL, int NW, bool XIDX, int HARM = 0, bool SPARSE = false>
__global__ __launch_bounds__(NW * 64) void fit_coop_kernel(FitArgs a)
{
    constexpr int KL = SPARSE ? 64 : KP;
    extern __shared__ __align__(16) unsigned char smem[];
    CoopLds<KL, PPL> &cl = *reinterpret_cast<CoopLds<KL, PPL> *>(smem);
    double *rbR = reinterpret_cast<double *>(smem + sizeof(CoopLds<KL, PPL>));
    const int rb_rows = SPARSE ? 12 : coop_rb_rows(a.NTmax);      // (SPARSE: <= 12 steps by launch; 16 would not fit the LDS beside the 64-column tables)
    double *rbU = rbR + (size_t)rb_rows * W, *rbV = rbU + (size_t)rb_rows * W;
    if (a.run_flag && (*a.run_flag != 0) != (a.run_if != 0)) return;          // launch guard (FitArgs::run_flag), as in fit_kernel
    const int wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);     // wave-uniform: addresses built from it stay scalar
    // direct mode: every series of the call, fitted here from its initial values; otherwise the fits
    // fit_kernel suspended
    const bool direct = a.coop_after == COOP_DIRECT;
    int n_ckpt = direct ? (int)a.N : a.coop_ctl[1];
    if (!direct && n_ckpt > a.coop_max) n_ckpt = a.coop_max;
    for (;;) {
        // (queue fetch kept branch-free: see the compiler note in DESIGN.md section 5)
        if (wid == 0) {
            const int it = atomicAdd(&a.coop_ctl[2], lane_id() == 0 ? 1 : 0);
            if (lane_id() == 0) cl.item = it;
        }
        __syncthreads();
        const int item = cl.item;
        __syncthreads();
        if (item >= n_ckpt) break;
        const int64_t n = direct ? (a.order ? (int64_t)a.order[item] : (int64_t)item) : (int64_t)a.coop_list[item];      // (cost hints: tsf_set_cost_hints)
        SeriesView sv;
        make_view<KL, PPL>(a, n, sv);
        if (direct && a.stab[n].status0 != 0) {
            // never reaches the optimiser (fbprophet raises: too few rows / cap <= floor; or skips the fit:
            // constant y): the owner reports it, nobody touches the series' tables (they may not exist)
            if (wid == 0) coop_report_unfitted<KL, PPL>(a, sv, n);
            continue;
        }
        if constexpr (SPARSE) {
            // the lanes' entry words and the columns' fold programs of this series' grid (as fit_kernel<..., SPARSE> stages them)
            const int64_t g = grid_index(a, n);
            for (int i = (int)threadIdx.x; i < SP_M * W; i += NW * W) cl.sp_list[i] = (unsigned short)a.sp_meta[(size_t)g * SP_M * W + i];
            if (threadIdx.x < W) cl.sp_list[SP_M * W + threadIdx.x] = (unsigned short)SP_END;
            if (threadIdx.x < SP_MAXC) cl.sp_prog[threadIdx.x] = a.sp_prog[(size_t)g * SP_MAXC + threadIdx.x];
            __syncthreads();
        }
        // rows [NT, COOP_NTB) of the row buffers: zeros (the straight-line chains of coop_helper_pf)
        if (sv.NT < COOP_NTB) {
            const int zr = rb_rows < COOP_NTB ? rb_rows : COOP_NTB;
            for (int i = sv.NT * W + (int)threadIdx.x; i < zr * W; i += NW * W) { rbR[i] = 0.0; rbU[i] = 0.0; rbV[i] = 0.0; }
        }
#ifdef TSF_COOP_TIMING
        if (threadIdx.x == 0) cl.dbg = a.grad_out ? (long long *)a.grad_out + (size_t)n * 64 : nullptr;
#endif
        if (wid == 0) {
            for (int i = lane_id(); i < TSF_MAX_P + W; i += W) cl.w.th[i] = 0.0;
            TSF_WAVE_SYNC();
            coop_owner<KP, GROWTH, MODE, PPL, NW, XIDX, HARM, SPARSE>(a, sv, n, direct ? nullptr : a.coop_slots + (size_t)item * a.coop_stride,
                                                                      cl, rbR, rbU, rbV,
                                                                      coop_xl_bytes(KP, a.NTmax) > 0 ? rbV + (size_t)(SPARSE ? 12 : COOP_NTB) * W : nullptr);
        } else if constexpr (SPARSE) {
            coop_helper_ntb<KP, GROWTH, MODE, PPL, NW, XIDX, 12, HARM, true>(a.sp, sv, cl, rbR, rbU, rbV, wid);    // (launched for NTmax <= 12 only)
        } else if (sv.NT > COOP_NTB) {
            coop_helper<KP, GROWTH, MODE, PPL, NW, XIDX>(a.sp, sv, cl, rbR, rbU, rbV, wid);
        } else if (a.NTmax > 12) {      // (the call's longest series decides: the LDS of the 12-step variant is sized by it)
            coop_helper_ntb<KP, GROWTH, MODE, PPL, NW, XIDX, COOP_NTB, 0>(a.sp, sv, cl, rbR, rbU, rbV, wid);
        } else {
            coop_helper_ntb<KP, GROWTH, MODE, PPL, NW, XIDX, 12, HARM>(a.sp, sv, cl, rbR, rbU, rbV, wid);
        }
    }
}

}  // namespace tsf
