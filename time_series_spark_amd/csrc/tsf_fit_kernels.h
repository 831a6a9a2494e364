// tsf_fit_kernels.h -- the fit / eval kernel templates (see tsf_common.h for the execution
// model and the canonical-arithmetic contract).  Instantiated by tsf_inst_g*m*.hip.
#pragma once
#include <cstddef>
#include "tsf_common.h"

namespace tsf {

// ---------------------------------------------------------------------------------------
// -log posterior and gradient (prophet.stan model block), one wave per series
// ---------------------------------------------------------------------------------------

struct SeriesView {
    int T, NT, S, P, cnt;               // cnt: valid rows of this lane's chunk
    unsigned long long sp_prog;         // (SPARSE kernels) lane c: the fold program of sparse column c
    double *sp_acc;                     // (SPARSE kernels) [SP_MAXC][SP_E] slots in LDS
    const unsigned short *sp_list;      // (SPARSE kernels) [SP_M + 1][64] entry words of the lanes, last row first, in LDS
    int S_out;                          // changepoints in the caller's layout (S = 1 > S_out = 0: dummy changepoint)
    const double *tw, *yw, *Xw;         // step-major tables
    // (quadratic-form kernels, round 6) the caller's own y rows instead of a scaled step-major copy: row i of the series at
    // y_raw[y_base + i], scaled in the register as setup_series_kernel would have -- (y - 0) / y_scale, linear growth
    const void *y_raw;
    int y_dtype;
    long long y_base;
    double y_scl;
    const double *Bw;                   // (HARM kernels) base pairs [NT][seasonality][64][2]: sin theta, cos theta of the row
    int n_xd;                           // (HARM kernels) dense explicit columns behind the Fourier columns (read from Xw)
    const int32_t *uw;                  // (lattice panels) row -> row of the shared table Xu, step-major
    const double *Xu;                   // (lattice panels) [U][KP] design rows of the timestamp lattice
    const double *Bu;                   // (lattice panels, HARM kernels) [U][seasonality][2] base pairs of the lattice points
    const uint16_t *cw;
    const int32_t *Lj;
    const double *t_change;
    double cap, tau;
    int n_eval;
    // Lane-resident copies of the per-series tables an evaluation needs (set_lane_tables): read from
    // global memory once per series -- inside the evaluation each of them was a dependent load at the
    // head of a serial chain (measured on a lone wave: 18 k of the 46 k cycles of an evaluation were
    // the set-up and the tail, section 5 of DESIGN.md).  p = lane + 64 s.
    double tc_l;                        // t_change[lane]            (0 for lane >= S)
    int Lj_l, Ljm1_l;                   // Lj[lane] (lane < S), Lj[lane - 1] (1 <= lane <= S); else 0
    double prior_l[2];                  // prior scale of design column p - 3 - S (1 where p is no beta)
    int Ljp_l[2];                       // Lj[p - 3], t_change[p - 3] where p is a delta (else 0)
    double tcp_l[2];
};

template <int PPL>
__device__ __forceinline__ void set_lane_tables(const DevSpec *sp, SeriesView &sv)
{
    const int lane = (int)threadIdx.x & (W - 1);
    const int S = sv.S;
    sv.tc_l = (lane < S) ? sv.t_change[lane] : 0.0;
    sv.Lj_l = (lane < S) ? sv.Lj[lane] : 0;
    sv.Ljm1_l = (lane >= 1 && lane <= S) ? sv.Lj[lane - 1] : 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        sv.prior_l[s] = 1.0; sv.Ljp_l[s] = 0; sv.tcp_l[s] = 0.0;
        if (s < PPL) {
            const int p = lane + s * W;
            if (p >= 3 + S && p < sv.P) sv.prior_l[s] = sp->prior[p - 3 - S];
            if (p >= 3 && p < 3 + S) { sv.Ljp_l[s] = sv.Lj[p - 3]; sv.tcp_l[s] = sv.t_change[p - 3]; }
        }
    }
}

// LDS carve-up for one wave
template <int KP, int PPL>
struct WaveLds {
    double th[TSF_MAX_P + W];          // theta as stored by the last evaluation; zero beyond P
    double ks[NTAB + 1], mc[NTAB + 1];
    double tp1[NTAB], tp2[NTAB];
    double tot1[W + 1], tot2[W + 1];
    double d1[NTAB + 1], d2[NTAB + 1], rb[NTAB + 1], ab[NTAB + 1];
    double rho[MAXH], alphas[MAXH];
    double accR[KP];                   // per-column sums X^T r
    // L-BFGS history, LAST and interleaved -- pair h: s at [(2h) PPL + slot][64], y at [(2h + 1) PPL + slot][64] -- so that a
    // launch allocates only the `history` pairs it uses (wave_lds_bytes): with all MAXH = 8 pairs a two-slot block took
    // 23.7 KB, six blocks per CU; with Stan's five pairs 17.7 KB, eight
    double SY[2 * MAXH * PPL * W];
};
template <int KP, int PPL>
__host__ __device__ inline size_t wave_lds_bytes(int history)
{
    const int H = history > MAXH ? MAXH : (history < 1 ? 1 : history);
    using WL = WaveLds<KP, PPL>;
    return offsetof(WL, SY) + sizeof(double) * 2 * (size_t)H * PPL * W;
}

// LDS hand-off inside one wave (see wave_sync in tsf_common.h); never a workgroup barrier, so
// the same code runs in one-wave and in multi-wave workgroups
#define TSF_WAVE_SYNC() wave_sync()

__device__ __forceinline__ int lane_id() { return (int)threadIdx.x & (W - 1); }

static_assert(3 + TSF_MAX_S <= W, "the changepoint parameters delta_j (p = 3 + j) live in slot 0 of every lane layout: "
                                  "readlane_f64(th[0], 3 + j) in the recurrences");
// theta[p] for a wave-uniform p, straight from the owning lane's register
template <int PPL>
__device__ __forceinline__ double theta_at(const double (&th)[PPL], int p)
{
    if (PPL == 1) return readlane_f64(th[0], p);
    return (p < W) ? readlane_f64(th[0], p) : readlane_f64(th[PPL - 1], p - W);
}

#ifdef TSF_FIT_TIMING      // dev only: cycles per phase of fit_kernel (s_memtime), summed per series
#define FT_DECL long long ft_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long ft_t0 = __builtin_readcyclecounter(), ft_start = ft_t0
#define FT_LAP(k) do { const long long t_ = __builtin_readcyclecounter(); ft_acc[k] += t_ - ft_t0; ft_t0 = t_; } while (0)
#define FT_FLUSH(dst, n) do { if ((dst) && lane_id() == 0) { ft_acc[7] = __builtin_readcyclecounter() - ft_start; for (int k_ = 0; k_ < 8; ++k_) ((long long *)(dst))[(size_t)(n) * 8 + k_] = ft_acc[k_]; } } while (0)
#define FT_ARGS , long long (&ft_acc)[8], long long &ft_t0
#define FT_PASS , ft_acc, ft_t0
#else
#define FT_DECL do { } while (0)
#define FT_LAP(k) do { } while (0)
#define FT_FLUSH(dst, n) do { } while (0)
#define FT_ARGS
#define FT_PASS
#endif

// Column sums ACC_j = sum over the 64 chunk partials acc[j], as a butterfly with offsets
// 32, 16, 1, 2, 4, 8, done for all KP columns at once: the 32- and 16-stages transpose pairs of
// registers with v_permlane32_swap / v_permlane16_swap (halving the register count each time),
// the last four stages are DPP butterflies inside the 16-lane rows.  Results go to lds.accR.
template <int KP, int PPL, class L>
__device__ __forceinline__ void column_sums(double (&acc)[KP], L &lds)
{
    static_assert(KP % 4 == 0, "KP must be a multiple of 4");
    const int lane = lane_id();
    double c[KP / 2];
#pragma unroll
    for (int i = 0; i < KP / 2; ++i) {
        double a = acc[2 * i], b = acc[2 * i + 1];
        swap32(a, b);                  // a = [col 2i lo | col 2i+1 lo], b = [col 2i hi | col 2i+1 hi]
        c[i] = a + b;                  // lanes 0-31: column 2i, lanes 32-63: column 2i+1
    }
    double d[KP / 4];
#pragma unroll
    for (int i = 0; i < KP / 4; ++i) {
        double a = c[2 * i], b = c[2 * i + 1];
        swap16(a, b);                  // a = [A.r0, B.r0, A.r2, B.r2], b = [A.r1, B.r1, A.r3, B.r3]
        d[i] = row_bfly_sum(a + b);    // row 0: col 4i, row 1: col 4i+2, row 2: col 4i+1, row 3: col 4i+3
    }
    if ((lane & 15) == 0) {
        const int r = lane >> 4;
        const int sub = (r == 0) ? 0 : (r == 1 ? 2 : (r == 2 ? 1 : 3));
#pragma unroll
        for (int i = 0; i < KP / 4; ++i) lds.accR[4 * i + sub] = d[i];
    }
}

// G-column slice of the register transpose network of column_sums (above)
template <int G>
__device__ __forceinline__ void column_sums_g(double (&acc)[G], double *accR)
{
    static_assert(G % 4 == 0, "column groups are multiples of 4");
    const int lane = lane_id();
    double c[G / 2];
#pragma unroll
    for (int i = 0; i < G / 2; ++i) {
        double a = acc[2 * i], b = acc[2 * i + 1];
        swap32(a, b);
        c[i] = a + b;
    }
    double d[G / 4];
#pragma unroll
    for (int i = 0; i < G / 4; ++i) {
        double a = c[2 * i], b = c[2 * i + 1];
        swap16(a, b);
        d[i] = row_bfly_sum(a + b);
    }
    if ((lane & 15) == 0) {
        const int r = lane >> 4;
        const int sub = (r == 0) ? 0 : (r == 1 ? 2 : (r == 2 ? 1 : 3));
#pragma unroll
        for (int i = 0; i < G / 4; ++i) accR[4 * i + sub] = d[i];
    }
}

// Ascending / descending loop over a run-time range, body written out four times per trip: the
// recurrences over the changepoints read lanes (v_readlane is convergent, so the compiler does not
// unroll such loops by itself); unrolled, the lane reads of the next steps issue ahead of the chain
// and three of four taken branches disappear.
#define TSF_UNROLL4_UP(j, lo, hi, BODY) do { int j = (lo); for (; j + 4 <= (hi); ) { BODY; ++j; BODY; ++j; BODY; ++j; BODY; ++j; } for (; j < (hi); ++j) { BODY; } } while (0)
#define TSF_UNROLL4_DOWN(c, hi, lo, BODY) do { int c = (hi); for (; c - 3 >= (lo); ) { BODY; --c; BODY; --c; BODY; --c; BODY; --c; } for (; c >= (lo); --c) { BODY; } } while (0)

// Logistic growth, slope and offset of the trend segments as SCANS over the parameter lanes (round 5; oracle cn_eval):
// lane p = 3 + j holds delta_j (th_l) and t_change[j] (tcp_l).  ks[j+1] = k + (inclusive prefix sum of delta)[p]
// (prophet.stan: k + cumulative_sum(delta)); the offset recurrence m[j+1] = m[j] + (t_j - m[j]) (1 - ks[j] / ks[j+1]) is
// the affine map m -> rho_j m + t_j (1 - rho_j), rho_j = ks[j] / ks[j+1] (ONE lane-parallel division), and
// m[j+1] = fma(A, m, B) with (A, B) the prefix composition of those maps.  Out: lane 3 + j holds ks[j+1], mc[j+1].
// ~110 vector instructions and 10 dependent stages instead of two chains of S steps (~12 instructions per step).
__device__ __forceinline__ void logistic_tables_lanes(double k, double m, double th_l, double tcp_l, int S,
                                                      double &ksn_out, double &mcn_out)
{
    const int lane = (int)threadIdx.x & (W - 1);
    const bool isd = lane >= 3 && lane < 3 + S;
    const double pd = prefix_scan(isd ? th_l : 0.0);
    const double ksn = k + pd;
    double ksp = dpp_mov<DPP_WAVE_SHR1>(ksn);              // ks[j] for j >= 1
    if (lane == 3) ksp = k;
    const double rho = isd ? ksp / ksn : 1.0;
    double fa = rho, fb = isd ? tcp_l * (1.0 - rho) : 0.0;
    affine_prefix_scan(fa, fb);
    ksn_out = ksn;
    mcn_out = __builtin_fma(fa, m, fb);
}

// Segment tables ks[c], mc[c] (slope and offset of trend segment c): sequential recurrences over
// the changepoints.  Every lane runs the same S steps and lane c stops updating after its first c
// terms, so lane c ends with exactly the sequentially rounded ks[c], mc[c] (no per-step LDS write /
// table load).  Writes lds.ks / lds.mc; the caller synchronises.
template <int GROWTH, int PPL, class L>
__device__ __forceinline__ void segment_tables(const SeriesView &sv, L &lds, const double (&th)[PPL])
{
    const int lane = lane_id();
    const int S = sv.S;
    const double k = theta_at<PPL>(th, 0), m = theta_at<PPL>(th, 1);
    double ksv = k, mcv = m;
    const double tcl = sv.tc_l;
    if (GROWTH == 0) {
        TSF_UNROLL4_UP(j, 0, S, {
            const double dj = readlane_f64(th[0], 3 + j);
            const double ksn = ksv + dj;
            const double mcn = mcv + ((-readlane_f64(tcl, j)) * dj);
            if (j < lane) { ksv = ksn; mcv = mcn; }
        });
        if (lane <= S) { lds.ks[lane] = ksv; lds.mc[lane] = mcv; }
    } else {
        // logistic (round 5): the two S-step chains as scans, see logistic_tables_lanes
        double ksn, mcn;
        logistic_tables_lanes(k, m, th[0], sv.tcp_l[0], S, ksn, mcn);
        if (lane >= 3 && lane < 3 + S) { lds.ks[lane - 2] = ksn; lds.mc[lane - 2] = mcn; }
        if (lane == 0) { lds.ks[0] = k; lds.mc[0] = m; }
    }
}

// Logistic growth, the trend part of the gradient as scans (round 5; oracle cn_eval): per-segment sums from the suffix
// sums (lane c <= S), the reverse sweep through the offset recurrence abar[c] = D2[c] + rho_c abar[c+1] (abar[S] = D2[S])
// as the suffix composition of the maps x -> rho_c x + D2[c] (lane S: the constant map), rho_bar[c] = abar[c+1]
// (t_c - mc[c]), the two adjustments of D1 lane-parallel, and the slope gradient's running sums as one suffix scan.
// Out (without the factor -1/sigma^2): gk_raw = sum of all adjusted D1, gm_raw = abar[0], and in lane 3 + j the running
// sum that belongs to delta_j.  Reads lds.tot1 / tot2 / tp1 / tp2 / ks / mc; no scratch.
// The part of the reverse sweep that depends on the segment tables alone (ks, mc): the three quotients and the lane's
// segment values.  A kernel whose trend wave waits for the rows (fit_coop_kernel) forms them while it waits; same
// operations on the same operands as when they were formed inside the sweep.
struct LogisticReversePre { double ks_c, mc_c, fa, q1, q2, tdm; };
template <class L>
__device__ __forceinline__ void logistic_reverse_pre(const SeriesView &sv, const L &lds, LogisticReversePre &o)
{
    const int lane = (int)threadIdx.x & (W - 1);
    const int S = sv.S, c = lane;
    const bool seg = c <= S;
    const int cc = seg ? c : S;                             // (lanes past S read in-range entries and are masked)
    o.ks_c = lds.ks[cc]; o.mc_c = lds.mc[cc];
    const double ks_n = lds.ks[cc < S ? cc + 1 : S], ks_p = lds.ks[cc > 0 ? cc - 1 : 0];
    o.fa = (c < S) ? o.ks_c / ks_n : (c == S ? 0.0 : 1.0);
    o.q1 = -1.0 / ks_n;
    o.q2 = (ks_p / o.ks_c) / o.ks_c;
    o.tdm = sv.tc_l - o.mc_c;
}
template <class L>
__device__ __forceinline__ void logistic_reverse_post(const SeriesView &sv, L &lds, const LogisticReversePre &pr, double TA, double TB,
                                                      double &gk_raw, double &gm_raw, double &gd_l)
{
    const int lane = (int)threadIdx.x & (W - 1);
    const int S = sv.S, c = lane;
    const bool seg = c <= S;
    double D1 = 0.0, D2 = 0.0;
    if (seg) {
        const int Ljm = (c > 0) ? sv.Ljm1_l : 0, Ljc = (c < S) ? sv.Lj_l : 0;
        const double hiA = (c == 0) ? TA : lds.tp1[c - 1] + lds.tot1[Ljm + 1];
        const double hiB = (c == 0) ? TB : lds.tp2[c - 1] + lds.tot2[Ljm + 1];
        const double loA = (c == S) ? 0.0 : lds.tp1[c] + lds.tot1[Ljc + 1];
        const double loB = (c == S) ? 0.0 : lds.tp2[c] + lds.tot2[Ljc + 1];
        const double A = hiA - loA, B = hiB - loB;
        D1 = A - pr.mc_c * B;
        D2 = -(pr.ks_c * B);
    }
    double fa = pr.fa, fb = seg ? D2 : 0.0;
    affine_suffix_scan(fa, fb);                             // fb: abar[c]
    const double ab_next = dpp_mov<DPP_WAVE_SHL1>(fb);      // abar[c + 1]
    const double rb = (c < S) ? ab_next * pr.tdm : 0.0;
    const double rb_prev = dpp_mov<DPP_WAVE_SHR1>(rb);      // rb[c - 1]
    double d = D1;
    if (c < S) d = d + rb * pr.q1;
    if (c >= 1 && seg) d = d + rb_prev * pr.q2;
    const double ss = suffix_scan(seg ? d : 0.0);
    gk_raw = readlane_f64(ss, 0);
    gm_raw = readlane_f64(fb, 0);
    gd_l = dpp_mov<DPP_WAVE_SHR1>(dpp_mov<DPP_WAVE_SHR1>(ss));      // lane 3 + j: ss[j + 1]
}
template <class L>
__device__ __forceinline__ void logistic_reverse_lanes(const SeriesView &sv, L &lds, double TA, double TB,
                                                       double &gk_raw, double &gm_raw, double &gd_l)
{
    LogisticReversePre pr;
    logistic_reverse_pre(sv, lds, pr);
    logistic_reverse_post(sv, lds, pr, TA, TB, gk_raw, gm_raw, gd_l);
}

// f and the gradient from the time-axis sums of one evaluation: lds.tot1 / tot2 (suffix sums of the
// per-chunk trend sums, [W] = 0), lds.tp1 / tp2 (chunk-local partial sums at the changepoint rows),
// lds.accR (per-column sums X^T r), lds.ks / mc, and sse_t.  scr: d1, d2, rb, ab scratch (logistic
// growth).  The caller has synchronised the wave after writing those.
template <int GROWTH, int PPL, class L, class D>
__device__ __forceinline__ bool eval_tail(const DevSpec *__restrict__ sp, const SeriesView &sv, L &lds,
                                          D &scr, const double (&th)[PPL], double sigma,
                                          double inv_s2, double sse_t, double &f_out,
                                          double (&g)[PPL])
{
    const int lane = lane_id();
    const int S = sv.S, T = sv.T;
    const double k = theta_at<PPL>(th, 0), m = theta_at<PPL>(th, 1), ls = theta_at<PPL>(th, 2);
    const double TA = lds.tot1[0], TB = lds.tot2[0];

    // prior terms
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
    f = f + (0.5 * sse_t) * inv_s2;

    const double nis = -inv_s2;
    double gk = 0.0, gm = 0.0;
#pragma unroll
    for (int s = 0; s < PPL; ++s) g[s] = 0.0;
    if (GROWTH == 1) {
        double gd_l;
        logistic_reverse_lanes(sv, lds, TA, TB, gk, gm, gd_l);
        gk = nis * gk; gm = nis * gm;
        if (lane >= 3 && lane < 3 + S) g[0] = nis * gd_l;
    } else {
        gk = nis * TA;
        gm = nis * TB;
    }
    bool bad = !finite_f64(f);
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const int p = lane + s * W;
        double gv = 0.0;
        if (p == 0) gv = gk + k / 25.0;
        else if (p == 1) gv = gm + m / 25.0;
        else if (p == 2) gv = ((double)T - sse_t * inv_s2) + 4.0 * s2;
        else if (p < 3 + S) {
            const int j = p - 3;
            double gd;
            if (GROWTH == 0) {
                const int Lj = sv.Ljp_l[s];
                const double SA = lds.tp1[j] + lds.tot1[Lj + 1];
                const double SB = lds.tp2[j] + lds.tot2[Lj + 1];
                gd = nis * (SA - sv.tcp_l[s] * SB);
            } else {
                gd = g[s];
            }
            const double dj = th[s];
            const double sgn = (double)((dj > 0.0) - (dj < 0.0));
            gv = gd + sgn / sv.tau;
        } else if (p < sv.P) {
            const int j = p - 3 - S;
            const double pr = sv.prior_l[s];
            gv = nis * lds.accR[j] + th[s] / (pr * pr);
        }
        g[s] = gv;
        bad = bad || !finite_f64(gv);
    }
    f_out = f;
    TSF_WAVE_SYNC();
    return __any(bad);
}

// The Fourier columns of one seasonality from its base pair, in column order (sin h, cos h; h = 1 .. O): the canonical
// recurrence of fourier_harmonics (tsf_common.h) written out at compile time -- emit(column, value) sees a constant column.
template <int O, int COL0, class F>
__device__ __forceinline__ void harm_columns(double s1, double c1, F &&emit)
{
    if constexpr (O > 0) {
        const double c2 = 2.0 * c1;
        double sp = 0.0, cp = 1.0, sc = s1, cc = c1;
#pragma unroll
        for (int h = 1; h <= O; ++h) {
            if (h > 1) {
                const double sn = __builtin_fma(c2, sc, -sp), cn = __builtin_fma(c2, cc, -cp);
                sp = sc; cp = cc; sc = sn; cc = cn;
            }
            emit(COL0 + 2 * (h - 1), sc);
            emit(COL0 + 2 * (h - 1) + 1, cc);
        }
    }
}
// all Fourier columns of a row: bp[s] = (sin theta_s, cos theta_s)
template <int HARM, class F>
__device__ __forceinline__ void harm_row(const double2 *bp, F &&emit)
{
    constexpr int O0 = harm_order(HARM, 0), O1 = harm_order(HARM, 1), O2 = harm_order(HARM, 2);
    harm_columns<O0, 0>(bp[0].x, bp[0].y, emit);
    if constexpr (O1 > 0) harm_columns<O1, 2 * O0>(bp[1].x, bp[1].y, emit);
    if constexpr (O2 > 0) harm_columns<O2, 2 * (O0 + O1)>(bp[2].x, bp[2].y, emit);
}

// MODE: 0 all columns additive, 1 all multiplicative, 2 mixed (Ka additive first)
// HARM != 0 (round 5): the Fourier columns are not read but EXPANDED from the row's base pairs (FitArgs::Bw: two
// doubles per seasonality and row instead of 2 x order) by the recurrence that defines the canonical design values --
// the same bits as the table Xw holds --, once for X.beta and once more for the per-column sums: no design row is held,
// the kernel runs at three waves per SIMD, and a series' per-evaluation traffic is its base pairs, t, y and the segment
// word (38 KB for 730 rows of the yearly + weekly model instead of 185 KB).  Dense explicit columns behind the Fourier
// columns (sv.n_xd of them) still come from Xw.
// L: the wave's LDS carve-up (WaveLds, or NewtonLds: any struct with th, ks, mc, tp1, tp2, tot1,
// tot2, d1, d2, rb, ab, accR)
// GNTR > 0 (wide models, KP > 32, series of <= 64 GNTR rows): the per-column sums in groups of 8 columns AFTER
// the row pass, with the weights of the rows kept in GNTR registers per lane -- 8 accumulators live instead of KP.
// With all 64 accumulators live the kernel needs 379 registers (one wave per SIMD, design values through AGPRs);
// grouped it runs at two.  Column by column the fma chain (rows q descending) and the reduction network are those
// of the ungrouped form: same bits.
template <int KP, int GROWTH, int MODE, int PPL, bool XIDX = false, class L = WaveLds<KP, PPL>, int GNTR = 0, bool SPARSE = false, int HARM = 0, bool PF = false>
__device__ __forceinline__ bool eval_fg(const DevSpec *__restrict__ sp, SeriesView &sv,
                                        L &lds, const double (&th)[PPL],
                                        double &f_out, double (&g)[PPL] FT_ARGS)
{
    const int lane = lane_id();
    const int S = sv.S, NT = sv.NT, T = sv.T;
    const int Ka = (MODE == 0) ? KP : (MODE == 1 ? 0 : sp->Ka);
    // design row + coefficients held in registers / SGPRs where they fit
#ifndef TSF_FIT_HOLD_MAX
#define TSF_FIT_HOLD_MAX 32     // (dev: -DTSF_FIT_HOLD_MAX=0 streams every design row twice instead of holding it)
#endif
    constexpr bool HOLD = KP <= TSF_FIT_HOLD_MAX;
    static_assert(!SPARSE || (KP == SP_DENSE && HOLD && !XIDX && GNTR == 0 && MODE != 2), "sparse columns: the 28-column row in registers, one column mode");
    static_assert(HARM == 0 || (HOLD && GNTR == 0 && MODE != 2 && harm_kf(HARM) <= KP), "harmonics in registers: the 28-column row form, one column mode");
    static_assert(!(HARM != 0 && XIDX) || !SPARSE, "base pairs of a timestamp lattice: no sparse columns (and no dense column behind the Fourier block: sv.n_xd == 0, by the launch rule)");
    constexpr int KF = harm_kf(HARM), NS = harm_ns(HARM) > 0 ? harm_ns(HARM) : 1;
    constexpr int NXD = HARM != 0 ? KP - KF : 0;            // dense explicit columns at most
    constexpr int XCOLS = SPARSE ? 64 : KP;     // columns per row of the design table
    sv.n_eval++;
    const double k = theta_at<PPL>(th, 0), m = theta_at<PPL>(th, 1), ls = theta_at<PPL>(th, 2);
    const double sigma = dm_exp_sel(ls);
    const double inv_s2 = 1.0 / (sigma * sigma);
    if (!HOLD || SPARSE) {
#pragma unroll
        for (int s = 0; s < PPL; ++s) lds.th[lane + s * W] = th[s];
    }
    segment_tables<GROWTH, PPL>(sv, lds, th);
    double bs[HOLD ? KP : 1];
    if (HOLD) {
        if (PPL == 2 && 3 + S + KP <= W) {
            // two parameters per lane, but every held coefficient lives in slot 0 (25 changepoints: p = 28 .. 55): one
            // lane read each instead of two and a select
#pragma unroll
            for (int j = 0; j < (HOLD ? KP : 1); ++j) bs[j] = readlane_f64(th[0], 3 + S + j);
        } else {
#pragma unroll
            for (int j = 0; j < (HOLD ? KP : 1); ++j) bs[j] = (3 + S + j < PPL * W) ? theta_at<PPL>(th, 3 + S + j) : 0.0;
        }
    }
    TSF_WAVE_SYNC();
    FT_LAP(1);

    double sse = 0.0, rt1 = 0.0, rt2 = 0.0;
    constexpr bool GROUPED = GNTR > 0 && !HOLD;
    double acc[GROUPED ? 1 : KP];
#pragma unroll
    for (int j = 0; j < (GROUPED ? 1 : KP); ++j) acc[j] = 0.0;
    double wr[GROUPED ? GNTR : 1], wg[(GROUPED && MODE != 0) ? GNTR : 1];     // r and r * trend of the lane's rows
#pragma unroll
    for (int j = 0; j < (GROUPED ? GNTR : 1); ++j) wr[j] = 0.0;
#pragma unroll
    for (int j = 0; j < ((GROUPED && MODE != 0) ? GNTR : 1); ++j) wg[j] = 0.0;
    unsigned sp_cur = SP_END;
    int sp_i = 0;
    if constexpr (SPARSE) sp_cur = sv.sp_list[lane];
    // PF (round 5, HARM): the row's inputs (segment word, t, y, base pairs: 50 bytes) are requested one step AHEAD of their
    // use, in a kernel compiled for two waves per SIMD.  A panel whose series have tables of their own (irregular
    // timestamps: 365 MB for 10 000 series) reads them from HBM, and three waves per SIMD without the prefetch do not cover
    // that latency: 107 -> 97 ms on the irregular bench panel; an aligned panel (tables in L2) is better off with three
    // waves and no prefetch (0.60 against 0.64 s on 100 000 x 730): the launcher picks (FitArgs::harm_pf).
    constexpr bool PREF = HARM != 0 && !SPARSE && PF;
    // XIDX with HARM (round 6): the rows of a panel on a timestamp lattice keep t, y, the segment word and the lattice point
    // (22 bytes); the base pairs are the lattice point's, gathered from a table every series shares (FitArgs::Bu: 32 bytes
    // per point, cache-resident).  The point of row q - 1 is requested with row q, so the gather never waits for its index.
    constexpr int BS = XIDX ? 1 : W;            // stride between the seasonalities of a row's base pairs
    struct RowIn { unsigned cwv; int un; double ti, yi; double2 bp[NS]; };
    auto row_fetch = [&](int q, int u, RowIn &ri) {
        const int idx = q * W + lane;
        ri.cwv = (unsigned)sv.cw[idx]; ri.ti = sv.tw[idx]; ri.yi = sv.yw[idx];
        const double2 *bq = XIDX ? reinterpret_cast<const double2 *>(sv.Bu) + (size_t)u * NS
                                 : reinterpret_cast<const double2 *>(sv.Bw) + (size_t)q * NS * W + lane;
#pragma unroll
        for (int se = 0; se < NS; ++se) ri.bp[se] = bq[se * BS];
        ri.un = (XIDX && q > 0) ? sv.uw[idx - W] : 0;
    };
    RowIn rin;
    if constexpr (PREF) { if (NT > 0) row_fetch(NT - 1, XIDX ? sv.uw[(NT - 1) * W + lane] : 0, rin); }
    for (int q = NT - 1; q >= 0; --q) {
        RowIn rcur;
        if constexpr (PREF) { rcur = rin; if (q > 0) row_fetch(q - 1, rcur.un, rin); }     // (two steps ahead: measured, no better)
        if (q < sv.cnt) {
            const int idx = q * W + lane;
            const unsigned cwv = PREF ? rcur.cwv : (unsigned)sv.cw[idx];
            const int c = (int)(cwv & 0xffu), cprev = (int)(cwv >> 8);
            const double ti = PREF ? rcur.ti : sv.tw[idx];
            const double yi = PREF ? rcur.yi : sv.yw[idx];
            constexpr int XS = XIDX ? 1 : W;      // stride between the columns of a design row
            const double *xp = (XIDX && HARM != 0) ? nullptr : (XIDX ? sv.Xu + (size_t)sv.uw[idx] * KP : sv.Xw + (size_t)q * XCOLS * W + lane);
            double x[(HOLD && HARM == 0) ? KP : 1];
            double xa = 0.0, xm = 0.0;
            double2 bp[NS];
            double xd[NXD > 0 ? NXD : 1];
            if constexpr (HARM != 0) {
                const double2 *bq = PREF ? nullptr : (XIDX ? reinterpret_cast<const double2 *>(sv.Bu) + (size_t)sv.uw[idx] * NS
                                                           : reinterpret_cast<const double2 *>(sv.Bw) + (size_t)q * NS * W + lane);
#pragma unroll
                for (int se = 0; se < NS; ++se) bp[se] = PREF ? rcur.bp[se] : bq[se * BS];
                double ch = 0.0;
                harm_row<HARM>(bp, [&](int j, double v) { ch = __builtin_fma(v, bs[j], ch); });
                if (NXD > 0 && sv.n_xd > 0) {
#pragma unroll
                    for (int j = 0; j < NXD; ++j) xd[j] = xp[(KF + j) * XS];
#pragma unroll
                    for (int j = 0; j < NXD; ++j) ch = __builtin_fma(xd[j], bs[KF + j], ch);
                }
                if (MODE == 0) xa = ch; else xm = ch;
            } else if (HOLD) {
#pragma unroll
                for (int j = 0; j < (HOLD ? KP : 1); ++j) {
                    if (XIDX) {       // gathered row: 16-byte loads (rows are KP*8 bytes, KP even)
                        if ((j & 1) == 0) {
                            const double2 v2 = reinterpret_cast<const double2 *>(xp)[j >> 1];
                            x[j] = v2.x; x[HOLD ? j + 1 : 0] = v2.y;
                        }
                    } else {
                        x[j] = xp[j * XS];
                    }
                }
#pragma unroll
                for (int j = 0; j < (HOLD ? KP : 1); ++j) {
                    const double bj = bs[j];
                    if (MODE == 0) xa = __builtin_fma(x[j], bj, xa);
                    else if (MODE == 1) xm = __builtin_fma(x[j], bj, xm);
                    else { if (j < Ka) xa = __builtin_fma(x[j], bj, xa); else xm = __builtin_fma(x[j], bj, xm); }
                }
            } else {
#pragma unroll 8
                for (int j = 0; j < KP; ++j) {
                    const double xv = xp[j * XS], bv = lds.th[3 + S + j];
                    if (MODE == 0) xa = __builtin_fma(xv, bv, xa);
                    else if (MODE == 1) xm = __builtin_fma(xv, bv, xm);
                    else { if (j < Ka) xa = __builtin_fma(xv, bv, xa); else xm = __builtin_fma(xv, bv, xm); }
                }
            }
            if constexpr (SPARSE) {
                // the ones of this row, ascending column: fma(1, b, chain) = chain + b.  (sp_cur: the lane's next
                // entry; it stays on the row's first one for the second walk below.)
                unsigned c2 = sp_cur;
                int i2 = sp_i;
                while (__any(c2 != SP_END && (int)(c2 & 127u) == q)) {
                    if (c2 != SP_END && (int)(c2 & 127u) == q) {
                        const double bv = lds.th[3 + S + SP_DENSE + (int)((c2 >> 7) & 63u)];
                        if (MODE == 0) xa = xa + bv; else xm = xm + bv;
                        ++i2;
                        c2 = sv.sp_list[i2 * W + lane];
                    }
                }
            }
            const double ksc = lds.ks[c], mcc = lds.mc[c];
            double gtr, qv = 0.0;
            if (GROWTH == 0) {
                gtr = __builtin_fma(ksc, ti, mcc);
            } else {
                const double z = ksc * (ti - mcc);
                const double e = dm_exp_sel(-z);     // branch-free: the chain can run beside the X.beta chain
                const double sg = 1.0 / (1.0 + e);
                gtr = sv.cap * sg;
                qv = gtr * (1.0 - sg);
            }
            const double opm = 1.0 + xm;
            const double mu = __builtin_fma(gtr, opm, xa);
            const double r = yi - mu;
            sse = __builtin_fma(r, r, sse);
            const double rg = r * gtr;
            if constexpr (HARM != 0) {
                // per-column partial sums: the harmonics once more (same recurrence, same bits).  The empty asm keeps the
                // compiler from recognising the second expansion as the first and holding all 2 x order values of the
                // row across the trend arithmetic (52 registers for yearly + weekly: spills at three waves per SIMD).
#ifndef TSF_HARM_HOLD
#pragma unroll
                for (int se = 0; se < NS; ++se) { asm volatile("" : "+v"(bp[se].x)); asm volatile("" : "+v"(bp[se].y)); }
#endif
                const double wv = (MODE == 0) ? r : rg;
                harm_row<HARM>(bp, [&](int j, double v) { acc[j] = __builtin_fma(v, wv, acc[j]); });
                if (NXD > 0 && sv.n_xd > 0) {
#pragma unroll
                    for (int j = 0; j < NXD; ++j) acc[KF + j] = __builtin_fma(xd[j], wv, acc[KF + j]);
                }
            } else if constexpr (GROUPED) {
                wr[GROUPED ? q : 0] = r;
                if (MODE != 0) wg[(GROUPED && MODE != 0) ? q : 0] = rg;
            } else {
            // batches of 8 columns: all of acc[] stays in registers, only 8 design values in flight
#pragma unroll
            for (int j0 = 0; j0 < KP; j0 += 8) {
                double xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (j0 + u < KP) xv[u] = HOLD ? x[HOLD ? j0 + u : 0] : xp[(j0 + u) * XS];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (j0 + u < KP) {
                        const int j = j0 + u;
                        if (MODE == 0) acc[j] = __builtin_fma(xv[u], r, acc[j]);
                        else if (MODE == 1) acc[j] = __builtin_fma(xv[u], rg, acc[j]);
                        else acc[j] = __builtin_fma(xv[u], (j < Ka) ? r : rg, acc[j]);
                    }
                }
                if (!HOLD) __builtin_amdgcn_sched_barrier(0);
            }
            }
            if constexpr (SPARSE) {
                // fma(1, w, +0) = w: the lane's partial of that column (a column comes once per lane)
                while (__any(sp_cur != SP_END && (int)(sp_cur & 127u) == q)) {
                    if (sp_cur != SP_END && (int)(sp_cur & 127u) == q) {
                        sv.sp_acc[((sp_cur >> 7) & 63u) * SP_E + (sp_cur >> 13)] = (MODE == 0) ? r : rg;
                        ++sp_i;
                        sp_cur = sv.sp_list[sp_i * W + lane];
                    }
                }
            }
            double v = r * opm;
            if (GROWTH == 1) v = v * qv;
            rt1 = __builtin_fma(v, ti, rt1);
            rt2 = rt2 + v;
            for (int j = cprev; j < c; ++j) { lds.tp1[j] = rt1; lds.tp2[j] = rt2; }
        }
    }
    FT_LAP(2);
    // reductions over the time axis
    const double sse_t = bfly_sum(sse);
    const double s1 = suffix_scan(rt1), s2v = suffix_scan(rt2);
    lds.tot1[lane] = s1; lds.tot2[lane] = s2v;
    if (lane == 0) { lds.tot1[W] = 0.0; lds.tot2[W] = 0.0; }
    if constexpr (GROUPED) {
        static_assert(!GROUPED || (KP % 8 == 0 && !XIDX), "grouped column sums: aligned step-major design tiles, KP a multiple of 8");
        // rows past the end of the series: weight 0 (initialised above, never written) against the zero-filled
        // padding of the design tile: fma(0, 0, acc) leaves acc unchanged, as skipping the row does
#pragma unroll 1
        for (int j0 = 0; j0 < KP; j0 += 8) {
            double ga[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) ga[u] = 0.0;
#pragma unroll 2
            for (int q = NT - 1; q >= 0; --q) {
                const double *xp = sv.Xw + ((size_t)q * KP + j0) * W + lane;
                double xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) xv[u] = xp[u * W];
                const double w_r = wr[GROUPED ? q : 0];
                const double w_g = (MODE != 0) ? wg[(GROUPED && MODE != 0) ? q : 0] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u;
                    if (MODE == 0) ga[u] = __builtin_fma(xv[u], w_r, ga[u]);
                    else if (MODE == 1) ga[u] = __builtin_fma(xv[u], w_g, ga[u]);
                    else ga[u] = __builtin_fma(xv[u], (j < Ka) ? w_r : w_g, ga[u]);
                }
            }
            column_sums_g<8>(ga, lds.accR + j0);
        }
    } else {
        column_sums<KP, PPL, L>(acc, lds);
    }
    if constexpr (SPARSE) {
        // the sparse columns' sums: lane c folds the slots of column c in the reduction network's order; a sum that
        // met only zeros in the dense network has been added to +0 there (-0 + 0 = +0): the final + 0.0
        TSF_WAVE_SYNC();
        if (lane < sv.P - 3 - S - SP_DENSE) {
            const unsigned long long pg = sv.sp_prog;
            const int nm = (int)(pg & 7u), root = (int)((pg >> 3) & 7u), kl = (int)((pg >> 6) & 15u);
            double *w = sv.sp_acc + lane * SP_E;
            for (int i = 0; i < nm; ++i) {
                const int d = (int)((pg >> (10 + 6 * i)) & 7u), r2 = (int)((pg >> (13 + 6 * i)) & 7u);
                w[d] = w[d] + w[r2];
            }
            lds.accR[SP_DENSE + lane] = (kl > 0 ? w[root] : 0.0) + 0.0;
        }
    }
    TSF_WAVE_SYNC();
    FT_LAP(3);
    const bool bad_ = eval_tail<GROWTH, PPL>(sp, sv, lds, lds, th, sigma, inv_s2, sse_t, f_out, g);
    FT_LAP(4);
    return bad_;
}

// ---------------------------------------------------------------------------------------
// Stan L-BFGS (BFGSMinimizer<LBFGSUpdate>::step, WolfeLineSearch, WolfLSZoom, CubicInterp)
// ---------------------------------------------------------------------------------------

// a / 3.0, correctly rounded, without the ~22-instruction IEEE division sequence: with y = RN(1/3),
// q = RN(a y), r = a - 3 q (exact: one fma), RN(q + r y) IS RN(a / 3) (Markstein's correction step; the
// divisor's significand is not all ones), wherever nothing over- or underflows -- so only for
// 2^-900 <= |a| <= 2^1000, and the wave takes the true division if any lane holds anything else (zeros, NaN,
// infinities included).  Checked against a / 3.0 on 4e8 operands (round-4 notes in DESIGN.md).
__device__ __forceinline__ double div3_rn(double a)
{
    const double aa = __builtin_fabs(a);
    const bool safe = aa >= 0x1p-900 && aa <= 0x1p1000;
    if (__all(safe)) {
        const double y = 0x1.5555555555555p-2;
        const double q = a * y;
        const double r = __builtin_fma(-3.0, q, a);
        return __builtin_fma(r, y, q);
    }
    return a / 3.0;
}

// value a/b/c/d in lane 0/1/2/3 (other lanes: d)
__device__ __forceinline__ double lanes4(double a, double b, double c, double d)
{
    const int l = (int)threadIdx.x & (W - 1);
    return l == 0 ? a : (l == 1 ? b : (l == 2 ? c : d));
}

// Stan's CubicInterp (cubic through f(0)=0, f'(0)=df0, f(x1)=f1, f'(x1)=df1; minimiser in
// [loX, hiX]).  All operands are wave-uniform scalars, and an fp64 division is a ~150-cycle
// dependent sequence, so the INDEPENDENT divisions / polynomial evaluations are done in
// different lanes of one vector operation (three quotients of c3/c2, the two roots, the four
// candidate points) and read back with v_readlane: 5 division latencies instead of 13.  Every
// element goes through exactly the operations of the scalar expression (oracle cubic_interp6).
__device__ __forceinline__ double cubic_interp6(double df0, double x1, double f1, double df1,
                                                double loX, double hiX)
{
    const double x1sq = x1 * x1;
    const double q = lanes4(-12.0 * f1 + 6.0 * x1 * (df0 + df1), -(4.0 * df0 + 2.0 * df1), 6.0 * f1, 1.0) /
                     lanes4(x1sq * x1, x1, x1sq, 1.0);
    const double c3 = readlane_f64(q, 0);
    const double c2 = readlane_f64(q, 1) + readlane_f64(q, 2);
    const double c1 = df0;
    const double t_s = __builtin_sqrt(c2 * c2 - 2.0 * c1 * c3);
    const double sr = lanes4(-(c2 + t_s), -(c2 - t_s), 0.0, 0.0) / c3;
    const double s1 = readlane_f64(sr, 0), s2 = readlane_f64(sr, 1);
    const double xs = lanes4(loX, hiX, s1, s2);
    const double pv = xs * (xs * (div3_rn(xs * c3) + c2) / 2.0 + c1);
    double tmpF, minF, minX;
    minF = readlane_f64(pv, 0);
    minX = loX;
    tmpF = readlane_f64(pv, 1);
    if (tmpF < minF) { minF = tmpF; minX = hiX; }
    if (loX < s1 && s1 < hiX) {
        tmpF = readlane_f64(pv, 2);
        if (tmpF < minF) { minF = tmpF; minX = s1; }
    }
    if (loX < s2 && s2 < hiX) {
        tmpF = readlane_f64(pv, 3);
        if (tmpF < minF) { minF = tmpF; minX = s2; }
    }
    return minX;
}

// optimiser settings by VALUE in the kernel arguments (scalar, invariant): reading them through
// the DevSpec pointer cost a chain of dependent memory loads in every termination test
struct LbfgsOpts {
    double init_alpha, tol_obj, tol_rel_obj_eps, tol_grad, tol_rel_grad_eps, tol_param;
    int max_iter, history;
};

struct FitArgs {
    const DevSpec *sp;
    LbfgsOpts opt;
    int64_t N;
    int aligned, NTmax, theta_stride;
    const GridTab *gtab;
    const SeriesTab *stab;
    const double *tw, *yw, *Xw;
    const uint16_t *cw;
    // outputs
    double *theta, *y_scale, *fval;
    int32_t *status, *n_iter, *n_eval;
    tsf_grid_info *grid_out;
    // eval-only mode
    const double *theta_in;
    double *grad_out;
    // ragged panel whose timestamps all lie on one lattice base + u*step: the design row of a
    // timestamp is a function of the timestamp only, so ONE table over the lattice serves every
    // series (L2 resident) instead of a per-series copy streamed from HBM at every evaluation
    // base pairs of the Fourier columns (setup_grid_kernel; fit_kernel<..., HARM>): [grid][NTmax][bw_ns][64][2]
    const double *Bw;
    int bw_ns, harm;                    // seasonalities per row of Bw; the model's harmonic structure (harm_code), 0 = none compiled
    int coop_harm;                      // the cooperative kernel's rows from the base pairs too (harm != 0 and no dense column behind the Fourier block)
    int harm_pf;                        // HARM: the two-waves-per-SIMD kernel with the row prefetch (tables per series, read from HBM)
    int opt_coop_sparse;                // the sparse-column kernel's tail on the sparse cooperative kernel (TSF_OPT_SPARSE_EXTRA != 2 ... tests: off)
    const int32_t *uw;                  // [grid][NTmax][64] lattice row of each series row (zero where a lane's chunk has no such row)
    const double *Xu;                   // [U][KP]
    const double *Bu;                   // [U][bw_ns][2]: the lattice points' base pairs (fit_kernel<..., XIDX, ..., HARM>), or null
    int xidx;
    // launch guard: when run_flag is set the kernel runs only if (*run_flag != 0) == (run_if != 0)
    // (the one-wave kernel as the fallback of the matrix-core kernel, decided on the device)
    const int *run_flag;
    int run_if;
    // cooperative tail (tsf_coop_kernels.h): where coop_ctl is set, a fit that is still running when
    // the launch has started its last block (coop_after < 0) or that has used coop_after evaluations
    // writes its optimiser state to a checkpoint slot and returns; fit_coop_kernel finishes it with a
    // whole workgroup.  ctl[0] blocks started, ctl[1] slots taken, ctl[2] queue head of the tail,
    // ctl[3] blocks finished or suspended.
    int *coop_ctl;
    int32_t *coop_list;                 // [coop_max] series of each slot
    double *coop_slots;                 // [coop_max][coop_stride]
    int coop_max, coop_stride, coop_after, coop_blocks;
    int coop_tail_at;                   // tail rule: suspend once no more fits than this are still running (tsf_api.hip)
    // quadratic-form fit (tsf_quad_kernels.h): the residual passes read the caller's y rows (series n at y_raw[y_T * n + i], or
    // at y_offsets[n] + i) and scale them in the register; yw is then not written at all (null)
    const void *y_raw;
    const int64_t *y_offsets;
    int y_raw_dtype, y_T;
    int map_harm;                       // converge = MAP: harm_code of the model when map_kernel may read base-pair rows (Bw built), else 0
    int map_max_iter;                   // converge = MAP (tsf_map_kernels.h): iteration limit and KKT tolerance of the continuation
    double map_tol;
    // scheduling hint (tsf_set_cost_hints): the q-th series the launch starts is order[q]; null = q
    const int32_t *order;
    // ragged panels whose series SHARE timestamp vectors (round 4): grid_of[n] = the grid (timestamp vector with its
    // derived tables: tw, cw, Xw, GridTab) of series n, grids numbered 0 .. G-1 in order of first appearance; null =
    // one grid per series (grid n).  Found on the host by tsf_fit_ragged (identical vectors only, models without
    // explicit columns); the tables of a grid are then built once and shared, as on an aligned panel.
    const int32_t *grid_of;
    // sparse indicator columns (SP_* above): per grid the lanes' ones [grid][SP_M][64] and the columns' fold programs
    // [grid][SP_MAXC]; sp_flag: cleared by sparse_extra_kernel when a grid does not qualify (fit_kernel<..., SPARSE> is
    // launched with run_flag = sp_flag, run_if = 1, the dense kernel behind it with run_if = 0)
    const uint32_t *sp_meta;
    const unsigned long long *sp_prog;
    int *sp_flag;
};

__device__ __forceinline__ int64_t grid_index(const FitArgs &a, int64_t n)
{
    return a.aligned ? 0 : (a.grid_of ? (int64_t)a.grid_of[n] : n);
}

// ---- checkpoint of a suspended fit (written by fit_kernel, read by fit_coop_kernel) -----------
// slot layout (doubles): [0, COOP_VARS_D) CoopVars; [COOP_VARS_D, +MAXH) rho; then the vectors
// x, g, p of the current iterate (0..2; 3..5 unused) and the L-BFGS history S (6..), Y (6+MAXH..),
// each [PPL][64]
constexpr int COOP_MAX_NT = 64;         // longest series the cooperative tail takes: 64 steps per chunk (4096 rows)
constexpr int COOP_DIRECT = -2;          // FitArgs::coop_after: no one-wave phase, fit_coop_kernel fits every series from scratch
constexpr int COOP_VARS_D = 32;
constexpr int COOP_NVEC = 6 + 2 * MAXH;

struct CoopVars {
    double fk, fk1, alpha, gammak, dfp, c1dfp, c2dfp, alpha0, prevF, prevDFp;
    double alo, aloF, aloDFp, ahi, ahiF, ahiDFp, gp;
    int32_t itNum, resetB, hist_len, hist_head, nits, lsRestarts, zoom, zit, gp_valid, pk1_scaled,
        n_eval, pad_;
};
static_assert(sizeof(CoopVars) <= COOP_VARS_D * sizeof(double), "CoopVars outgrew its slot header");

__host__ __device__ constexpr int coop_slot_doubles(int PPL) { return COOP_VARS_D + MAXH + COOP_NVEC * PPL * W; }

template <int PPL>
__device__ __forceinline__ void coop_put_vec(double *slot, int idx, const double (&v)[PPL])
{
#pragma unroll
    for (int s = 0; s < PPL; ++s) slot[COOP_VARS_D + MAXH + (idx * PPL + s) * W + lane_id()] = v[s];
}
template <int PPL>
__device__ __forceinline__ void coop_get_vec(const double *slot, int idx, double (&v)[PPL])
{
#pragma unroll
    for (int s = 0; s < PPL; ++s) v[s] = slot[COOP_VARS_D + MAXH + (idx * PPL + s) * W + lane_id()];
}

// suspend now?  coop_after >= 0: once that many evaluations are spent (tests, latency mode).
// Otherwise in the tail of the launch: every block has been started AND no more fits are still running
// than there are compute units (looked at every fourth evaluation).  While more are running the
// one-wave kernel has the higher aggregate rate (R waves at one evaluation per ~16 us each against
// n_cu workgroups at one per ~8.4 us); from there on every remaining fit gets a workgroup of its own
// at once, which halves its time per evaluation.  ctl[0] blocks started, ctl[3] blocks finished or suspended.
__device__ __forceinline__ bool coop_should_suspend(const FitArgs &a, int n_eval)
{
    if (a.coop_after >= 0) return n_eval >= a.coop_after;
    if ((n_eval & 3) != 0) return false;
    const int started = __hip_atomic_load(&a.coop_ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (started < (int)a.N) return false;
    const int done = __hip_atomic_load(&a.coop_ctl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return started - done <= a.coop_tail_at;
}

template <int KP, int PPL>
__device__ __forceinline__ void make_view(const FitArgs &a, int64_t n, SeriesView &sv)
{
    const int64_t g = grid_index(a, n);
    const GridTab &gt = a.gtab[g];
    sv.T = gt.info.T; sv.NT = gt.info.NT; sv.S = gt.S_fit; sv.S_out = gt.info.S;
    sv.P = 3 + sv.S + a.sp->K;
    int cnt = sv.T - lane_id() * sv.NT;
    cnt = cnt < 0 ? 0 : (cnt > sv.NT ? sv.NT : cnt);
    sv.cnt = cnt;
    sv.tw = a.tw + (size_t)g * a.NTmax * W;
    sv.cw = a.cw + (size_t)g * a.NTmax * W;
    sv.Xw = a.Xw + (size_t)g * a.NTmax * KP * W;
    sv.uw = a.uw + (size_t)g * a.NTmax * W;
    sv.Xu = a.Xu;
    sv.Bu = a.Bu;
    sv.Bw = a.Bw ? a.Bw + (size_t)g * a.NTmax * a.bw_ns * 2 * W : nullptr;
    sv.n_xd = 0;
    sv.yw = a.yw + (size_t)n * a.NTmax * W;
    sv.y_raw = nullptr;
    sv.Lj = gt.Lj;
    sv.t_change = gt.info.t_change;
    sv.cap = a.stab[n].cap;
    sv.tau = a.sp->tau;
    sv.n_eval = 0;
    set_lane_tables<PPL>(a.sp, sv);
}

// theta (internal order, registers) -> caller layout [k,m,log sigma,delta[n_cp],beta[K]]
// (every slot of the row is written exactly once: fitted entries, zeros elsewhere).
// A series fitted on the dummy changepoint (sv.S = 1, sv.S_out = 0) has no delta slot there:
// FOLD = true (fitted parameters) stores k + delta as k -- Prophet.fit's `k = k + delta;
// delta = 0` when there are no changepoints --, FOLD = false (a gradient) just drops the entry.
template <int PPL, bool FOLD = true>
__device__ __forceinline__ void store_theta(const FitArgs &a, const SeriesView &sv, int64_t n,
                                            const double (&x)[PPL], double *dst)
{
    const int n_cp = a.sp->n_cp;
    double *out = dst + (size_t)n * a.theta_stride;
    for (int i = lane_id(); i < a.theta_stride; i += W) {
        bool fitted = i < 3 + sv.S_out;
        if (i >= 3 + n_cp && i < 3 + n_cp + a.sp->K) fitted = true;
        if (!fitted) out[i] = 0.0;
    }
    const double d0 = readlane_f64(x[0], 3);
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const int p = lane_id() + s * W;
        double v = x[s];
        if (FOLD && p == 0 && sv.S_out != sv.S) v = v + d0;
        if (p < 3 + sv.S_out) out[p] = v;
        else if (p >= 3 + sv.S && p < sv.P) out[3 + n_cp + a.sp->perm[p - 3 - sv.S]] = v;
    }
}

template <int PPL>
__device__ __forceinline__ void load_theta(const FitArgs &a, const SeriesView &sv, int64_t n,
                                           const double *src, double (&x)[PPL])
{
    const int n_cp = a.sp->n_cp;
    const double *in = src + (size_t)n * a.theta_stride;
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const int p = lane_id() + s * W;
        double v = 0.0;
        if (p < 3 + sv.S_out) v = in[p];
        else if (p >= 3 + sv.S && p < sv.P) v = in[3 + n_cp + a.sp->perm[p - 3 - sv.S]];
        x[s] = v;
    }
}

// eval-only kernel (parity tests): f and gradient at a caller-supplied theta
template <int KP, int GROWTH, int MODE, int PPL>
__global__ __launch_bounds__(64) void eval_kernel(FitArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    WaveLds<KP, PPL> &lds = *reinterpret_cast<WaveLds<KP, PPL> *>(smem);
    if ((int64_t)blockIdx.x >= a.N) return;
    const int64_t n = a.order ? (int64_t)a.order[blockIdx.x] : (int64_t)blockIdx.x;
    SeriesView sv;
    make_view<KP, PPL>(a, n, sv);
    for (int i = threadIdx.x; i < TSF_MAX_P + W; i += W) lds.th[i] = 0.0;
    TSF_WAVE_SYNC();
    double x[PPL], g[PPL], f;
    load_theta<PPL>(a, sv, n, a.theta_in, x);
    FT_DECL;
    const bool bad = eval_fg<KP, GROWTH, MODE, PPL>(a.sp, sv, lds, x, f, g FT_PASS);
    store_theta<PPL, false>(a, sv, n, g, a.grad_out);
    if (threadIdx.x == 0) { a.fval[n] = f; a.status[n] = bad ? 1 : 0; }
}

#ifndef TSF_FIT_WPS
#define TSF_FIT_WPS 1
#endif
#ifndef TSF_HARM_WPS
#define TSF_HARM_WPS 3      // waves per SIMD the HARM kernels are compiled for (<= 168 registers)
#endif
#ifndef TSF_HARM_SPARSE_WPS
#define TSF_HARM_SPARSE_WPS 2   // ... and their sparse-column form (the entry lists' cursors and the two-slot optimiser vectors
#endif                          // do not fit 168 registers: at three waves per SIMD it spills 69 of them and loses to the table kernel)
template <int KP, int GROWTH, int MODE, int PPL, bool XIDX = false, int GNTR = 0, bool SPARSE = false, int HARM = 0, bool PF = false>
__global__ __launch_bounds__(64, HARM != 0 ? ((SPARSE || PF) ? TSF_HARM_SPARSE_WPS : TSF_HARM_WPS) : ((GNTR > 0 || SPARSE) ? 2 : TSF_FIT_WPS)) void fit_kernel(FitArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int KL = SPARSE ? 64 : KP;            // SPARSE: tables and LDS of the 64-column model, registers of the 28-column one
    using WL = WaveLds<KL, PPL>;
    WL &lds = *reinterpret_cast<WL *>(smem);
    if ((int64_t)blockIdx.x >= a.N) return;
    const int64_t n = a.order ? (int64_t)a.order[blockIdx.x] : (int64_t)blockIdx.x;
    if (a.run_flag && (*a.run_flag != 0) != (a.run_if != 0)) return;
    const int lane = threadIdx.x;
    if (a.coop_ctl) atomicAdd(&a.coop_ctl[0], lane == 0 ? 1 : 0);      // (branch-free: see coop_should_suspend)
    bool coop_try = a.coop_ctl != nullptr;
    int coop_ticket = -1;
    const DevSpec *sp = a.sp;
    SeriesView sv;
    make_view<KL, PPL>(a, n, sv);
    if constexpr (HARM != 0) {
        const int kd = a.sp->K < KP ? a.sp->K : KP;         // dense columns of the model
        sv.n_xd = kd - harm_kf(HARM);
    }
    if constexpr (SPARSE) {
        const int64_t g = grid_index(a, n);
        sv.sp_prog = lane < SP_MAXC ? a.sp_prog[(size_t)g * SP_MAXC + lane] : 0ull;
        sv.sp_acc = lds.d1;                  // [SP_MAXC][SP_E] over d1, d2, rb, ab (written by eval_tail after the fold)
        unsigned short *lst = reinterpret_cast<unsigned short *>(smem + wave_lds_bytes<KL, PPL>(a.opt.history));
        for (int e = 0; e < SP_M; ++e) lst[e * W + lane] = (unsigned short)a.sp_meta[((size_t)g * SP_M + e) * W + lane];
        lst[SP_M * W + lane] = (unsigned short)SP_END;
        sv.sp_list = lst;
    }
    for (int i = threadIdx.x; i < TSF_MAX_P + W; i += W) lds.th[i] = 0.0;
    TSF_WAVE_SYNC();
    const SeriesTab st = a.stab[n];
    if (lane == 0) {
        a.y_scale[n] = st.y_scale;
        if (!a.aligned || n == 0) a.grid_out[a.aligned ? 0 : n] = a.gtab[grid_index(a, n)].info;
    }

    double xk[PPL], gk[PPL], pk[PPL], xk1[PPL], gk1[PPL], pk1[PPL];
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const int p = lane + s * W;
        xk[s] = (p == 0) ? st.k0 : (p == 1 ? st.m0 : 0.0);
        gk[s] = 0.0; pk[s] = 0.0; xk1[s] = xk[s]; gk1[s] = 0.0; pk1[s] = 0.0;
    }
    if (st.status0 != 0) {
        // fbprophet raises (too few rows / cap <= floor) or skips optimisation (constant y)
        if (st.status0 == TSF_ST_CONSTANT) {
#pragma unroll
            for (int s = 0; s < PPL; ++s) if (lane + s * W == 2) xk[s] = -20.72326583694641;
        }
        store_theta<PPL>(a, sv, n, xk, a.theta);
        if (lane == 0) { a.status[n] = st.status0; a.n_iter[n] = 0; a.n_eval[n] = 0; a.fval[n] = 0.0; }
        if (a.coop_ctl) atomicAdd(&a.coop_ctl[3], lane == 0 ? 1 : 0);
        return;
    }

    const int H = a.opt.history > MAXH ? MAXH : a.opt.history;
    const double eps = 2.220446049250313e-16;
    const double c1 = 1e-4, c2 = 0.9, minAlpha = 1e-12, min_range = 1e-16;
    const int maxLSIts = 20, maxLSRestarts = 10;

    double fk = 0.0, fk1 = 0.0, alpha = a.opt.init_alpha, gammak = 1.0;
    int itNum = 0, ret = 0, resetB = 0, hist_len = 0, hist_head = 0;
    // line-search state
    double dfp = 0, c1dfp = 0, c2dfp = 0, alpha0 = 0, prevF = 0, prevDFp = 0;
    double alo = 0, aloF = 0, aloDFp = 0, ahi = 0, ahiF = 0, ahiDFp = 0;
    int nits = 0, lsRestarts = 0, zoom = 0, zit = 0;
    // g.p of the current / previous iterate: computed once per iterate and carried (see
    // fit_one_quad)
    double gp = 0.0;
    bool gp_valid = false, pk1_scaled = false;

    enum { ST_INIT = 0, ST_START_ITER, ST_START_LS, ST_LS_PRE, ST_LS_EVAL };
    int stage = ST_INIT;
    FT_DECL;
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
                // guard against a line search that never settles (oracle cn_lbfgs eval_limit)
                if (sv.n_eval >= 64 * a.opt.max_iter + 1024) { ret = TSF_ST_EVAL_LIMIT; break; }
                if (coop_try && coop_should_suspend(a, sv.n_eval)) {
                    // hand this fit to the cooperative tail (the ticket fetch is branch-free: every
                    // lane adds, lane 0 adds 1); the state is written after the loop
                    coop_ticket = __builtin_amdgcn_readfirstlane(atomicAdd(&a.coop_ctl[1], lane == 0 ? 1 : 0));
                    if (coop_ticket < a.coop_max) break;
                    coop_ticket = -1;
                    coop_try = false;           // no slot left: this fit stays where it is
                }
#pragma unroll
                for (int s = 0; s < PPL; ++s) xk1[s] = __builtin_fma(alpha, pk[s], xk[s]);
            }
            double f1;
            FT_LAP(0);
            const bool bad = eval_fg<KP, GROWTH, MODE, PPL, XIDX, WL, GNTR, SPARSE, HARM, PF>(sp, sv, lds, xk1, f1, gk1 FT_PASS);
            f1 = uniform_f64(f1);       // every lane holds the same bits: let the compiler know (scalar branches)
            if (stage == ST_INIT) {
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
                // g.g, s.s, y.s, y.y: one four-fold butterfly (lanes 0, 2, 1, 3); the two square roots
                // and the three quotients one lane each (see fit_one_quad)
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
                    int slot;
                    // (ring indices wrap by comparison: `% H` with a run-time H is an integer division)
                    if (hist_len < H) { slot = hist_head + hist_len; if (slot >= H) slot -= H; hist_len++; }
                    else { slot = hist_head; hist_head = hist_head + 1; if (hist_head >= H) hist_head -= H; }
                    if (lane == 0) lds.rho[slot] = rho_new;
#pragma unroll
                    for (int s = 0; s < PPL; ++s) {
                        lds.SY[((2 * slot) * PPL + s) * W + lane] = sk[s];
                        lds.SY[((2 * slot + 1) * PPL + s) * W + lane] = yk[s];
                    }
                }
                TSF_WAVE_SYNC();
#pragma unroll
                for (int s = 0; s < PPL; ++s) pk[s] = -gk[s];
                for (int h = hist_len - 1; h >= 0; --h) {
                    int slot = hist_head + h;
                    if (slot >= H) slot -= H;
                    double si[PPL], yi[PPL];
#pragma unroll
                    for (int s = 0; s < PPL; ++s) {
                        si[s] = lds.SY[((2 * slot) * PPL + s) * W + lane];
                        yi[s] = lds.SY[((2 * slot + 1) * PPL + s) * W + lane];
                    }
                    const double aa = lane63(lds.rho[slot] * pdot_l63<PPL>(si, pk));
#pragma unroll
                    for (int s = 0; s < PPL; ++s) pk[s] = __builtin_fma(-aa, yi[s], pk[s]);
                    if (lane == 0) lds.alphas[h] = aa;
                }
                TSF_WAVE_SYNC();
#pragma unroll
                for (int s = 0; s < PPL; ++s) pk[s] = pk[s] * gammak;
                for (int h = 0; h < hist_len; ++h) {
                    int slot = hist_head + h;
                    if (slot >= H) slot -= H;
                    double si[PPL], yi[PPL];
#pragma unroll
                    for (int s = 0; s < PPL; ++s) {
                        si[s] = lds.SY[((2 * slot) * PPL + s) * W + lane];
                        yi[s] = lds.SY[((2 * slot + 1) * PPL + s) * W + lane];
                    }
                    const double cc = lane63(lds.alphas[h] - lds.rho[slot] * pdot_l63<PPL>(yi, pk));
#pragma unroll
                    for (int s = 0; s < PPL; ++s) pk[s] = __builtin_fma(cc, si[s], pk[s]);
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
    if (a.coop_ctl) atomicAdd(&a.coop_ctl[3], lane == 0 ? 1 : 0);       // finished or suspended
    if (coop_ticket >= 0) {
        // suspended: the whole optimiser state goes to the slot; fit_coop_kernel resumes at this
        // line-search evaluation
        double *slot = a.coop_slots + (size_t)coop_ticket * a.coop_stride;
        if (lane == 0) {
            CoopVars cv;
            cv.fk = fk; cv.fk1 = fk1; cv.alpha = alpha; cv.gammak = gammak; cv.dfp = dfp;
            cv.c1dfp = c1dfp; cv.c2dfp = c2dfp; cv.alpha0 = alpha0; cv.prevF = prevF; cv.prevDFp = prevDFp;
            cv.alo = alo; cv.aloF = aloF; cv.aloDFp = aloDFp; cv.ahi = ahi; cv.ahiF = ahiF; cv.ahiDFp = ahiDFp;
            cv.gp = gp; cv.itNum = itNum; cv.resetB = resetB; cv.hist_len = hist_len; cv.hist_head = hist_head;
            cv.nits = nits; cv.lsRestarts = lsRestarts; cv.zoom = zoom; cv.zit = zit;
            cv.gp_valid = gp_valid ? 1 : 0; cv.pk1_scaled = pk1_scaled ? 1 : 0; cv.n_eval = sv.n_eval; cv.pad_ = 0;
            *reinterpret_cast<CoopVars *>(slot) = cv;
            a.coop_list[coop_ticket] = (int32_t)n;
        }
        if (lane < MAXH) slot[COOP_VARS_D + lane] = lds.rho[lane];
        coop_put_vec<PPL>(slot, 0, xk); coop_put_vec<PPL>(slot, 1, gk); coop_put_vec<PPL>(slot, 2, pk);
        // (x, g, p of the previous iterate are dead at a line-search evaluation: the evaluation and the
        // acceptance that follows overwrite all three before anything reads them)
        for (int h = 0; h < H; ++h) {
            double hs_[PPL], hy_[PPL];
#pragma unroll
            for (int s = 0; s < PPL; ++s) { hs_[s] = lds.SY[((2 * h) * PPL + s) * W + lane]; hy_[s] = lds.SY[((2 * h + 1) * PPL + s) * W + lane]; }
            coop_put_vec<PPL>(slot, 6 + h, hs_); coop_put_vec<PPL>(slot, 6 + MAXH + h, hy_);
        }
        return;
    }
    store_theta<PPL>(a, sv, n, xk, a.theta);
    if (lane == 0) { a.status[n] = ret; a.n_iter[n] = itNum; a.n_eval[n] = sv.n_eval; a.fval[n] = fk; }
    FT_LAP(0);
    FT_FLUSH(a.grad_out, n);
}

}  // namespace tsf
