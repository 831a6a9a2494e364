// tsf_common.h -- gfx950 kernels for the batched Prophet-model MAP fit / predict.
//
// Replaces, for a whole panel at once, what the reference does one series at a time inside
// its grouped-map pandas_udf: fbprophet 0.5 `Prophet.fit` (-> pystan L-BFGS on prophet.stan)
// at /root/reference/src/jobs/prophet_modeler.py:65-66 and `Prophet.predict` at
// /root/reference/src/jobs/prophet_scorer.py:70.
//
// Execution model: ONE WAVEFRONT (64 lanes) PER SERIES.
//   * time axis: lane L owns the contiguous chunk of rows [L*NT, (L+1)*NT), NT = ceil(T/64);
//     panel arrays are stored "step-major" ([q][lane], row = lane*NT + q) so that every
//     per-step access of the wave is one coalesced 512-byte transaction;
//   * parameter axis: parameter p of theta = [k, m, log sigma, delta[S], beta[K]] lives in
//     lane p%64, register slot p/64 (P <= 128), so L-BFGS vector updates are one or two
//     scalar ops per lane and dot products are a per-lane partial + xor-butterfly;
//   * the whole optimisation (Stan's L-BFGS + Wolfe line search with zoom) runs inside one
//     launch; the host never sees an iteration.
//
// CANONICAL ARITHMETIC.  Every floating-point operation below is performed in a fixed order
// (chunk partials accumulated with fma from the last row of a chunk to the first, butterfly
// offsets 1,2,4,8,16,32, Hillis-Steele suffix scan, sequential column sums over chunks ...),
// with fma only where written (`-ffp-contract=off`) and transcendentals from tsf_detmath.h.
// oracle/prophet_canon.c performs the identical sequence on the CPU; tests require the two to
// agree to the last bit.  DO NOT "simplify" an expression here without changing the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tsf_dev.h"
#include "tsf_detmath.h"

namespace tsf {

constexpr int W = 64;
constexpr int MAXH = 8;                 // max L-BFGS history
constexpr int NTAB = TSF_MAX_S + 4;     // per-changepoint tables

// Sparse indicator columns (round 4, eval_fg<..., SPARSE>): a model with more than 28 design columns whose columns
// from the 29th on are 0/1 indicators that are almost always 0 -- holidays: 30 columns, 60 ones among 21 900 entries
// on BASELINE cfg4 -- streams 64-column rows twice per evaluation to multiply zeros.  Adding a zero is exact
// (fma(0, b, a) = a for finite b; a + 0 = a), so the dense chains and the reduction network collapse onto the few
// entries that exist WITHOUT changing a bit: the first SP_DENSE columns run as the 28-column kernel runs them (row and
// accumulators in registers), the ones of a lane's rows stand in a short list (row step, column, slot) in LDS, last row
// first, which the lane walks with a cursor as the row loop descends: a one adds its coefficient to the row's chain
// after the dense columns (ascending column: the chain's order) and hands r g to a slot of a small LDS array, and the
// lane that owns a column folds that column's <= SP_E slots in the order of the reduction network (column_sums: lane
// bits 5, 4, 0, 1, 2, 3), a precomputed list of merges.  sparse_extra_kernel (tsf_aux_kernels.h) builds the tables and
// sets FitArgs::sp_flag if a grid does not qualify (a value other than 0 / 1, more than SP_M ones in a lane's rows, a
// column twice in one lane, more than SP_E lanes per column); the dense kernel, launched behind with the opposite
// guard, then does the work.
constexpr int SP_DENSE = 28;            // columns [0, SP_DENSE) dense, [SP_DENSE, K) sparse
constexpr int SP_M = 12;                // entries per lane
constexpr int SP_E = 8;                 // lanes per column
constexpr int SP_MAXC = 32;            // sparse columns at most (their slots fit eval_tail's scratch)
// entry word (16 bits): bits 0-6 row step q, bits 7-12 sparse column, bits 13-15 rank of the lane among the column's
// lanes (slot = column * SP_E + rank); a lane's list ends with SP_END (SP_M + 1 words per lane).  The list lives in
// SP_LDS_BYTES of LDS behind the wave's tables; the slots share the bytes of eval_tail's scratch (WaveLds::d1 .. ab),
// which is written only after the fold has read them.
// program word of a column: bits 0-2 number of merges, 3-5 slot that ends up holding the sum, 6-9 number of lanes,
// then 6 bits per merge from bit 10: dst slot (3), src slot (3)
constexpr unsigned SP_END = 0xffffu;
constexpr int SP_MAX_NT = 127;          // row steps per lane at most (7 bits)
constexpr size_t SP_LDS_BYTES = sizeof(unsigned short) * (SP_M + 1) * 64;
static_assert(SP_MAXC * SP_E <= 4 * (NTAB + 1), "the sparse columns' slots overlay WaveLds::d1 .. ab");

// Model description resident in device memory (built on the host by tsf_api).
struct DevSpec {
    int32_t growth, n_cp, K, Ka, KP, n_seas, n_extra, n_pairs;
    int32_t max_iter, history;
    double cp_range, tau, init_alpha, tol_obj, tol_rel_obj, tol_grad, tol_rel_grad, tol_param;
    int32_t inv_perm[TSF_MAX_P];        // original column -> internal column
    int32_t perm[TSF_MAX_P];            // internal column -> original column
    double prior[TSF_MAX_P];            // prior scale per internal column
    double seas_period[TSF_MAX_SEAS];   // per seasonality: period in days,
    int32_t seas_order[TSF_MAX_SEAS];   // Fourier order,
    int32_t seas_col[TSF_MAX_SEAS];     // original column of sin(1 theta) (columns: sin 1, cos 1, sin 2, cos 2, ...)
    int32_t harm;                       // harmonic structure the residual-form kernel is compiled for (HARM_*), 0 = none
    int32_t pad_;
};

// Canonical design values (round 5; oracle fourier_row): the FIRST harmonic of a seasonality from dm_sincos at
// fbprophet's argument 2 pi t / period, harmonic h + 1 by u[h+1] = 2 cos(theta) u[h] - u[h-1] (one fma per value;
// u[0] = 0 for the sines, 1 for the cosines).  A design row is a function of the base pair (sin theta, cos theta) of
// each seasonality: the residual-form kernel keeps only those per row (FitArgs::Bw) and expands them in registers.
__device__ __forceinline__ double fourier_base_arg(int64_t ds_ns, double period)
{
    const double tdays = (1e-9 * (double)ds_ns) / 86400.0;
    return (2.0 * 3.141592653589793 * tdays) / period;
}
// emit(h, sin(h theta), cos(h theta)) for h = 1 .. order
template <class F>
__device__ __forceinline__ void fourier_harmonics(double s1, double c1, int order, F &&emit)
{
    const double c2 = 2.0 * c1;
    double sp = 0.0, cp = 1.0, sc = s1, cc = c1;
    for (int h = 1; h <= order; ++h) {
        if (h > 1) {
            const double sn = __builtin_fma(c2, sc, -sp), cn = __builtin_fma(c2, cc, -cp);
            sp = sc; cp = cc; sc = sn; cc = cn;
        }
        emit(h, sc, cc);
    }
}

// Harmonic structure of a model as a compile-time code: the Fourier orders of up to three seasonalities, one byte
// each, in column order (fit_kernel<..., HARM>: the kernel that reads the base pairs FitArgs::Bw and expands the
// harmonics in registers).  0 = none (design values streamed from the table Xw).
constexpr int harm_code(int o0, int o1 = 0, int o2 = 0) { return o0 | (o1 << 8) | (o2 << 16); }
constexpr int harm_order(int harm, int s) { return (harm >> (8 * s)) & 255; }
constexpr int harm_ns(int harm) { return (harm_order(harm, 0) > 0) + (harm_order(harm, 1) > 0) + (harm_order(harm, 2) > 0); }
constexpr int harm_kf(int harm) { return 2 * (harm_order(harm, 0) + harm_order(harm, 1) + harm_order(harm, 2)); }
constexpr int HARM_Y10_W3 = harm_code(10, 3);       // yearly + weekly: daily data of two years or more (K = 26)
constexpr int HARM_W3_D4 = harm_code(3, 4);         // weekly + daily: sub-daily data of less than two years (K = 14: the reference's fixture)
constexpr int HARM_W3 = harm_code(3);               // weekly alone (K = 6)

// Per-grid derived tables (one grid per call for aligned panels, one per series for ragged).
struct GridTab {
    tsf_grid_info info;
    int32_t Lj[NTAB];                   // chunk holding the first row with t >= t_change[j]
    // changepoints the FIT runs with: info.S, or 1 when info.S == 0 -- fbprophet then fits the Stan
    // model on a dummy changepoint at t = 0 (set_changepoints: `changepoints_t = np.array([0.])`)
    // and folds its delta into k afterwards (store_theta); the caller never sees it
    int32_t S_fit;
    int32_t pad_;
};

// one value of the caller's y column (f64, f32 or the reference schema's int32: prophet_modeler.py:16)
__device__ __forceinline__ double load_y(const void *y, int dtype, int64_t i)
{
    if (dtype == TSF_Y_F64) return ((const double *)y)[i];
    if (dtype == TSF_Y_F32) return (double)((const float *)y)[i];
    return (double)((const int32_t *)y)[i];
}

struct SeriesTab {
    double y_scale, cap, k0, m0, floor_;
    int32_t status0;                    // 0 ok, TSF_ST_TOO_FEW / TSF_ST_CAP / TSF_ST_CONSTANT
    int32_t pad_;
};

// ---------------------------------------------------------------------------------------
// cross-lane helpers (wave64)
// ---------------------------------------------------------------------------------------

__device__ __forceinline__ double readlane_f64(double v, int lane);

// ---- DPP / permlane primitives (wave64, gfx950) -------------------------------------------
// All of these only MOVE data; the arithmetic tree they implement is written next to each use.

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)          // invalid source lanes read 0.0
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

constexpr int DPP_XOR1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // lane i <-> 7-i within 8  (== xor 4 once quads are uniform)
constexpr int DPP_MIRROR = 0x140;       // lane i <-> 15-i within 16 (== xor 8 once octets are uniform)
#define DPP_ROW_SHL(n) (0x100 + (n))    // lane i reads lane i+n of its 16-lane row

// v_permlane16_swap: rows 1,3 of a <-> rows 0,2 of b.   v_permlane32_swap: a.hi32 <-> b.lo32.
__device__ __forceinline__ void swap16(double &a, double &b)
{
    auto r0 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)r1[0], (int)r0[0]);
    b = __hiloint2double((int)r1[1], (int)r0[1]);
}
__device__ __forceinline__ void swap32(double &a, double &b)
{
    auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)r1[0], (int)r0[0]);
    b = __hiloint2double((int)r1[1], (int)r0[1]);
}

// DPP move that only writes the rows of ROW_MASK / banks of BANK_MASK (4 lanes each); the other
// lanes keep `old`
template <int CTRL, int ROW_MASK, int BANK_MASK = 0xF>
__device__ __forceinline__ double dpp_mov_masked(double old, double v)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_ROW_BCAST15 = 0x142;  // lane 15 of each row -> every lane of the next row
constexpr int DPP_ROW_BCAST31 = 0x143;  // lane 31 -> every lane of rows 2 and 3
constexpr int DPP_ROW_ROR8 = 0x128;     // lane i reads lane (i + 8) mod 16 of its row  (== xor 8)

// xor-butterfly sum, offsets 1,2,4,8,16,32.
// (a + b is commutative, so in the butterfly every lane of a 2^k block holds the same bits after
// stage k; exchanging through mirrors gives the butterfly's values.)  After the four in-row stages
// every lane of row r holds R_r.  Stage 16 is then needed in rows 1 and 3 only (R1 + R0, R3 + R2,
// operands in the butterfly's own/partner order for those lanes) and stage 32 in row 3 only
// ((R3 + R2) + (R1 + R0)): two row-broadcast moves instead of two full swaps, and lane 63 holds
// exactly what the butterfly leaves in every lane.  (The broadcasts also write rows that do not need
// them -- row 0 reads nothing and gets 0 -- so that no `old` operand has to be set up.)
// bfly_sum_l63: the sum in LANE 63 of the returned register (the other lanes hold partial sums);
// lane63(): that lane as a wave-uniform scalar.  Arithmetic on the lane-63 value before the
// v_readlane (a scale factor, say) saves moving the scalar back into a vector register.
__device__ __forceinline__ double bfly_sum_l63(double v)
{
    v = v + dpp_mov<DPP_XOR1>(v);
    v = v + dpp_mov<DPP_XOR2>(v);
    v = v + dpp_mov<DPP_HALF_MIRROR>(v);
    v = v + dpp_mov<DPP_MIRROR>(v);
    v = v + dpp_mov<DPP_ROW_BCAST15>(v);
    v = v + dpp_mov<DPP_ROW_BCAST31>(v);
    return v;
}
__device__ __forceinline__ double lane63(double v)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bfly_sum(double v) { return lane63(bfly_sum_l63(v)); }

// Four butterfly sums at once, bit-identical to four bfly_sum calls.  Stage 1 and 2 are done
// "transposed": a lane keeps two (then one) of the four quantities and sends the others to its
// partner, which needs exactly those -- each partial sum of the butterfly is formed once instead
// of in every lane of its block (own + partner order as in the lane that keeps it).  From stage 4
// on the lanes = 0, 2, 1, 3 (mod 4) carry a, b, c, d.
// bfly_sum4_lanes: the register whose lanes 0, 1, 2, 3 (and every lane = 0..3 mod 4) hold the sums of
// a, c, b, d
__device__ __forceinline__ double bfly_sum4_lanes(double a, double b, double c, double d)
{
    const int lane = (int)threadIdx.x & (W - 1);
    const bool odd = lane & 1, hi2 = lane & 2;
    const double k0 = odd ? c : a, k1 = odd ? d : b;
    const double s0 = odd ? a : c, s1 = odd ? b : d;
    const double r0 = k0 + dpp_mov<DPP_XOR1>(s0);
    const double r1 = k1 + dpp_mov<DPP_XOR1>(s1);
    const double kk = hi2 ? r1 : r0, ss = hi2 ? r0 : r1;
    double v = kk + dpp_mov<DPP_XOR2>(ss);
    // xor 4: banks 0 and 2 (lanes 0-3, 8-11 of a row) read lane + 4, banks 1 and 3 lane - 4
    {
        double t = dpp_mov_masked<DPP_ROW_SHL(4), 0xF, 0x5>(0.0, v);
        t = dpp_mov_masked<0x110 + 4, 0xF, 0xA>(t, v);          // row_shr:4
        v = v + t;
    }
    v = v + dpp_mov<DPP_ROW_ROR8>(v);
    { double x = v, y = v; swap16(x, y); v = x + y; }
    { double x = v, y = v; swap32(x, y); v = x + y; }
    return v;
}
__device__ __forceinline__ void bfly_sum4(double a, double b, double c, double d,
                                          double &sa, double &sb, double &sc, double &sd)
{
    const double v = bfly_sum4_lanes(a, b, c, d);
    sa = readlane_f64(v, 0); sc = readlane_f64(v, 1); sb = readlane_f64(v, 2); sd = readlane_f64(v, 3);
}

// within-row butterfly 1,2,4,8 (used after the 32/16 stages of the column network)
__device__ __forceinline__ double row_bfly_sum(double v)
{
    v = v + dpp_mov<DPP_XOR1>(v);
    v = v + dpp_mov<DPP_XOR2>(v);
    v = v + dpp_mov<DPP_HALF_MIRROR>(v);
    v = v + dpp_mov<DPP_MIRROR>(v);
    return v;
}

__device__ __forceinline__ double bcast(double v, int lane) { return __shfl(v, lane, W); }

__device__ __forceinline__ double uniform_f64(double v)
{
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// value held by `lane` (wave-uniform index) as a scalar
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// inclusive suffix sum over the 64 lanes: Hillis-Steele (1,2,4,8) inside each 16-lane row, then
// the totals of the later rows are added as one carry ((T3), (T2+T3), (T1+(T2+T3))).
__device__ __forceinline__ double suffix_scan(double v)
{
    v = v + dpp_mov<DPP_ROW_SHL(1)>(v);
    v = v + dpp_mov<DPP_ROW_SHL(2)>(v);
    v = v + dpp_mov<DPP_ROW_SHL(4)>(v);
    v = v + dpp_mov<DPP_ROW_SHL(8)>(v);
    const double t1 = readlane_f64(v, 16), t2 = readlane_f64(v, 32), t3 = readlane_f64(v, 48);
    const double s2 = t2 + t3;
    const double s1 = t1 + s2;
    const int row = ((int)threadIdx.x & (W - 1)) >> 4;
    const double carry = (row == 0) ? s1 : (row == 1 ? s2 : (row == 2 ? t3 : 0.0));
    return v + carry;
}

// ---- scans of round 5 (oracle prefix_scan / affine_prefix_scan / affine_suffix_scan) -----------------------------
#define DPP_ROW_SHR(n) (0x110 + (n))    // lane i reads lane i-n of its 16-lane row
constexpr int DPP_WAVE_SHL1 = 0x130;    // lane i reads lane i+1 of the WAVE (lane 63: nothing)
constexpr int DPP_WAVE_SHR1 = 0x138;    // lane i reads lane i-1 of the WAVE (lane 0: nothing)

// inclusive prefix sum over the 64 lanes, the mirror image of suffix_scan: Hillis-Steele (1,2,4,8) inside each 16-lane
// row, then the totals of the earlier rows as one carry ((T0), (T0+T1), ((T0+T1)+T2))
__device__ __forceinline__ double prefix_scan(double v)
{
    v = v + dpp_mov<DPP_ROW_SHR(1)>(v);
    v = v + dpp_mov<DPP_ROW_SHR(2)>(v);
    v = v + dpp_mov<DPP_ROW_SHR(4)>(v);
    v = v + dpp_mov<DPP_ROW_SHR(8)>(v);
    const double t0 = readlane_f64(v, 15), t1 = readlane_f64(v, 31), t2 = readlane_f64(v, 47);
    const double s1 = t0 + t1;
    const double s2 = s1 + t2;
    const int row = ((int)threadIdx.x & (W - 1)) >> 4;
    const double carry = (row == 0) ? 0.0 : (row == 1 ? t0 : (row == 2 ? s1 : s2));
    return v + carry;
}

// Scans of affine maps x -> a x + b, one map per lane; (a2, b2) o (a1, b1) = (a2 a1, fma(a2, b1, b2)), out-of-row
// operands the identity (1, 0).  Prefix: lane L ends with f_L o f_(L-1) o ... o f_0; suffix: f_L o f_(L+1) o ... o f_63
// (own map applied last in both).  In-row Hillis-Steele stages 1,2,4,8, then the composed maps of the rows before
// (after) as one carry -- prefix: T0 | T1 o T0 | T2 o (T1 o T0); suffix: T3 | T2 o T3 | T1 o (T2 o T3).
template <int CTRL>
__device__ __forceinline__ void affine_stage(double &a, double &b)
{
    const double ea = dpp_mov_masked<CTRL, 0xF>(1.0, a), eb = dpp_mov<CTRL>(b);
    const double na = a * ea;
    b = __builtin_fma(a, eb, b);
    a = na;
}
__device__ __forceinline__ void affine_prefix_scan(double &a, double &b)
{
    affine_stage<DPP_ROW_SHR(1)>(a, b);
    affine_stage<DPP_ROW_SHR(2)>(a, b);
    affine_stage<DPP_ROW_SHR(4)>(a, b);
    affine_stage<DPP_ROW_SHR(8)>(a, b);
    const double a0 = readlane_f64(a, 15), b0 = readlane_f64(b, 15), a1 = readlane_f64(a, 31), b1 = readlane_f64(b, 31),
                 a2 = readlane_f64(a, 47), b2 = readlane_f64(b, 47);
    const double c2a = a1 * a0, c2b = __builtin_fma(a1, b0, b1);
    const double c3a = a2 * c2a, c3b = __builtin_fma(a2, c2b, b2);
    const int row = ((int)threadIdx.x & (W - 1)) >> 4;
    const double ea = (row == 0) ? 1.0 : (row == 1 ? a0 : (row == 2 ? c2a : c3a));
    const double eb = (row == 0) ? 0.0 : (row == 1 ? b0 : (row == 2 ? c2b : c3b));
    const double na = a * ea;
    b = __builtin_fma(a, eb, b);
    a = na;
}
__device__ __forceinline__ void affine_suffix_scan(double &a, double &b)
{
    affine_stage<DPP_ROW_SHL(1)>(a, b);
    affine_stage<DPP_ROW_SHL(2)>(a, b);
    affine_stage<DPP_ROW_SHL(4)>(a, b);
    affine_stage<DPP_ROW_SHL(8)>(a, b);
    const double a1 = readlane_f64(a, 16), b1 = readlane_f64(b, 16), a2 = readlane_f64(a, 32), b2 = readlane_f64(b, 32),
                 a3 = readlane_f64(a, 48), b3 = readlane_f64(b, 48);
    const double c1a = a2 * a3, c1b = __builtin_fma(a2, b3, b2);
    const double c0a = a1 * c1a, c0b = __builtin_fma(a1, c1b, b1);
    const int row = ((int)threadIdx.x & (W - 1)) >> 4;
    const double ea = (row == 3) ? 1.0 : (row == 2 ? a3 : (row == 1 ? c1a : c0a));
    const double eb = (row == 3) ? 0.0 : (row == 2 ? b3 : (row == 1 ? c1b : c0b));
    const double na = a * ea;
    b = __builtin_fma(a, eb, b);
    a = na;
}

// dot product over the parameter axis: slot 0 product, slot 1 fma'd in, then butterfly
template <int PPL>
__device__ __forceinline__ double pdot_part(const double (&a)[PPL], const double (&b)[PPL])
{
    double part = a[0] * b[0];
    if (PPL == 2) part = __builtin_fma(a[PPL - 1], b[PPL - 1], part);
    return part;
}
template <int PPL>
__device__ __forceinline__ double pdot(const double (&a)[PPL], const double (&b)[PPL])
{
    return bfly_sum(pdot_part<PPL>(a, b));
}
// the dot product in lane 63 of the result (bfly_sum_l63)
template <int PPL>
__device__ __forceinline__ double pdot_l63(const double (&a)[PPL], const double (&b)[PPL])
{
    return bfly_sum_l63(pdot_part<PPL>(a, b));
}

// LDS hand-off between the lanes of ONE wave (multi-wave workgroups must not use s_barrier for
// this): LDS operations of a wave execute in program order, so only the compiler has to be
// kept from moving accesses across the point.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ bool finite_f64(double x)
{
    return (x - x) == 0.0;
}

}  // namespace tsf
