// tsf_quad_kernels.h -- fit kernel for models that are LINEAR in (k, m, delta, beta): linear
// growth with only additive design columns (the BASELINE cfg2 / cfg3 / cfg5 shape), aligned or
// ragged panels.  Same model, same Stan L-BFGS as tsf_fit_kernels.h; what changes is how the
// normal log-likelihood's data term is evaluated.
//
// With mu = Z theta_L (Z = [t, 1, (t - s_j)+ ..., X], shared by every series of an aligned
// panel) and a reference point `ref` with residual r_ref = y - Z ref:
//       SSE(theta) = s0 - 2 c.D + D.(M D),   Z^T r(theta) = c - M D,    D = theta - ref,
//       s0 = |r_ref|^2,  c = Z^T r_ref,  M = Z^T Z  (P x P, built once per grid).
// One evaluation is then a P x P mat-vec out of LDS instead of a pass over T x K design values:
// ~3 k instead of ~40 k fma for cfg2, and no design-matrix traffic at all.  s0 and c come from
// a residual-form pass (the initial point, then again whenever |Z D|^2 > recenter_ratio * s0
// at an accepted iterate or after recenter_every accepted iterates), which keeps the rounding
// error of the quadratic form at the level of the residual form's (measured by
// oracle/prophet_canon.c cn_fit_checked: f agrees to ~1e-14 relative on every evaluation).
//
// Execution model: persistent workgroups of NW waves; on an aligned panel M is shared and lives
// once per workgroup in LDS, on a ragged panel every wave builds the M of its current series
// into its own slot of global memory; every wave pulls series indices from a global atomic
// counter and runs the whole L-BFGS for its series (one wavefront per series, parameter p in
// lane p%64).  The L-BFGS history is held in registers, or -- the shared-M kernel, compiled for three
// waves per SIMD (12 per CU) -- in an LDS ring.  oracle/prophet_canon.c (cn_resid_q / cn_eval_gram / cn_assemble_q / cn_lbfgs)
// performs the identical operation sequence; tests require bit equality.
#pragma once
#include "tsf_fit_kernels.h"

namespace tsf {

// Wave-uniform scalars (every lane holds the same bits) are passed through v_readfirstlane where
// they are produced: the value is unchanged, but the compiler then KNOWS it is uniform -- the
// branches of the line search become scalar branches (no exec masking, no per-lane copies of the
// optimiser state at every join) and the state lives in SGPRs instead of one VGPR pair per scalar.
#define UQ(x) uniform_f64(x)
#ifndef TSF_QUAD_EXPSC
#define TSF_QUAD_EXPSC 0        // exp's literals as scalar operands of VOP3 instructions (dm_exp_sel_sc): see there
#endif

constexpr int QH = 5;                   // L-BFGS history of the register-resident path
// waves per SIMD the kernels are compiled for (register budget 512 / this per lane): 3 for the
// shared-M one-slot kernel (<= 168 VGPRs, 12 waves per CU), else 2.  Every variant with LDS to spare
// keeps its L-BFGS history in an LDS ring (HLDS); the ragged variant with a private Z^T Z per wave in
// LDS has none left and keeps it in registers
constexpr int quad_waves_per_simd(bool w3, int nw = 0) { return nw == 16 ? 4 : (w3 ? 3 : 2); }
// the kernel variant that is: shared M in LDS, one parameter per lane, L-BFGS history in LDS
constexpr bool quad_three_waves(int mmode, int ppl, bool hlds) { return hlds && mmode == 0 && ppl == 1; }

struct QuadArgs {
    FitArgs f;
    const double *Mg;                   // aligned: [P4][PPL][64] Gram matrix, column-major over q
    double *Mslot;                      // ragged: [slots][P4][PPL][64], one per resident wave
    const double *Mpre;                 // ragged, shared grids: [n_pre][P4][PPL][64], one per distinct timestamp vector; or null
    int64_t n_pre;
    double *rbuf;                       // [slots][NTmax][64] residual scratch
    int *counter;                       // work queue head (zeroed before the launch)
    int P4;                             // P rounded up to a multiple of 4
    int recenter_every;
    double recenter_ratio;
    long long *dbg;                     // -DTSF_QUAD_TIMING builds only: [N][8] cycles per phase
    void *nb_buf;                       // slot records of newton_batch_kernel (tsf_newton_batch.h), or null
    size_t nb_bytes;
    int gram_harm;                      // ragged, a grid per series, M in registers: harm_code of the model when the build expands the
                                        // Fourier columns from the rows' base pairs (FitArgs::Bw; gram_columns_harm), else 0
};

template <int KP, int PPL>
struct QuadLds {
    double th[PPL * W + W];             // theta of the running residual pass; zero beyond P
    double ks[NTAB + 1], mc[NTAB + 1];
    double tp1[NTAB], tp2[NTAB];
    double tot1[W + 1], tot2[W + 1];
    double accR[KP];
    // per-series vectors that are cold during an evaluation live here, not in registers
    double ref[PPL * W], cvec[PPL * W];                // reference point and c = Z^T r_ref
};

// cn_assemble_q lane constants (lc, sc, qc: [3][PPL][64] doubles).  They depend on the model and
// on the number of changepoints only: one copy per workgroup on an aligned panel (every wave
// writes the same bits), one per wave on a ragged one.
// ... followed (fit_quad_kernel / eval_quad_kernel) by a table of QC_N wave-uniform constants, read with
// broadcast LDS loads where the code needs them.  Round 4: the fp64 literals of exp (13 coefficients: two
// v_mov_b32 each, 26 vector instructions per evaluation) and the optimiser's tolerances (kernel arguments the
// compiler kept in scalar registers spilled into VGPR lanes: ~22 v_readlane per iteration to get six doubles
// back) cost vector-issue slots, which is what this kernel is bound by; an LDS read of a uniform address costs
// none.  Same values, same operations (the oracle is untouched).
constexpr int QC_N = 32;
enum { QC_LOG2E = 0, QC_LN2HI = 1, QC_LN2LO = 2, QC_EXP_P = 3 /* 11 coefficients */, QC_EXP_HI = 14, QC_EXP_LO = 15,
       QC_TOL_OBJ = 16, QC_TOL_REL_OBJ = 17, QC_TOL_GRAD = 18, QC_TOL_REL_GRAD = 19, QC_TOL_PARAM = 20,
       QC_INIT_ALPHA = 21, QC_RC_RATIO = 22,
       QC_GRAD2_LO = 23, QC_GRAD2_HI = 24, QC_PARAM2_LO = 25, QC_PARAM2_HI = 26, QC_RELGRAD_LO = 27, QC_RELGRAD_HI = 28 };
template <int PPL>
constexpr size_t quad_lanec_bytes() { return sizeof(double) * (3 * PPL * W + QC_N); }

// a wave-uniform constant from the LDS table.  volatile: the load stays where it is written -- hoisted out of the
// optimiser's loops each constant would occupy a VGPR pair for the whole fit, which is what the table exists to
// avoid.  (Tried instead: ordinary loads through a pointer re-derived once per iteration behind an empty asm
// statement -- the 12-wave kernel then spilled 49 registers per lane to scratch; with volatile loads none.)
// (The cast names the address space: a volatile access through a generic pointer is not rewritten to LDS by the
// compiler's address-space inference and would become a flat load.)
typedef const volatile __attribute__((address_space(3))) double *qc_lds_ptr;
__device__ __forceinline__ double qc(const double *ct, int k) { return *((qc_lds_ptr)ct + k); }

template <int PPL>
struct QuadWave { double dl[PPL * W], ref[PPL * W], cvec[PPL * W]; };

struct QuadPool {
    int *locks;                         // [ns] in LDS, 0 = free
    unsigned char *slots;               // [ns] x slot_bytes in LDS: one QuadLds each, + NTmax x 64 staging rows where they fit
    int ns, first;                      // first: where this wave starts looking (spreads the waves)
    unsigned slot_bytes;
};
constexpr size_t QUAD_POOL_LOCK_BYTES = 64;

__device__ __forceinline__ int pool_acquire(const QuadPool &pl)
{
    int i = pl.first;
    for (;;) {
        const int old = atomicCAS(pl.locks + i, 0, 1);
        if (__any(old == 0)) break;
        i = (i + 1 == pl.ns) ? 0 : i + 1;
        if (i == pl.first) __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return __builtin_amdgcn_readfirstlane(i);
}

__device__ __forceinline__ void pool_release(const QuadPool &pl, int i)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_store(pl.locks + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}


// NTR > 0: the weights of steps q < NTR are in the register array rr[NTR], those of later steps (series of more
// than 64 NTR rows) in the staging rows rb
template <int KP, int G, int NTR = 0>
__device__ __forceinline__ void column_group(const SeriesView &sv, const double *rb, int g0,
                                             double *accR, const double (&rr)[NTR > 0 ? NTR : 1])
{
    const int lane = lane_id();
    double acc[G];
#pragma unroll
    for (int j = 0; j < G; ++j) acc[j] = 0.0;
    // branch-free over the NT steps: rows past the end of the series hold r = 0 (written by
    // ztr_pass) and X = 0 (zero-filled padding), and fma(0, 0, acc) leaves acc unchanged
    auto step = [&](int q, double r) {
        const double *xp = sv.Xw + ((size_t)q * KP + g0) * W + lane;
        double x[G];
#pragma unroll
        for (int j = 0; j < G; ++j) x[j] = xp[j * W];
#pragma unroll
        for (int j = 0; j < G; ++j) acc[j] = __builtin_fma(x[j], r, acc[j]);
    };
    if (NTR > 0) {
        // (two loops, q descending throughout: a select between the register array and memory inside ONE
        // loop sends the array to scratch)
#pragma unroll 2
        for (int q = sv.NT - 1; q >= NTR; --q) step(q, rb[q * W + lane]);
        const int q1 = sv.NT < NTR ? sv.NT : NTR;
#pragma unroll 4
        for (int q = q1 - 1; q >= 0; --q) step(q, rr[NTR > 0 ? q : 0]);
    } else {
#pragma unroll 4
        for (int q = sv.NT - 1; q >= 0; --q) step(q, rb[q * W + lane]);
    }
    column_sums_g<G>(acc, accR + g0);
}

// column_group for a series whose rows are kept as BASE PAIRS (SeriesView::Bw; a ragged panel with a calendar per
// series, round 5): the Fourier columns G0 .. G0 + G - 1 of a row expanded from its pairs (harm_row: the recurrence runs
// up to the group's last harmonic, the rest is dead code), dense columns behind the Fourier block from Xw where the
// model has any.  Same chains as column_group: the values are the bits Xw holds.
template <int KP, int HARM, int G0, int G>
__device__ __forceinline__ void column_group_harm(const SeriesView &sv, const double *rb, double *accR, bool has_xd)
{
    constexpr int KF = harm_kf(HARM), NS = harm_ns(HARM);
    const int lane = lane_id();
    double acc[G];
#pragma unroll
    for (int j = 0; j < G; ++j) acc[j] = 0.0;
#pragma unroll 2
    for (int q = sv.NT - 1; q >= 0; --q) {
        const double r = rb[q * W + lane];
        const bool valid = q < sv.cnt;
        const double2 *bq = reinterpret_cast<const double2 *>(sv.Bw) + (size_t)q * NS * W + lane;
        double2 bp[NS];
#pragma unroll
        for (int se = 0; se < NS; ++se) {
            // (rows a chunk does not have were never written in Bw: any finite pair, their weight is 0)
            const double2 b = bq[se * W];
            bp[se].x = valid ? b.x : 0.0; bp[se].y = valid ? b.y : 0.0;
        }
        harm_row<HARM>(bp, [&](int j, double v) {
            if (j >= G0 && j < G0 + G) acc[j - G0 < 0 ? 0 : (j - G0 < G ? j - G0 : 0)] = __builtin_fma(v, r, acc[j - G0 < 0 ? 0 : (j - G0 < G ? j - G0 : 0)]);
        });
        if (G0 + G > KF && has_xd) {
#pragma unroll
            for (int j = (G0 > KF ? G0 : KF); j < G0 + G; ++j)
                acc[j - G0] = __builtin_fma(sv.Xw[((size_t)q * KP + j) * W + lane], r, acc[j - G0]);
        }
    }
    column_sums_g<G>(acc, accR + G0);
}

// Z^T r and r.r, in the operation order of eval_fg<GROWTH 0, MODE 0> (cn_ztr).  The weight of
// row lane*NT+q comes from gen(q, idx, c, ti) (called for valid rows only, q descending); it is
// parked in rb[q*64+lane] for the per-column passes (0 for rows past the end of the series).
// ztr[s]: entry p = lane + 64 s.
// NTR > 0: the weights of the first NTR steps stay in NTR registers of the lane that made them (later steps, if any:
// staging rows) -- row lane*NT+q is
// produced and consumed by the same lane, so the staging rows are nothing but spill space: 12 doubles per
// lane for cfg2, which the 168-register budget of the 12-waves-per-CU kernel has room for (138 used).  Round 2
// staged them through global memory (the LDS is full at 12 waves): 7 x the algorithmic HBM bytes.  The array is
// indexed by the wave-uniform step q (s_set_gpr_idx / v_movrel, no scratch).
template <int KP, int PPL, int NTR = 0, int HARM = 0, class RGen>
__device__ __forceinline__ void ztr_pass(const SeriesView &sv, QuadLds<KP, PPL> &wl, double *rb,
                                         RGen gen, double &sse_out, double (&ztr)[PPL])
{
    const int lane = lane_id();
    const int S = sv.S;
    double sse = 0.0, rt1 = 0.0, rt2 = 0.0;
    double rr[NTR > 0 ? NTR : 1];
    if (NTR > 0) {
#pragma unroll
        for (int q = 0; q < (NTR > 0 ? NTR : 1); ++q) rr[q] = 0.0;
    }
    auto row_step = [&](int q) -> double {
        const bool valid = q < sv.cnt;
        const int idx = q * W + lane;
        const unsigned cwv = valid ? (unsigned)sv.cw[idx] : 0u;
        const int c = (int)(cwv & 0xffu), cprev = (int)(cwv >> 8);
        const double ti = valid ? sv.tw[idx] : 0.0;
        double r = gen(q, idx, c, ti);
        if (!valid) r = 0.0;
        sse = __builtin_fma(r, r, sse);
        rt1 = __builtin_fma(r, ti, rt1);
        rt2 = rt2 + r;
        for (int j = cprev; j < c; ++j) { wl.tp1[j] = rt1; wl.tp2[j] = rt2; }
        return r;
    };
    if (NTR > 0) {
#pragma unroll 1
        for (int q = sv.NT - 1; q >= NTR; --q) rb[q * W + lane] = row_step(q);
        const int q1 = sv.NT < NTR ? sv.NT : NTR;
#pragma unroll 1
        for (int q = q1 - 1; q >= 0; --q) rr[NTR > 0 ? q : 0] = row_step(q);
    } else {
#pragma unroll 1
        for (int q = sv.NT - 1; q >= 0; --q) rb[q * W + lane] = row_step(q);
    }
    sse_out = bfly_sum(sse);
    const double s1 = suffix_scan(rt1), s2v = suffix_scan(rt2);
    wl.tot1[lane] = s1; wl.tot2[lane] = s2v;
    if (lane == 0) { wl.tot1[W] = 0.0; wl.tot2[W] = 0.0; }
    constexpr int G8 = (KP / 8) * 8;
    if constexpr (HARM != 0) {
        static_assert(HARM == 0 || (KP == 28 && NTR == 0), "base-pair rows: the 28-column kernels, weights in the staging rows");
        const bool has_xd = sv.P - 3 - S > harm_kf(HARM);
        column_group_harm<KP, HARM, 0, 8>(sv, rb, wl.accR, has_xd);
        column_group_harm<KP, HARM, 8, 8>(sv, rb, wl.accR, has_xd);
        column_group_harm<KP, HARM, 16, 8>(sv, rb, wl.accR, has_xd);
        column_group_harm<KP, HARM, 24, 4>(sv, rb, wl.accR, has_xd);
    } else {
#pragma unroll 1
    for (int g0 = 0; g0 < G8; g0 += 8) column_group<KP, 8, NTR>(sv, rb, g0, wl.accR, rr);
    if (KP % 8 != 0) column_group<KP, 4, NTR>(sv, rb, G8, wl.accR, rr);
    }
    wave_sync();
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const int p = lane + s * W;
        double v = 0.0;
        if (p == 0) v = wl.tot1[0];
        else if (p == 1) v = wl.tot2[0];
        else if (p >= 3 && p < 3 + S) {
            const int j = p - 3, Lj = sv.Ljp_l[s];
            const double SA = wl.tp1[j] + wl.tot1[Lj + 1];
            const double SB = wl.tp2[j] + wl.tot2[Lj + 1];
            v = SA - sv.tcp_l[s] * SB;
        } else if (p >= 3 + S && p < sv.P) {
            v = wl.accR[p - 3 - S];
        }
        ztr[s] = v;
    }
    wave_sync();
}

// per-lane constants of cn_assemble_q
template <int PPL>
struct LaneConst {
    const double *lc, *sc, *qc;         // in LDS, entry p = lane + 64 s at [s * 64 + lane]
    double inv_tau;
    const double *ct;                   // table of wave-uniform constants (QC_*), or null (Newton kernels)
};

// the math constants of the table (dm_exp_sel's, in its order)
__device__ const double QC_MATH[16] = {
    1.4426950408889634, 6.93147180369123816490e-01, 1.90821492927058770002e-10,
    1.6059043836821613e-10, 2.08767569878681e-09, 2.505210838544172e-08, 2.755731922398589e-07,
    2.7557319223985893e-06, 2.48015873015873e-05, 1.984126984126984e-04, 1.388888888888889e-03,
    8.333333333333333e-03, 4.1666666666666664e-02, 1.6666666666666666e-01,
    709.782712893384, -745.2};

// fills the table behind the lane constants (lane k writes entry k); opt: the call's optimiser settings
__device__ __forceinline__ void quad_const_table(double *ct, const LbfgsOpts &opt, double recenter_ratio)
{
    const int l = lane_id();
    double v = 0.0;
    if (l < 16) v = QC_MATH[l];
    else if (l == QC_TOL_OBJ) v = opt.tol_obj;
    else if (l == QC_TOL_REL_OBJ) v = opt.tol_rel_obj_eps;
    else if (l == QC_TOL_GRAD) v = opt.tol_grad;
    else if (l == QC_TOL_REL_GRAD) v = opt.tol_rel_grad_eps;
    else if (l == QC_TOL_PARAM) v = opt.tol_param;
    else if (l == QC_INIT_ALPHA) v = opt.init_alpha;
    else if (l == QC_RC_RATIO) v = recenter_ratio;
    // brackets of the termination tests that compare a NORM or a QUOTIENT with a tolerance (fit_one_quad): where the
    // operand is clearly on one side the square root / the division is not taken.  (1 +- 4 eps) covers every
    // rounding on the way; a tolerance whose square leaves the normal range, or that is not positive and finite,
    // gets the bracket (0, inf) / (-inf, inf): every test then takes the exact path.
    else if (l >= QC_GRAD2_LO && l <= QC_PARAM2_HI) {
        const double tol = (l <= QC_GRAD2_HI) ? opt.tol_grad : opt.tol_param;
        const double t2 = tol * tol;
        const bool ok = tol > 0.0 && t2 >= 1e-290 && t2 <= 1e290;
        const bool lo = (l == QC_GRAD2_LO || l == QC_PARAM2_LO);
        v = ok ? t2 * (lo ? 1.0 - 8.881784197001252e-16 : 1.0 + 8.881784197001252e-16) : (lo ? 0.0 : __builtin_huge_val());
    } else if (l == QC_RELGRAD_LO || l == QC_RELGRAD_HI) {
        const double t = opt.tol_rel_grad_eps;
        const bool ok = t >= 1e-290 && t <= 1e290;
        v = (l == QC_RELGRAD_LO) ? (ok ? t * (1.0 - 8.881784197001252e-16) : -__builtin_huge_val())
                                 : (ok ? t * (1.0 + 8.881784197001252e-16) : __builtin_huge_val());
    }
    if (l < QC_N) ct[l] = v;
}

// fp64 arithmetic with a LITERAL operand held in a scalar register pair.  A VOP3 instruction of gfx950 cannot encode
// a 64-bit literal, and left to itself the compiler materialises each one in a VGPR pair (two v_mov_b32) to use the
// two-address v_fmac form: 26 vector instructions per evaluation for exp's 13 coefficients, in a kernel whose large
// launches are bound by vector issue (SQ_ACTIVE_INST_VALU 91 % at 16 waves per CU: profiles/r04_pmc_wide).  Written
// as the VOP3 instruction itself, the constant is two s_mov_b32 on the scalar unit, which has the room (26 % busy).
// Same instruction, same rounding: only the operand's register file differs.
// MEASURED (round 4, static: tools/dev/isa_mix.sh) and NOT enabled: the asm statements pin the schedule of the
// evaluation, the 12- and 16-wave kernels then spill ~49 registers per lane to scratch and issue MORE vector
// instructions per evaluation (371 against 339 in the trial loop), not fewer.  Kept behind TSF_QUAD_EXPSC.
// (fma_vvs / fma_vsv / mul_vs and dm_exp_sel_sc: tsf_detmath.h)

// dm_exp_sel (tsf_detmath.h) with its constants read from the table: the same operations on the same values
__device__ __forceinline__ double dm_exp_sel_tab(double x, const double *ct)
{
    const double n = __builtin_rint(x * qc(ct, QC_LOG2E));
    double r = __builtin_fma(-n, qc(ct, QC_LN2HI), x);
    r = __builtin_fma(-n, qc(ct, QC_LN2LO), r);
    double p = qc(ct, QC_EXP_P);
#pragma unroll
    for (int k = 1; k < 11; ++k) p = __builtin_fma(p, r, qc(ct, QC_EXP_P + k));
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double hi = qc(ct, QC_EXP_HI), lo = qc(ct, QC_EXP_LO);
    const bool in_range = (x <= hi) && (x >= lo);
    const int ni = in_range ? (int)n : 0;
    const int n1 = ni / 2, n2 = ni - n1;
    double e = (p * dm_pow2i(n1)) * dm_pow2i(n2);
    if (x > hi) e = __builtin_huge_val();
    if (x < lo) e = 0.0;
    if (x != x) e = x;
    return e;
}

template <int PPL>
__device__ __forceinline__ void lane_consts(const DevSpec *sp, const SeriesView &sv,
                                            double *dst, LaneConst<PPL> &k)
{
    const double C25 = 1.0 / 25.0;
    k.ct = nullptr;
    k.inv_tau = 1.0 / sv.tau;
    double *lcp = dst, *scp = dst + PPL * W, *qcp = dst + 2 * PPL * W;
    k.lc = lcp; k.sc = scp; k.qc = qcp;
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const int p = lane_id() + s * W;
        double lc = 0.0, sc = 0.0, qc = 0.0;
        if (p < 2) lc = C25;
        else if (p >= 3 && p < 3 + sv.S) sc = k.inv_tau;
        else if (p >= 3 + sv.S && p < sv.P) {
            const double pr = sp->prior[p - 3 - sv.S];
            lc = 1.0 / (pr * pr);
            qc = 1.0 / pr;
        }
        lcp[p] = lc; scp[p] = sc; qcp[p] = qc;
    }
    wave_sync();
}

// f and gradient from (SSE, Z^T r): cn_assemble_q, in two parts -- everything that does not depend
// on (SSE, Z^T r) first, so that the exp / division / prior-sum chains can run underneath the
// mat-vec of gram_eval_q; the floating-point operations and their order are those of cn_assemble_q
template <int PPL>
struct AsmPre {
    double inv_s2, s2, f0, tls, pa, pb;         // pa, pb: per-lane partials of the two prior sums
    double lc[PPL], sc[PPL];
};

template <int PPL, bool CTAB = false>
__device__ __forceinline__ void assemble_pre(const SeriesView &sv, const LaneConst<PPL> &lk,
                                             const double (&th)[PPL], AsmPre<PPL> &ap)
{
    const int lane = lane_id();
    const double k = readlane_f64(th[0], 0), m = readlane_f64(th[0], 1), ls = readlane_f64(th[0], 2);
    const double C25 = 1.0 / 25.0;
    const double sigma = CTAB ? dm_exp_sel_tab(ls, lk.ct) : (TSF_QUAD_EXPSC ? dm_exp_sel_sc(ls) : dm_exp_sel(ls));
    const double s2 = sigma * sigma;
    ap.inv_s2 = 1.0 / s2;
    ap.s2 = s2;
    ap.tls = (double)sv.T * ls;
    double pa = 0.0, pb = 0.0;
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        ap.lc[s] = lk.lc[s * W + lane]; ap.sc[s] = lk.sc[s * W + lane];
        if (ap.sc[s] != 0.0) pa = pa + __builtin_fabs(th[s]);          // delta lanes
        const double qq = th[s] * lk.qc[s * W + lane];
        pb = __builtin_fma(qq, qq, pb);
    }
    ap.pa = pa; ap.pb = pb;
    ap.f0 = ((0.5 * k) * k) * C25 + ((0.5 * m) * m) * C25;
}

// sabs, sb: the butterfly sums of ap.pa, ap.pb
template <int PPL>
__device__ __forceinline__ bool assemble_post(const SeriesView &sv, const LaneConst<PPL> &lk,
                                              const AsmPre<PPL> &ap, double sabs, double sb,
                                              const double (&th)[PPL], double sse,
                                              const double (&ztr)[PPL], double &f_out,
                                              double (&g)[PPL])
{
    const int lane = lane_id();
    double f = ap.f0;
    f = f + sabs * lk.inv_tau;
    f = f + 2.0 * ap.s2;
    f = f + 0.5 * sb;
    f = f + ap.tls;
    f = f + (0.5 * sse) * ap.inv_s2;
    const double nis = -ap.inv_s2;
    const double g2 = ((double)sv.T - sse * ap.inv_s2) + 4.0 * ap.s2;
    bool bad = false;
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const double sgn = (double)((th[s] > 0.0) - (th[s] < 0.0));
        double gv = __builtin_fma(th[s], ap.lc[s], nis * ztr[s]) + sgn * ap.sc[s];
        if (s == 0 && lane == 2) gv = g2;
        if (lane + s * W >= sv.P) gv = 0.0;
        g[s] = gv;
        bad = bad || !finite_f64(gv);
    }
    f_out = f;
    return !finite_f64(f) || __any(bad);
}

template <int PPL>
__device__ __forceinline__ bool assemble_q(const SeriesView &sv, const LaneConst<PPL> &lk,
                                           const double (&th)[PPL], double sse,
                                           const double (&ztr)[PPL], double &f_out,
                                           double (&g)[PPL])
{
    AsmPre<PPL> ap;
    assemble_pre<PPL>(sv, lk, th, ap);
    const double sabs = bfly_sum(ap.pa), sb = bfly_sum(ap.pb);
    return assemble_post<PPL>(sv, lk, ap, sabs, sb, th, sse, ztr, f_out, g);
}

// residual-form evaluation (cn_resid_q): r -> rb, then Z^T r, then assemble_q
template <int KP, int PPL, int NTR = 0, int HARM = 0>
__device__ __forceinline__ bool resid_eval_q(const SeriesView &sv, QuadLds<KP, PPL> &wl,
                                             const LaneConst<PPL> &lk, double *rb,
                                             const double (&th)[PPL], double &f_out,
                                             double (&g)[PPL], double &sse_out,
                                             double (&ztr)[PPL])
{
    const int lane = lane_id();
    const int S = sv.S;
#pragma unroll
    for (int s = 0; s < PPL; ++s) wl.th[lane + s * W] = th[s];
    {   // ks[c], mc[c]: lane c accumulates the first c terms in sequential order (see eval_fg)
        double ksv = readlane_f64(th[0], 0), mcv = readlane_f64(th[0], 1);
        const double tcl = sv.tc_l;
        for (int j = 0; j < S; ++j) {
            const double dj = readlane_f64(th[0], 3 + j);
            const double ksn = ksv + dj;
            const double mcn = mcv + ((-readlane_f64(tcl, j)) * dj);
            if (j < lane) { ksv = ksn; mcv = mcn; }
        }
        if (lane <= S) { wl.ks[lane] = ksv; wl.mc[lane] = mcv; }
    }
    wave_sync();
    const double *beta = wl.th + 3 + S;
    auto gen = [&](int q, int idx, int c, double ti) -> double {
        // y of the row: from the scaled step-major copy, or (round 6) the caller's own row, scaled here with
        // setup_series_kernel's operation -- (y - 0) / y_scale: linear growth has no floor -- so the same bits, and no
        // second f64 panel on the device (HBM bytes of the step: 4.1 x the algorithmic figure -> 2.2 x)
        double yi;
        if (sv.y_raw) {
            yi = 0.0;
            if (q < sv.cnt) yi = load_y(sv.y_raw, sv.y_dtype, sv.y_base + (long long)lane * sv.NT + q) / sv.y_scl;
        } else {
            yi = sv.yw[idx];
        }
        const double *xp = sv.Xw + (size_t)q * KP * W + lane;
        double xa = 0.0;
        if constexpr (HARM != 0) {
            // (round 5) the row as its base pairs, the Fourier columns expanded (harm_row); a row the chunk does not have
            // reads whatever Bw holds there: ztr_pass drops its r
            constexpr int KF = harm_kf(HARM), NS = harm_ns(HARM);
            const double2 *bq = reinterpret_cast<const double2 *>(sv.Bw) + (size_t)q * NS * W + lane;
            double2 bp[NS];
#pragma unroll
            for (int se = 0; se < NS; ++se) bp[se] = bq[se * W];
            harm_row<HARM>(bp, [&](int j, double v) { xa = __builtin_fma(v, beta[j], xa); });
            if (sv.P - 3 - S > KF) {
#pragma unroll
                for (int j = KF; j < KP; ++j) xa = __builtin_fma(xp[j * W], beta[j], xa);
            }
        } else {
            double x[KP];
#pragma unroll
            for (int j = 0; j < KP; ++j) x[j] = xp[j * W];
#pragma unroll
            for (int j = 0; j < KP; ++j) xa = __builtin_fma(x[j], beta[j], xa);
        }
        const double gtr = __builtin_fma(wl.ks[c], ti, wl.mc[c]);
        return yi - (gtr + xa);
    };
    ztr_pass<KP, PPL, NTR, HARM>(sv, wl, rb, gen, sse_out, ztr);
    return assemble_q<PPL>(sv, lk, th, sse_out, ztr, f_out, g);
}

// quadratic-form evaluation (cn_eval_gram).  Ml: [P4][PPL][64] in LDS (or global).
// PQ > 0: compile-time row count of M (PPL == 1; rows >= P are zero, which is bit-neutral), the
// whole evaluation is then ONE basic block and the scheduler can run the exp / division /
// prior-sum chains of assemble_q underneath the LDS reads and the four fma chains.
// MRS: doubles between consecutive rows of Ml (64, or PQ for the compact per-wave copies of ragged
// panels: lanes >= MRS then read finite entries of the next row, which only ever meet D = 0)
// MREG: row p of M (= column p: M is symmetric) in PQ registers of lane p instead of LDS / global
// memory -- the per-series Z^T Z of a ragged panel: 25 KB of LDS per wave allowed 4 waves per CU, 112
// registers per lane allow 8, and the mat-vec loses its 56 LDS reads of M.  Same operands, same order.
#ifndef TSF_QUAD_MPIPE
#define TSF_QUAD_MPIPE 1
#endif
// exp's fp64 literals from the LDS table (dm_exp_sel_tab) instead of two v_mov_b32 each.  Measured (round 4,
// profiles/r04_quad/phase_cycles.txt): 26 vector instructions fewer per evaluation and 1 000 cycles MORE -- each
// Horner step then waits for its own LDS round trip (the loads are volatile and stay at their use), and a wave of
// this kernel is bound by its dependent chains, not by vector issue.  Off; the table keeps the tolerances.
#ifndef TSF_QUAD_EXPTAB
#define TSF_QUAD_EXPTAB 0
#endif
template <int PPL, int PQ, int MRS = W, int MB_ = 16, bool MREG = false, bool MPIPE = (TSF_QUAD_MPIPE != 0), bool CTAB = false>
__device__ __forceinline__ bool gram_eval_q(const SeriesView &sv, const LaneConst<PPL> &lk,
                                            const double *Ml, int P4, const double (&th)[PPL],
                                            const double *ref_l, const double *cvec_l,
                                            double s0, double &f_out, double (&g)[PPL],
                                            double &q2_out, double *dl,
                                            const double (&mreg)[(MREG && PQ > 0) ? PQ : 1],
                                            int mstride = W)     // generic path (PQ == 0), PPL == 1: doubles between rows of Ml
{
    double ref[PPL], cvec[PPL];
#pragma unroll
    for (int s = 0; s < PPL; ++s) { ref[s] = ref_l[s * W + lane_id()]; cvec[s] = cvec_l[s * W + lane_id()]; }
    const int lane = lane_id();
    double D[PPL], v[PPL], a[PPL][4];
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const int p = lane + s * W;
        D[s] = (p == 2 || p >= sv.P) ? 0.0 : th[s] - ref[s];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[s][u] = 0.0;
    }
    if (PQ > 0) { dl[lane] = D[0]; wave_sync(); }
    AsmPre<PPL> ap;
    assemble_pre<PPL, CTAB>(sv, lk, th, ap);
    const double *mp = Ml + lane;
    if (PQ > 0) {
        static_assert(PQ == 0 || PPL == 1, "compile-time M rows only for P <= 64");
        // batches of MB rows (16, or 8 in the three-waves-per-SIMD kernel): MB LDS reads in flight, then their fmas; the scheduling barrier
        // keeps the compiler from hoisting every read of M to the top (register pressure)
        constexpr int MB = MB_;
        if constexpr (!MREG && MPIPE) {
            // Software-pipelined: the reads of batch b + 1 are issued BEFORE the fmas of batch b, so only the first
            // batch pays the LDS latency (one batch at a time cost ~230 cycles each, seven of them per evaluation:
            // 1.0 k of the 2.6 k cycles of an evaluation; with M in registers the whole evaluation takes 1.7 k).
            // Same operands, same chains (q & 3), same order.
            constexpr int NB = (PQ + MB - 1) / MB;
            double m[2][MB], dq[2][MB];
#pragma unroll
            for (int u = 0; u < MB; ++u) if (u < PQ) { m[0][u] = mp[u * MRS]; dq[0][u] = dl[u]; }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int q1 = (b + 1) * MB;
#pragma unroll
                for (int u = 0; u < MB; ++u) if (b + 1 < NB && q1 + u < PQ) { m[(b + 1) & 1][u] = mp[(q1 + u) * MRS]; dq[(b + 1) & 1][u] = dl[q1 + u]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < MB; ++u) {
                    const int q = b * MB + u;
                    if (q < PQ) a[0][q & 3] = __builtin_fma(m[b & 1][u], dq[b & 1][u], a[0][q & 3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int q0 = 0; q0 < PQ; q0 += MB) {
            double m[MB];
#pragma unroll
            for (int u = 0; u < MB; ++u) if (q0 + u < PQ) m[u] = MREG ? mreg[(MREG && q0 + u < PQ) ? q0 + u : 0] : mp[(q0 + u) * MRS];
            double dq[MB];              // D_q for the whole wave: one broadcast LDS read each
#pragma unroll
            for (int u = 0; u < MB; ++u) if (q0 + u < PQ) dq[u] = dl[q0 + u];
#pragma unroll
            for (int u = 0; u < MB; ++u) {
                if (q0 + u < PQ) a[0][(q0 + u) & 3] = __builtin_fma(m[u], dq[u], a[0][(q0 + u) & 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
    } else {
    const int q_lo = P4 < W ? P4 : W;
    // P4 is a multiple of 4 (rows >= P are zero); the q & 3 chain assignment is kept while the
    // loop is unrolled by 8 so that the LDS reads of a batch are issued ahead of the fmas
    int q = 0;
    for (; q + 8 <= q_lo; q += 8) {
        double m[8][PPL];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int s = 0; s < PPL; ++s) m[u][s] = mp[((q + u) * PPL + s) * (PPL == 1 ? mstride : W)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double Dq = readlane_f64(D[0], q + u);
#pragma unroll
            for (int s = 0; s < PPL; ++s) a[s][u & 3] = __builtin_fma(m[u][s], Dq, a[s][u & 3]);
        }
    }
    for (; q < q_lo; q += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double Dq = readlane_f64(D[0], q + u);
#pragma unroll
            for (int s = 0; s < PPL; ++s)
                a[s][u] = __builtin_fma(mp[((q + u) * PPL + s) * (PPL == 1 ? mstride : W)], Dq, a[s][u]);
        }
    }
    if (PPL == 2) {
        for (q = W; q < P4; q += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double Dq = readlane_f64(D[PPL - 1], q + u - W);
#pragma unroll
                for (int s = 0; s < PPL; ++s)
                    a[s][u] = __builtin_fma(mp[((q + u) * PPL + s) * W], Dq, a[s][u]);
            }
        }
    }
    }
#pragma unroll
    for (int s = 0; s < PPL; ++s) v[s] = (a[s][0] + a[s][1]) + (a[s][2] + a[s][3]);
    // D.(M D), c.D and the two prior sums: one four-fold butterfly (bit-identical to four)
    double q2, cd, sabs, sb;
    bfly_sum4(pdot_part<PPL>(D, v), pdot_part<PPL>(cvec, D), ap.pa, ap.pb, q2, cd, sabs, sb);
    const double sse = __builtin_fma(-2.0, cd, s0) + q2;
    q2_out = q2;
    double ztr[PPL];
#pragma unroll
    for (int s = 0; s < PPL; ++s) ztr[s] = cvec[s] - v[s];
    return assemble_post<PPL>(sv, lk, ap, sabs, sb, th, sse, ztr, f_out, g);
}

// lane-id based variants of make_view / store_theta for multi-wave workgroups
template <int KP, int PPL>
__device__ __forceinline__ void make_view_grid(const FitArgs &a, int64_t g, int64_t n, SeriesView &sv)
{
    const GridTab &gt = a.gtab[g];
    sv.T = gt.info.T; sv.NT = gt.info.NT; sv.S = gt.S_fit; sv.S_out = gt.info.S;
    sv.P = 3 + sv.S + a.sp->K;
    int cnt = sv.T - lane_id() * sv.NT;
    cnt = cnt < 0 ? 0 : (cnt > sv.NT ? sv.NT : cnt);
    sv.cnt = cnt;
    sv.tw = a.tw + (size_t)g * a.NTmax * W;
    sv.cw = a.cw + (size_t)g * a.NTmax * W;
    sv.Xw = a.Xw + (size_t)g * a.NTmax * KP * W;
    sv.Bw = a.Bw ? a.Bw + (size_t)g * a.NTmax * a.bw_ns * 2 * W : nullptr;
    sv.yw = a.yw ? a.yw + (size_t)n * a.NTmax * W : nullptr;
    sv.y_raw = a.y_raw; sv.y_dtype = a.y_raw_dtype;
    sv.y_base = a.y_raw ? (a.y_offsets ? (long long)a.y_offsets[n] : (long long)n * a.y_T) : 0;
    sv.y_scl = 1.0;                      // (set by the caller from series_tab_wave's result)
    sv.Lj = gt.Lj;
    sv.t_change = gt.info.t_change;
    sv.cap = 0.0;
    sv.tau = a.sp->tau;
    sv.n_eval = 0;
    set_lane_tables<PPL>(a.sp, sv);
}

template <int KP, int PPL>
__device__ __forceinline__ void make_view_q(const FitArgs &a, int64_t n, SeriesView &sv)
{
    // ragged panels: one grid per series, or per distinct timestamp vector
    make_view_grid<KP, PPL>(a, grid_index(a, n), n, sv);
}

// ---------------------------------------------------------------------------------------
// Gram matrix: column q of M = Z^T Z is "Z^T r" of the residual machinery with r := column q
// of Z (cn_build_gram).  Aligned panels: gram_build_kernel, one workgroup per column, once per
// grid.  Ragged panels: every wave builds the M of its current series itself (fit_one_quad).
// ---------------------------------------------------------------------------------------
template <int KP, int PPL>
__device__ __forceinline__ void gram_column(const SeriesView &sv, QuadLds<KP, PPL> &wl, double *rb,
                                            int q, double (&ztr)[PPL])
{
    const int lane = lane_id();
    auto gen = [&](int st, int idx, int c, double ti) -> double {
        if (q == 0) return ti;
        if (q == 1) return 1.0;
        if (q < 3 + sv.S) return (c > q - 3) ? ti - sv.t_change[q - 3] : 0.0;
        return sv.Xw[((size_t)st * KP + (q - 3 - sv.S)) * W + lane];
    };
    double sse;
    ztr_pass<KP, PPL>(sv, wl, rb, gen, sse, ztr);
}

// Two columns of M in ONE pass over the rows (ragged panels: every wave builds the Z^T Z of its own
// series, and with the matrix in registers the build was HALF of a fit: 54 passes, each streaming the
// whole design matrix and staging its weights).  The weights of both columns are generated per row from
// the row's own values (no staging), every design value of the row is loaded once and feeds both
// columns' 28 accumulators.  Entry by entry the chains are those of ztr_pass / column_group: same
// operands, same order, same bits.  gx: the second column's trend tables.
struct GramX { double tp1[NTAB], tp2[NTAB], tot1[W + 1], tot2[W + 1], accR[32]; };

template <int KP>
__device__ __forceinline__ void gram_columns2(const SeriesView &sv, QuadLds<KP, 1> &wl, GramX &gx,
                                              int qa_, int qb_, double &za, double &zb)
{
    const int lane = lane_id();
    const int S = sv.S;
    auto zval = [&](int qq, int st, int c, double ti) -> double {
        if (qq == 0) return ti;
        if (qq == 1) return 1.0;
        if (qq < 3 + S) return (c > qq - 3) ? ti - sv.t_change[qq - 3] : 0.0;
        return sv.Xw[((size_t)st * KP + (qq - 3 - S)) * W + lane];
    };
    double acca[KP], accb[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) { acca[j] = 0.0; accb[j] = 0.0; }
    double rt1a = 0.0, rt2a = 0.0, rt1b = 0.0, rt2b = 0.0;
#pragma unroll 1
    for (int q = sv.NT - 1; q >= 0; --q) {
        const bool valid = q < sv.cnt;
        const int idx = q * W + lane;
        const unsigned cwv = valid ? (unsigned)sv.cw[idx] : 0u;
        const int c = (int)(cwv & 0xffu), cprev = (int)(cwv >> 8);
        const double ti = valid ? sv.tw[idx] : 0.0;
        const double *xp = sv.Xw + (size_t)q * KP * W + lane;
        double x[KP];
#pragma unroll
        for (int j = 0; j < KP; ++j) x[j] = xp[j * W];
        double ra = zval(qa_, q, c, ti), rb_ = zval(qb_, q, c, ti);
        if (!valid) { ra = 0.0; rb_ = 0.0; }
        rt1a = __builtin_fma(ra, ti, rt1a); rt2a = rt2a + ra;
        rt1b = __builtin_fma(rb_, ti, rt1b); rt2b = rt2b + rb_;
        for (int j = cprev; j < c; ++j) { wl.tp1[j] = rt1a; wl.tp2[j] = rt2a; gx.tp1[j] = rt1b; gx.tp2[j] = rt2b; }
#pragma unroll
        for (int j = 0; j < KP; ++j) { acca[j] = __builtin_fma(x[j], ra, acca[j]); accb[j] = __builtin_fma(x[j], rb_, accb[j]); }
    }
    {
        const double s1 = suffix_scan(rt1a), s2v = suffix_scan(rt2a);
        wl.tot1[lane] = s1; wl.tot2[lane] = s2v;
        const double s1b = suffix_scan(rt1b), s2b = suffix_scan(rt2b);
        gx.tot1[lane] = s1b; gx.tot2[lane] = s2b;
        if (lane == 0) { wl.tot1[W] = 0.0; wl.tot2[W] = 0.0; gx.tot1[W] = 0.0; gx.tot2[W] = 0.0; }
    }
    constexpr int G8 = (KP / 8) * 8;
#pragma unroll
    for (int g0 = 0; g0 < G8; g0 += 8) {
        double ta[8], tb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { ta[u] = acca[g0 + u]; tb[u] = accb[g0 + u]; }
        column_sums_g<8>(ta, wl.accR + g0);
        column_sums_g<8>(tb, gx.accR + g0);
    }
    if (KP % 8 != 0) {
        double ta[4], tb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { ta[u] = acca[(G8 + u) < KP ? G8 + u : 0]; tb[u] = accb[(G8 + u) < KP ? G8 + u : 0]; }
        column_sums_g<4>(ta, wl.accR + G8);
        column_sums_g<4>(tb, gx.accR + G8);
    }
    wave_sync();
    {
        const int p = lane;
        double va = 0.0, vb = 0.0;
        if (p == 0) { va = wl.tot1[0]; vb = gx.tot1[0]; }
        else if (p == 1) { va = wl.tot2[0]; vb = gx.tot2[0]; }
        else if (p >= 3 && p < 3 + S) {
            const int j = p - 3, Lj = sv.Ljp_l[0];
            va = (wl.tp1[j] + wl.tot1[Lj + 1]) - sv.tcp_l[0] * (wl.tp2[j] + wl.tot2[Lj + 1]);
            vb = (gx.tp1[j] + gx.tot1[Lj + 1]) - sv.tcp_l[0] * (gx.tp2[j] + gx.tot2[Lj + 1]);
        } else if (p >= 3 + S && p < sv.P) {
            va = wl.accR[p - 3 - S]; vb = gx.accR[p - 3 - S];
        }
        za = va; zb = vb;
    }
    wave_sync();
}

// NC columns of M in one pass over the rows, the Fourier columns of a row EXPANDED from its base pairs (FitArgs::Bw,
// harm_row of tsf_fit_kernels.h: the values are the bits setup_grid_kernel wrote into Xw) instead of read: a pass
// loads the row's base pairs, its dense columns behind the Fourier block and the NC weights -- 9 values where
// gram_columns2 loads 30 -- and without the row's 28 values in registers a third column's accumulators fit, so a
// series' Z^T Z takes 18 passes instead of 27 (round 5: a ragged panel with a calendar per series read its tables
// 27 times per series, 31-33 GB per launch).  Entry by entry the chains are gram_columns2's: same operands, same
// order, same bits.  qs[c] < 0: no such column (the last pass), its results are dropped.
template <int KP, int HARM, int NC>
__device__ __forceinline__ void gram_columns_harm(const SeriesView &sv, QuadLds<KP, 1> &wl, GramX *gx,
                                                  const int (&qs)[NC], double (&z)[NC])
{
    static_assert(NC >= 1 && NC <= 3 && KP == 28, "written for the 28-column kernel");
    constexpr int KF = harm_kf(HARM), NS = harm_ns(HARM);
    const int lane = lane_id();
    const int S = sv.S;
    double acc[NC][KP];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < KP; ++j) acc[c][j] = 0.0;
    double rt1[NC], rt2[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { rt1[c] = 0.0; rt2[c] = 0.0; }
    double *tp1[NC], *tp2[NC], *tot1[NC], *tot2[NC], *accR[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        tp1[c] = c == 0 ? wl.tp1 : gx[c - 1].tp1; tp2[c] = c == 0 ? wl.tp2 : gx[c - 1].tp2;
        tot1[c] = c == 0 ? wl.tot1 : gx[c - 1].tot1; tot2[c] = c == 0 ? wl.tot2 : gx[c - 1].tot2;
        accR[c] = c == 0 ? wl.accR : gx[c - 1].accR;
    }
    // The loads of step q - 1 are issued before the arithmetic of step q (a step is ~140 dependent-free fmas: enough to
    // cover them); the dense columns behind the Fourier block only where the model has any (n_xd).
    const bool has_xd = sv.P - 3 - S > KF;
    bool wdes[NC];                      // the weight is a design column (read), not a trend column (formed from t)
    int wcol[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { wdes[c] = qs[c] >= 3 + S; wcol[c] = wdes[c] ? qs[c] - 3 - S : 0; }
    double tcq[NC];                     // the changepoint time of a trend column (read once per pass, not per row)
#pragma unroll
    for (int c = 0; c < NC; ++c) tcq[c] = (qs[c] >= 3 && qs[c] < 3 + S) ? sv.t_change[qs[c] - 3] : 0.0;
    struct Row { double2 bp[NS]; double w[NC]; double ti; unsigned cwv; };
    auto fetch = [&](int q, Row &rw) {
        const int idx = q * W + lane;
        rw.cwv = (unsigned)sv.cw[idx];
        rw.ti = sv.tw[idx];
        const double *xp = sv.Xw + (size_t)q * KP * W + lane;
        const double2 *bq = reinterpret_cast<const double2 *>(sv.Bw) + (size_t)q * NS * W + lane;
#pragma unroll
        for (int se = 0; se < NS; ++se) rw.bp[se] = bq[se * W];
#pragma unroll
        for (int c = 0; c < NC; ++c) rw.w[c] = wdes[c] ? xp[(size_t)wcol[c] * W] : 0.0;
    };
    Row cur;
    fetch(sv.NT - 1, cur);
#pragma unroll 1
    for (int q = sv.NT - 1; q >= 0; --q) {
        Row nxt;
        fetch(q > 0 ? q - 1 : 0, nxt);
        const bool valid = q < sv.cnt;
        const unsigned cwv = valid ? cur.cwv : 0u;
        const int cseg = (int)(cwv & 0xffu), cprev = (int)(cwv >> 8);
        const double ti = valid ? cur.ti : 0.0;
        double2 bp[NS];
#pragma unroll
        for (int se = 0; se < NS; ++se) {
            // (rows a chunk does not have were never written in Bw; any finite pair will do: their weights are 0)
            bp[se].x = valid ? cur.bp[se].x : 0.0; bp[se].y = valid ? cur.bp[se].y : 0.0;
        }
        double r[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int qq = qs[c];
            double rv;
            if (qq == 0) rv = ti;
            else if (qq == 1) rv = 1.0;
            else if (qq < 3 + S) rv = (cseg > qq - 3) ? ti - tcq[c] : 0.0;
            else rv = cur.w[c];
            r[c] = (valid && qq >= 0) ? rv : 0.0;
            rt1[c] = __builtin_fma(r[c], ti, rt1[c]); rt2[c] = rt2[c] + r[c];
        }
        for (int j = cprev; j < cseg; ++j) {
#pragma unroll
            for (int c = 0; c < NC; ++c) { tp1[c][j] = rt1[c]; tp2[c][j] = rt2[c]; }
        }
        harm_row<HARM>(bp, [&](int j, double v) {
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c][j] = __builtin_fma(v, r[c], acc[c][j]);
        });
        if (has_xd) {
            const double *xp = sv.Xw + (size_t)q * KP * W + lane;
#pragma unroll
            for (int j = KF; j < KP; ++j) {
                const double xv = xp[j * W];
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c][j] = __builtin_fma(xv, r[c], acc[c][j]);
            }
        }
        cur = nxt;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double s1 = suffix_scan(rt1[c]), s2v = suffix_scan(rt2[c]);
        tot1[c][lane] = s1; tot2[c][lane] = s2v;
        if (lane == 0) { tot1[c][W] = 0.0; tot2[c][W] = 0.0; }
    }
    constexpr int G8 = (KP / 8) * 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int g0 = 0; g0 < G8; g0 += 8) {
            double ta[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) ta[u] = acc[c][g0 + u];
            column_sums_g<8>(ta, accR[c] + g0);
        }
        if (KP % 8 != 0) {
            double ta[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ta[u] = acc[c][(G8 + u) < KP ? G8 + u : 0];
            column_sums_g<4>(ta, accR[c] + G8);
        }
    }
    wave_sync();
    {
        const int p = lane;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            double v = 0.0;
            if (p == 0) v = tot1[c][0];
            else if (p == 1) v = tot2[c][0];
            else if (p >= 3 && p < 3 + S) {
                const int j = p - 3, Lj = sv.Ljp_l[0];
                v = (tp1[c][j] + tot1[c][Lj + 1]) - sv.tcp_l[0] * (tp2[c][j] + tot2[c][Lj + 1]);
            } else if (p >= 3 + S && p < sv.P) {
                v = accR[c][p - 3 - S];
            }
            z[c] = v;
        }
    }
    wave_sync();
}

template <int KP, int PPL>
__global__ __launch_bounds__(64) void gram_build_kernel(QuadArgs qa, double *Mout)
{
    __shared__ QuadLds<KP, PPL> wl;
    const int q = blockIdx.x, lane = threadIdx.x;
    SeriesView sv;
    make_view_q<KP, PPL>(qa.f, 0, sv);
    double *out = Mout + (size_t)q * PPL * W;
    if (q == 2 || q >= sv.P) {
#pragma unroll
        for (int s = 0; s < PPL; ++s) out[s * W + lane] = 0.0;
        return;
    }
    double *rb = qa.rbuf + (size_t)q * qa.f.NTmax * W;
    double ztr[PPL];
    gram_column<KP, PPL>(sv, wl, rb, q, ztr);
#pragma unroll
    for (int s = 0; s < PPL; ++s) out[s * W + lane] = ztr[s];
}

// Ragged panel whose series share timestamp vectors (FitArgs::grid_of): Z^T Z of grids g0 .. g0 + gridDim.y - 1,
// one workgroup per (column, grid), into Mpre[g][P4][PPL][64].  The same gram_column as above, so the same bits
// the fit kernel's own build gives (tests: TSF_GRAM_SHARE=0 against the default).  gridDim.y * P4 <= slots of rbuf.
template <int KP, int PPL>
__global__ __launch_bounds__(64) void gram_grids_kernel(QuadArgs qa, double *Mpre, int64_t g0)
{
    __shared__ QuadLds<KP, PPL> wl;
    const int q = blockIdx.x, lane = threadIdx.x;
    const int64_t g = g0 + blockIdx.y;
    SeriesView sv;
    make_view_grid<KP, PPL>(qa.f, g, 0, sv);          // series 0's y is not read: the columns of Z depend on the grid alone
    double *out = Mpre + ((size_t)g * qa.P4 + q) * PPL * W;
    if (q == 2 || q >= sv.P) {
#pragma unroll
        for (int s = 0; s < PPL; ++s) out[s * W + lane] = 0.0;
        return;
    }
    double *rb = qa.rbuf + ((size_t)blockIdx.y * qa.P4 + q) * qa.f.NTmax * W;
    double ztr[PPL];
    gram_column<KP, PPL>(sv, wl, rb, q, ztr);
#pragma unroll
    for (int s = 0; s < PPL; ++s) out[s * W + lane] = ztr[s];
}

// ---------------------------------------------------------------------------------------
// the fit kernel
// ---------------------------------------------------------------------------------------
#ifdef TSF_QUAD_TIMING      // dev only: cycles per phase (s_memtime), summed per series into qa.dbg
#define QT_DECL long long qt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long qt_t0 = __builtin_readcyclecounter(), qt_start = qt_t0
#define QT_LAP(k) do { const long long t_ = __builtin_readcyclecounter(); qt_acc[k] += t_ - qt_t0; qt_t0 = t_; } while (0)
#define QT_FLUSH() do { if (qa.dbg && lane == 0) { qt_acc[7] = __builtin_readcyclecounter() - qt_start; for (int k_ = 0; k_ < 8; ++k_) qa.dbg[(size_t)n * 8 + k_] = qt_acc[k_]; } } while (0)
#else
#define QT_DECL do { } while (0)
#define QT_LAP(k) do { } while (0)
#define QT_FLUSH() do { } while (0)
#endif

// One series, start to finish, by one wave.
// HLDS: the L-BFGS history lives in the wave's LDS ring `hist` ([2][QH][PPL][64]: s then y, slot of
// age h = (h0 + h) mod QH) instead of 4 QH PPL registers per lane -- what lets a third wave per
// SIMD fit the register file (tsf_inst_quad.hip); same values, same operation order.
// NTR > 0: the weights of a residual pass in NTR registers (ztr_pass), rb unused.
// What setup_series_kernel derives from a series -- y_scale = max |y|, the constant-history flag, fbprophet's
// linear_growth_init (k, m) from the first row and the first row that holds the last timestamp -- computed by the wave that
// is about to fit the series, from the caller's own rows (round 6: the quadratic-form route has no set-up pass over y and
// no scaled copy of it; the step moves 1.6 x its algorithmic bytes instead of 4.1 x).  Same operations on the same
// operands as the kernel's (linear growth: the floor is 0 and (y - 0) / y_scale is y / y_scale; the two scaled times are
// the grid table's own values, (ds - start) / t_scale by the same expression), so the same bits.
__device__ __forceinline__ SeriesTab series_tab_wave(const SeriesView &sv, const GridTab &gt)
{
    SeriesTab st;
    st.cap = 0.0; st.floor_ = 0.0; st.k0 = 0.0; st.m0 = 0.0; st.y_scale = 1.0; st.status0 = 0; st.pad_ = 0;
    const int T = sv.T, NT = sv.NT;
    if (T < 2) { st.status0 = TSF_ST_TOO_FEW; return st; }
    const int lane = lane_id();
    double amax = 0.0, ymin = __builtin_huge_val(), ymax = -__builtin_huge_val();
    for (int i = lane; i < T; i += W) {
        const double v = load_y(sv.y_raw, sv.y_dtype, sv.y_base + i);
        amax = __builtin_fmax(amax, __builtin_fabs(v - 0.0));
        ymin = __builtin_fmin(ymin, v);
        ymax = __builtin_fmax(ymax, v);
    }
#pragma unroll
    for (int off = 1; off < W; off <<= 1) {
        amax = __builtin_fmax(amax, __shfl_xor(amax, off, W));
        ymin = __builtin_fmin(ymin, __shfl_xor(ymin, off, W));
        ymax = __builtin_fmax(ymax, __shfl_xor(ymax, off, W));
    }
    const double ys = (amax == 0.0) ? 1.0 : amax;
    const int i1 = gt.info.i1;
    const double t0 = sv.tw[0];                                           // row 0: chunk 0, step 0
    const double t1 = sv.tw[(i1 - (i1 / NT) * NT) * W + i1 / NT];
    const double y0 = (load_y(sv.y_raw, sv.y_dtype, sv.y_base) - 0.0) / ys;
    const double y1 = (load_y(sv.y_raw, sv.y_dtype, sv.y_base + i1) - 0.0) / ys;
    const double Td = t1 - t0;
    st.k0 = (y1 - y0) / Td;
    st.m0 = y0 - st.k0 * t0;
    if (ymin == ymax) st.status0 = TSF_ST_CONSTANT;
    st.y_scale = ys;
    return st;
}

// POOL: wlp is null; a residual pass borrows a QuadLds of `pool` and the vectors that live across
// evaluations (D, ref, c) sit in the wave's QuadWave `qw`.
template <int KP, int PPL, int PQ, bool RAGGED, int MRS = W, bool HLDS = false, int MBATCH = 16, bool MREG = false, int NTR = 0, bool POOL = false,
          bool MPIPE = (TSF_QUAD_MPIPE != 0) && !POOL>     // (the 128-register kernel has no room for a second batch in flight)
__device__ __forceinline__ bool fit_one_quad(const QuadArgs &qa, QuadLds<KP, PPL> *wlp, double *rb,
                                          const double *Mp, double *Mown, int64_t n, double *lanec,
                                          double *hist = nullptr, GramX *gx = nullptr,
                                          QuadWave<PPL> *qw = nullptr, const QuadPool *pool = nullptr)
{
    static_assert(!POOL || (!RAGGED && PQ > 0 && PPL == 1), "pooled trend tables: the shared-M one-slot kernel");
    double *const dl_w = POOL ? qw->dl : wlp->th;
    double *const ref_w = POOL ? qw->ref : wlp->ref;
    double *const cvec_w = POOL ? qw->cvec : wlp->cvec;
    const FitArgs &a = qa.f;
    const DevSpec *sp = a.sp;
    const int lane = lane_id();
    const int P4 = qa.P4;
    const double eps = 2.220446049250313e-16;
    const double c1 = 1e-4, c2 = 0.9, minAlpha = 1e-12, min_range = 1e-16;
    const int maxLSIts = 20, maxLSRestarts = 10;
    SeriesView sv;
    make_view_q<KP, PPL>(a, n, sv);
    const SeriesTab st = a.y_raw ? series_tab_wave(sv, a.gtab[grid_index(a, n)]) : a.stab[n];
    sv.y_scl = st.y_scale;
    if (lane == 0) {
        a.y_scale[n] = st.y_scale;
        if (!a.aligned || n == 0) a.grid_out[a.aligned ? 0 : n] = a.gtab[grid_index(a, n)].info;
    }
    double xk[PPL], gk[PPL], pk[PPL], xk1[PPL], gk1[PPL];
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const int p = lane + s * W;
        xk[s] = (p == 0) ? st.k0 : (p == 1 ? st.m0 : 0.0);
        gk[s] = 0.0; pk[s] = 0.0; xk1[s] = xk[s]; gk1[s] = 0.0;
    }
    if (st.status0 != 0) {
        if (st.status0 == TSF_ST_CONSTANT) {
#pragma unroll
            for (int s = 0; s < PPL; ++s) if (lane + s * W == 2) xk[s] = -20.72326583694641;
        }
        store_theta<PPL>(a, sv, n, xk, a.theta);
        if (lane == 0) { a.status[n] = st.status0; a.n_iter[n] = 0; a.n_eval[n] = 0; a.fval[n] = 0.0; }
        return false;
    }
    LaneConst<PPL> lk;
    quad_const_table(lanec + 3 * PPL * W, a.opt, qa.recenter_ratio);
    lane_consts<PPL>(sp, sv, lanec, lk);
    lk.ct = lanec + 3 * PPL * W;
    // The table is READ by the aligned shared-M kernels only (12 and 16 waves per CU, register budgets 168 / 128: the
    // tolerances as kernel arguments were SGPR pairs spilled into VGPR lanes there, ~22 v_readlane per iteration).
    // The ragged and two-slot kernels run at 256 registers per lane with nothing to spare, and ANY read of the table
    // inside their iteration loop made the register allocator spill 40 more values per lane to scratch (measured:
    // tools/dev one-kernel compiles, ragged panels 15.3 -> 18 ms); they keep reading the kernel arguments.
    constexpr bool CT = HLDS && !RAGGED && !MREG && PQ > 0 && PPL == 1;
    const double *const ct = lk.ct;
    QT_DECL;
    if (RAGGED && MRS == W && qa.Mpre) {
        // ragged panel whose series share timestamp vectors: gram_grids_kernel built the M of this series' grid
        // (the M-in-LDS kernel, compact rows, keeps building its own: long series, where the build is the smaller part)
        Mp = qa.Mpre + (size_t)grid_index(a, n) * P4 * PPL * W;
    } else if (RAGGED) {
        // ragged panel: this series has its own grid, hence its own M = Z^T Z.  The wave builds
        // it column by column into its slot of global memory; lane p writes and later reads only
        // entries of its own row p, so no fence is needed.
        bool built = false;
        if constexpr (MREG && PPL == 1 && KP == 28) {
            if (qa.gram_harm == HARM_Y10_W3) {
                // three columns per pass, the Fourier columns expanded from the rows' base pairs (gram_columns_harm)
                int qs[3] = {-1, -1, -1}, nq = 0;
#pragma unroll 1
                for (int q = 0; q <= P4; ++q) {
                    if (q < P4) {
                        const bool real = q != 2 && q < sv.P;
                        if (!real) { Mown[(size_t)q * W + lane] = 0.0; continue; }
                        qs[nq++] = q;
                        if (nq < 3) continue;
                    } else if (nq == 0) break;
                    double z[3];
                    gram_columns_harm<KP, HARM_Y10_W3, 3>(sv, *wlp, gx, qs, z);
#pragma unroll
                    for (int c = 0; c < 3; ++c) if (qs[c] >= 0) Mown[(size_t)qs[c] * W + lane] = z[c];
                    qs[0] = -1; qs[1] = -1; qs[2] = -1; nq = 0;
                }
                built = true;
            }
        }
        if (built) {
        } else if constexpr (MREG && PPL == 1) {
            // two columns per pass (gram_columns2); rows 2 and >= P of M are zero
            int qprev = -1;
#pragma unroll 1
            for (int q = 0; q < P4; ++q) {
                const bool real = q != 2 && q < sv.P;
                if (!real) { Mown[(size_t)q * W + lane] = 0.0; continue; }
                if (qprev < 0) { qprev = q; continue; }
                double za, zb;
                gram_columns2<KP>(sv, *wlp, *gx, qprev, q, za, zb);
                Mown[(size_t)qprev * W + lane] = za;
                Mown[(size_t)q * W + lane] = zb;
                qprev = -1;
            }
            if (qprev >= 0) {           // an odd column left over
                double za, zb;
                gram_columns2<KP>(sv, *wlp, *gx, qprev, qprev, za, zb);
                Mown[(size_t)qprev * W + lane] = za;
            }
        } else {
#pragma unroll 1
        for (int q = 0; q < P4; ++q) {
            double col[PPL];
#pragma unroll
            for (int s = 0; s < PPL; ++s) col[s] = 0.0;
            if (q != 2 && q < sv.P) gram_column<KP, PPL>(sv, *wlp, rb, q, col);
            if (MRS == W) {
#pragma unroll
                for (int s = 0; s < PPL; ++s) Mown[((size_t)q * PPL + s) * W + lane] = col[s];
            } else if (lane < MRS) {
                Mown[(size_t)q * MRS + lane] = col[0];
            }
        }
        }
        Mp = Mown;
        QT_LAP(5);
    }
    // MREG: the matrix just built goes from its global slot into registers (lane p: row p; the lane
    // reads back what it wrote itself)
    double mreg[(MREG && PQ > 0) ? PQ : 1];
    if constexpr (MREG && PQ > 0) {
        static_assert(!MREG || (MRS == W && PPL == 1), "register M: one-slot kernels");
        const double *Msrc = Mp;                        // aligned panels: the call's one matrix; ragged: just built (Mown) or prebuilt
#pragma unroll
        for (int q = 0; q < PQ; ++q) mreg[q] = Msrc[(size_t)q * W + lane];
    } else {
        mreg[0] = 0.0;
    }

    double s0 = 0.0, q2 = 0.0;
    // L-BFGS history in registers (or in LDS: HLDS), age order (index 0 = oldest)
    constexpr int QHR = HLDS ? 1 : QH;
    double ShR[QHR][PPL], YhR[QHR][PPL], rh[QH];
#pragma unroll
    for (int h = 0; h < QH; ++h) rh[h] = 0.0;
#pragma unroll
    for (int h = 0; h < QHR; ++h) {
#pragma unroll
        for (int s = 0; s < PPL; ++s) { ShR[h][s] = 0.0; YhR[h][s] = 0.0; }
    }
    int h0 = 0;                          // HLDS: ring slot of the oldest pair
    double *const histS = hist, *const histY = hist + QH * PPL * W;
    // HLDS: rho of the pair in ring slot s at histR[s] (round 4: as five scalars in age order they were five SGPR
    // pairs live across the whole fit, shifted by select chains and spilled into VGPR lanes every iteration)
    double *const histR = hist + 2 * QH * PPL * W;

    double fk = 0.0, fk1 = 0.0, alpha = a.opt.init_alpha, gammak = 1.0;
    int itNum = 0, ret = 0, resetB = 0, hist_len = 0, since_rc = 0;
    double dfp = 0, c1dfp = 0, c2dfp = 0, alpha0 = 0, prevF = 0, prevDFp = 0;
    double alo = 0, aloF = 0, aloDFp = 0, ahi = 0, ahiF = 0, ahiDFp = 0;
    int nits = 0, lsRestarts = 0, zoom = 0, zit = 0;
    // g.p of the current and of the previous iterate are each needed two or three times per
    // iteration (termination test, cubic interpolation, line-search slope): same operands, same
    // bits, so they are computed once and carried
    double gp = 0.0;
    bool gp_valid = false, pk1_scaled = false;
    double gp1s = 0.0;                  // g_{k-1}.(p_{k-1} / B0fact) where the previous direction was rescaled (below)

    const int eval_limit = 64 * a.opt.max_iter + 1024;      // guard, see cn_lbfgs (oracle)
    // Residual-form evaluation at xk (cn_resid_q): the initial point, and a re-centring of the
    // quadratic form at an accepted iterate.  When it is finite it becomes the reference point.
    auto recenter = [&](double &fe_out) -> bool {
        double xe[PPL], ge[PPL], fe, sse_e, ztr_e[PPL];
#pragma unroll
        for (int s = 0; s < PPL; ++s) xe[s] = xk[s];
        sv.n_eval++;
        QT_LAP(2);
        bool bad;
        if constexpr (POOL) {
            const int slot = pool_acquire(*pool);
            unsigned char *sl = pool->slots + (size_t)slot * pool->slot_bytes;
            // short series: the slot also holds the staging rows (slot_bytes says so); else the global scratch
            double *rbs = pool->slot_bytes > sizeof(QuadLds<KP, PPL>) ? reinterpret_cast<double *>(sl + sizeof(QuadLds<KP, PPL>)) : rb;
            bad = resid_eval_q<KP, PPL, NTR>(sv, *reinterpret_cast<QuadLds<KP, PPL> *>(sl), lk, rbs, xe, fe, ge, sse_e, ztr_e);
            pool_release(*pool, slot);
        } else if constexpr (RAGGED && MREG && PPL == 1 && KP == 28 && NTR == 0) {
            // a calendar per series: the rows as base pairs here too (QuadArgs::gram_harm)
            if (qa.gram_harm == HARM_Y10_W3) bad = resid_eval_q<KP, PPL, NTR, HARM_Y10_W3>(sv, *wlp, lk, rb, xe, fe, ge, sse_e, ztr_e);
            else bad = resid_eval_q<KP, PPL, NTR>(sv, *wlp, lk, rb, xe, fe, ge, sse_e, ztr_e);
        } else {
            bad = resid_eval_q<KP, PPL, NTR>(sv, *wlp, lk, rb, xe, fe, ge, sse_e, ztr_e);
        }
        QT_LAP(3);
        if (!bad) {
#pragma unroll
            for (int s = 0; s < PPL; ++s) {
                const int p = lane + s * W;
                ref_w[p] = (p == 2) ? 0.0 : xe[s];
                cvec_w[p] = ztr_e[s];
                gk[s] = ge[s];
            }
            s0 = sse_e; since_rc = 0; fk = UQ(fe);
            wave_sync();
        }
        fe_out = fe;
        return bad;
    };
    // Structured as Stan's loops are (iterations > line searches > trials) rather than as a state
    // machine around one evaluation site: the compiler then keeps the optimiser state in place
    // instead of shuffling ~40 registers at the joins of the state machine's edges.  One pass of
    // the outer loop = [residual pass at x_k: the initial point, or a re-centring] + [the L-BFGS
    // update and the termination tests for x_k] + the line search from x_k; each evaluation form
    // still has a single call site.
    bool first = true, do_resid = true;
    for (;;) {
        QT_LAP(0);
        if (do_resid) {
            double fe0;
            const bool bad0 = recenter(fe0);
            if (first && bad0) { ret = TSF_ST_INIT_NONFINITE; fk = UQ(fe0); break; }
        }
        if (first) {
            first = false;
#pragma unroll
            for (int s = 0; s < PPL; ++s) { pk[s] = -gk[s]; gk1[s] = 0.0; xk1[s] = 0.0; }
        } else {
            // ---- accepted step: k is the most recent iterate ----
            double sk[PPL], yk[PPL];
#pragma unroll
            for (int s = 0; s < PPL; ++s) { sk[s] = xk[s] - xk1[s]; yk[s] = gk[s] - gk1[s]; }
            // g.g, s.s, y.s, y.y in one four-fold butterfly that leaves them in lanes 0, 2, 1, 3; the two
            // square roots and the three quotients (y.y / y.s, y.s / y.y, 1 / y.s) are independent:
            // one lane each, operands moved into place inside the quad
            const double dots = bfly_sum4_lanes(pdot_part<PPL>(gk, gk), pdot_part<PPL>(sk, sk),
                                                pdot_part<PPL>(yk, sk), pdot_part<PPL>(yk, yk));
            // |g| and |s| are only ever COMPARED with a tolerance: the squares decide wherever they are clearly on
            // one side (quad_const_table: brackets of width 8 eps around tol^2), the square root is taken otherwise
            // -- same outcome as `sqrt(x2) < tol` for every x2, NaN included
            const double g2sum = readlane_f64(dots, 0), s2sum = readlane_f64(dots, 2);
            auto norm_below = [&](double x2, int k_lo, int k_tol) -> bool {
                if (x2 < qc(ct, k_lo)) return true;
                if (x2 >= qc(ct, k_lo + 1)) return false;
                return __builtin_sqrt(x2) < qc(ct, k_tol);
            };
            double qnum = dpp_mov<0x07>(dots);          // quad_perm [3,1,0,0]: y.y, y.s, -, -
            if ((lane & 3) >= 2) qnum = 1.0;
            const double qden = dpp_mov<0x5D>(dots);    // quad_perm [1,3,1,1]: y.s, y.y, y.s, y.s
            const double qv = qnum / qden;
            if (resetB) {
                const double B0fact = readlane_f64(qv, 0);
                hist_len = 0;
                // Stan rescales the previous direction here and the next line search's first step is interpolated from
                // g_{k-1}.p_{k-1} with the rescaled p_{k-1}: the only use of that vector, so the dot product is taken now
                // (pk still holds the direction of the line search that just ended, gk1 the gradient it started from)
                // and the vector itself is not kept -- two registers per lane and a swap per iteration less
                double ps[PPL];
#pragma unroll
                for (int s = 0; s < PPL; ++s) ps[s] = pk[s] / B0fact;
                gp1s = pdot<PPL>(gk1, ps);
                alpha = UQ(alpha * B0fact);
                pk1_scaled = true;
            } else {
                pk1_scaled = false;
            }
            gammak = readlane_f64(qv, 1);
            const double rho_new = readlane_f64(qv, 2);
            double Sh[QH][PPL], Yh[QH][PPL];        // age order; HLDS: this iteration's copy of the ring
            if (HLDS) {
                if (resetB) h0 = 0;
                int slot;
                if (hist_len < QH) {
                    slot = h0 + hist_len; if (slot >= QH) slot -= QH;
                    hist_len++;
                } else {
                    slot = h0; h0 = (h0 + 1 == QH) ? 0 : h0 + 1;
                }
#pragma unroll
                for (int s = 0; s < PPL; ++s) {
                    histS[(slot * PPL + s) * W + lane] = sk[s];
                    histY[(slot * PPL + s) * W + lane] = yk[s];
                }
                histR[slot] = rho_new;          // (every lane: the same value to the same word)
                wave_sync();
#pragma unroll
                for (int h = 0; h < QH; ++h) {
                    int sl = h0 + h; if (sl >= QH) sl -= QH;
#pragma unroll
                    for (int s = 0; s < PPL; ++s) {
                        Sh[h][s] = histS[(sl * PPL + s) * W + lane];
                        Yh[h][s] = histY[(sl * PPL + s) * W + lane];
                    }
                    rh[h] = histR[sl];          // (slots of age >= hist_len: stale or unset, never used)
                }
            } else {
            if (hist_len < QH) {
#pragma unroll
                for (int h = 0; h < QH; ++h) {
                    if (h == hist_len) {
                        rh[h] = rho_new;
#pragma unroll
                        for (int s = 0; s < PPL; ++s) { ShR[h][s] = sk[s]; YhR[h][s] = yk[s]; }
                    }
                }
                hist_len++;
            } else {
#pragma unroll
                for (int h = 0; h + 1 < QH; ++h) {
                    rh[h] = rh[h + 1];
#pragma unroll
                    for (int s = 0; s < PPL; ++s) { ShR[h][s] = ShR[h + 1][s]; YhR[h][s] = YhR[h + 1][s]; }
                }
                rh[QH - 1] = rho_new;
#pragma unroll
                for (int s = 0; s < PPL; ++s) { ShR[QH - 1][s] = sk[s]; YhR[QH - 1][s] = yk[s]; }
            }
#pragma unroll
            for (int h = 0; h < QH; ++h) {
#pragma unroll
                for (int s = 0; s < PPL; ++s) { Sh[h][s] = ShR[HLDS ? 0 : h][s]; Yh[h][s] = YhR[HLDS ? 0 : h][s]; }
            }
            }
            double alphas[QH];
            QT_LAP(1);
#pragma unroll
            for (int s = 0; s < PPL; ++s) pk[s] = -gk[s];
#pragma unroll
            for (int h = QH - 1; h >= 0; --h) {
                if (h < hist_len) {
                    // (alphas as the VECTORS whose lane 63 holds the value -- the second loop uses them in lane 63 only --
                    // instead of five scalars saves 29 vector instructions per iteration (no SGPR pairs spilled into
                    // VGPR lanes) and costs ten registers per lane across the recursion: the kernel then spills to
                    // scratch, 114 instead of 80 MB of HBM traffic per cfg2 launch for 1.5 % of its time
                    // (profiles/r04_quad/ab_alphas.txt).  Scalars stay: no scratch at all.)
                    const double aa = lane63(rh[h] * pdot_l63<PPL>(Sh[h], pk));
#pragma unroll
                    for (int s = 0; s < PPL; ++s) pk[s] = __builtin_fma(-aa, Yh[h][s], pk[s]);
                    alphas[h] = aa;
                }
            }
#pragma unroll
            for (int s = 0; s < PPL; ++s) pk[s] = pk[s] * gammak;
#pragma unroll
            for (int h = 0; h < QH; ++h) {
                if (h < hist_len) {
                    const double cc = lane63(alphas[h] - rh[h] * pdot_l63<PPL>(Yh[h], pk));
#pragma unroll
                    for (int s = 0; s < PPL; ++s) pk[s] = __builtin_fma(cc, Sh[h][s], pk[s]);
                }
            }
            QT_LAP(6);                  // the two-loop recursion
            const double dF = __builtin_fabs(fk1 - fk);
            const double fmaxv = __builtin_fmax(__builtin_fabs(fk1),
                                                __builtin_fmax(__builtin_fabs(fk), 1.0));
            gp = pdot<PPL>(gk, pk);
            gp_valid = true;
            // -g.p / max(|f|, 1) < tol_rel_grad: the same bracketing for the quotient (m >= 1)
            auto relgrad_below = [&]() -> bool {
                const double m = __builtin_fmax(__builtin_fabs(fk), 1.0), ngp = -gp;
                if (ngp < qc(ct, QC_RELGRAD_LO) * m) return true;
                if (ngp > qc(ct, QC_RELGRAD_HI) * m) return false;
                return ngp / m < qc(ct, QC_TOL_REL_GRAD);
            };
            if constexpr (CT) {
                if (dF < qc(ct, QC_TOL_OBJ)) ret = TSF_ST_ABSF;
                else if (dF < qc(ct, QC_TOL_REL_OBJ) * fmaxv) ret = TSF_ST_RELF;
                else if (norm_below(g2sum, QC_GRAD2_LO, QC_TOL_GRAD)) ret = TSF_ST_ABSGRAD;
                else if (relgrad_below()) ret = TSF_ST_RELGRAD;
                else if (norm_below(s2sum, QC_PARAM2_LO, QC_TOL_PARAM)) ret = TSF_ST_ABSX;
                else if (itNum >= a.opt.max_iter) ret = TSF_ST_MAXIT;
                else ret = 0;
            } else {
                if (dF < a.opt.tol_obj) ret = TSF_ST_ABSF;
                else if (dF < a.opt.tol_rel_obj_eps * fmaxv) ret = TSF_ST_RELF;
                else if (__builtin_sqrt(g2sum) < a.opt.tol_grad) ret = TSF_ST_ABSGRAD;
                else if (-gp / __builtin_fmax(__builtin_fabs(fk), 1.0) < a.opt.tol_rel_grad_eps) ret = TSF_ST_RELGRAD;
                else if (__builtin_sqrt(s2sum) < a.opt.tol_param) ret = TSF_ST_ABSX;
                else if (itNum >= a.opt.max_iter) ret = TSF_ST_MAXIT;
                else ret = 0;
            }

            QT_LAP(1);
            if (ret != 0) break;
        }
        itNum++;
        resetB = (itNum == 1) ? 1 : 0;
        bool accepted = false;
        for (;;) {                                      // line search; once more from -g if it fails
            if (resetB) {
#pragma unroll
                for (int s = 0; s < PPL; ++s) pk[s] = -gk[s];
                gp_valid = false;
            }
            if (!gp_valid) gp = pdot<PPL>(gk, pk);
            gp_valid = false;
            if (itNum > 1 && resetB != 2) {
                // g_{k-1}.p_{k-1} is the slope `dfp` of the previous line search unless p_{k-1}
                // has been rescaled since
                const double gp1 = pk1_scaled ? gp1s : dfp;
                const double ci = cubic_interp6(gp1, alpha, fk - fk1, gp, minAlpha, 1.0);
                alpha = UQ(__builtin_fmin(1.0, 1.01 * ci));
            } else {
                alpha = CT ? qc(ct, QC_INIT_ALPHA) : a.opt.init_alpha;
            }
            dfp = gp;
            c1dfp = UQ(c1 * dfp); c2dfp = UQ(c2 * dfp);
            alpha0 = minAlpha; prevF = fk; prevDFp = dfp;
            nits = 0; lsRestarts = 0; zoom = 0; zit = 0;

            bool ls_fail = false, pre = true;
            for (;;) {                                  // trial points
                if (pre) {
                    if (!zoom) {
                        if (nits >= maxLSIts) ls_fail = true;
                    } else {
                        zit++;
                        if (__builtin_fabs(alo - ahi) < min_range) {
                            ls_fail = true;
                        } else if (zit % 5 == 0) {
                            alpha = UQ(0.5 * (alo + ahi));
                        } else {
                            const double d1 = aloDFp + ahiDFp - 3.0 * (aloF - ahiF) / (alo - ahi);
                            double d2 = __builtin_sqrt(d1 * d1 - aloDFp * ahiDFp);
                            if (ahi < alo) d2 = -d2;
                            alpha = ahi - (ahi - alo) * (ahiDFp + d2 - d1) / (ahiDFp - aloDFp + 2.0 * d2);
                            const double lo = __builtin_fmin(alo, ahi), hi = __builtin_fmax(alo, ahi),
                                         w = __builtin_fabs(alo - ahi);
                            if (!finite_f64(alpha) || alpha < lo + 0.01 * w || alpha > hi - 0.01 * w)
                                alpha = 0.5 * (alo + ahi);
                            alpha = UQ(alpha);
                        }
                    }

                    if (ls_fail) break;
                }
                pre = true;
                if (sv.n_eval >= eval_limit) { ret = TSF_ST_EVAL_LIMIT; break; }
                double xe[PPL], ge[PPL], fe;
#pragma unroll
                for (int s = 0; s < PPL; ++s) { xk1[s] = __builtin_fma(alpha, pk[s], xk[s]); xe[s] = xk1[s]; }
                sv.n_eval++;
                QT_LAP(2);
                const bool bad = gram_eval_q<PPL, PQ, MRS, MBATCH, MREG, MPIPE, (TSF_QUAD_EXPTAB != 0)>(sv, lk, Mp, P4, xe, ref_w, cvec_w, s0, fe, ge, q2, dl_w, mreg);
                QT_LAP(4);
#pragma unroll
                for (int s = 0; s < PPL; ++s) gk1[s] = ge[s];
                const double f1 = UQ(fe);
                if (bad) {
                    if (!zoom) {
                        if (lsRestarts >= maxLSRestarts) ls_fail = true;
                        else { alpha = UQ(0.5 * (alpha0 + alpha)); lsRestarts++; }
                    } else {
                        alpha = UQ(0.5 * (alpha + __builtin_fmin(alo, ahi)));
                        if (__builtin_fabs(__builtin_fmin(alo, ahi) - alpha) < min_range) ls_fail = true;
                    }
                    if (ls_fail) break;
                    pre = false;                        // re-evaluate at the shortened step
                    continue;
                }
                const double newDFp = pdot<PPL>(gk1, pk);
                QT_LAP(5);                  // (aligned panels: slot 5 = from the end of the evaluation to g.p of the trial point)
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
                        alpha = UQ(alpha * 10.0);
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
                if (ls_ok) { fk1 = f1; accepted = true; break; }
            }
            if (accepted || ret != 0) break;
            // line search failed
            if (resetB) { ret = TSF_ST_LSFAIL; break; }
            resetB = 2;
        }
        if (!accepted) break;
        // ---- accepted step: k becomes the most recent iterate ----
        { const double tf = fk; fk = fk1; fk1 = tf; }
#pragma unroll
        for (int s = 0; s < PPL; ++s) {
            const double tx = xk[s]; xk[s] = xk1[s]; xk1[s] = tx;
            const double tg = gk[s]; gk[s] = gk1[s]; gk1[s] = tg;
        }
        since_rc++;
        do_resid = q2 > (CT ? qc(ct, QC_RC_RATIO) : qa.recenter_ratio) * s0 || since_rc >= qa.recenter_every;
    }
    store_theta<PPL>(a, sv, n, xk, a.theta);
    if (lane == 0) { a.status[n] = ret; a.n_iter[n] = itNum; a.n_eval[n] = sv.n_eval; a.fval[n] = fk; }
    QT_FLUSH();
    return false;
}

// MMODE: where M lives -- 0 aligned panel, shared M in LDS; 1 aligned panel, shared M in global
// memory (two-slot kernel: too big for LDS); 2 ragged panel, one M per resident wave in global
// 3 ragged panel, one M per resident wave in LDS (fewer waves per workgroup: an evaluation reads all
// of M, and eight private 28 KB matrices per CU do not fit the 32 KB L1: from global memory the
// ragged fit ran at the L2's pace, 5x the aligned time)
// 4 ragged panel, M of the running series in the wave's registers (built in its global slot first)
enum { QM_LDS = 0, QM_GLOBAL = 1, QM_RAGGED = 2, QM_RAGGED_LDS = 3, QM_RAGGED_REG = 4, QM_GLOBAL_REG = 5 };

// (S and Y of the QH pairs, then their rho = 1 / y.s by ring slot: 8 doubles)
template <int PPL>
constexpr size_t quad_hist_bytes(bool hlds) { return hlds ? sizeof(double) * (2 * QH * PPL * W + 8) : 0; }

// NTR > 0: residual-pass weights of the first NTR steps in registers (RLDS false; steps beyond: global scratch)
// RPOOL (NW = 16, four waves per SIMD at <= 128 registers): trend tables from a pool of pool_slots QuadLds
template <int KP, int PPL, int NW, int MMODE, int PQ, bool RLDS, bool HLDS = false, int NTR = 0, bool RPOOL = false>
__global__ __launch_bounds__(NW * 64, quad_waves_per_simd(quad_three_waves(MMODE, PPL, HLDS), NW)) void fit_quad_kernel(QuadArgs qa, int pool_slots, int pool_slot_bytes)
{
    static_assert(NTR == 0 || (!RLDS && MMODE == QM_LDS), "register-resident weights: the shared-M kernel without LDS staging");
    static_assert(!RPOOL || (MMODE == QM_LDS && !RLDS && PPL == 1 && PQ > 0 && HLDS), "pooled trend tables: the shared-M one-slot kernel");
    extern __shared__ __align__(16) unsigned char smem[];
    const FitArgs &a = qa.f;
    const int lane = lane_id(), wid = (int)threadIdx.x >> 6;
    const int P4 = qa.P4;
    constexpr bool MLDS = MMODE == QM_LDS;
    // LDS: [M (aligned, P <= 64)] [lane constants: one (aligned) or NW (ragged)] [QuadLds x NW]
    //      [history ring x NW (HLDS)] [r staging x NW (RLDS)] [per-wave compact M (QM_RAGGED_LDS)]
    // RPOOL: [M] [lane constants] [QuadWave x NW] [history ring x NW] [locks] [QuadLds x pool_slots]
    double *Ml = reinterpret_cast<double *>(smem);
    const size_t m_bytes = MLDS ? sizeof(double) * (size_t)P4 * PPL * W : 0;
    constexpr bool RAGGED_K = MMODE == QM_RAGGED || MMODE == QM_RAGGED_LDS || MMODE == QM_RAGGED_REG;
    constexpr size_t LCB = quad_lanec_bytes<PPL>();
    double *lanec = reinterpret_cast<double *>(smem + m_bytes + (RAGGED_K ? LCB * wid : 0));
    const size_t off_wl = m_bytes + LCB * (RAGGED_K ? NW : 1);
    constexpr size_t WLB = RPOOL ? sizeof(QuadWave<PPL>) : sizeof(QuadLds<KP, PPL>);
    QuadLds<KP, PPL> *wlp = RPOOL ? nullptr : reinterpret_cast<QuadLds<KP, PPL> *>(smem + off_wl + WLB * wid);
    QuadWave<PPL> *qw = RPOOL ? reinterpret_cast<QuadWave<PPL> *>(smem + off_wl + WLB * wid) : nullptr;
    constexpr size_t HB = quad_hist_bytes<PPL>(HLDS);
    const size_t off_hist = off_wl + WLB * NW;
    double *hist = HLDS ? reinterpret_cast<double *>(smem + off_hist + HB * wid) : nullptr;
    const size_t off_rb = off_hist + HB * NW;
    QuadPool pool;
    pool.locks = reinterpret_cast<int *>(smem + off_rb);
    pool.slots = smem + off_rb + QUAD_POOL_LOCK_BYTES;
    pool.ns = pool_slots;
    pool.first = __builtin_amdgcn_readfirstlane(wid % (pool_slots > 0 ? pool_slots : 1));
    pool.slot_bytes = (unsigned)pool_slot_bytes;
    if (RPOOL) {
        // locks free; theta rows of every slot zero beyond P (resid_eval_q writes the first PPL x 64 only)
        for (int i = threadIdx.x; i < (int)(QUAD_POOL_LOCK_BYTES / sizeof(int)); i += NW * 64) pool.locks[i] = 0;
        for (int i = threadIdx.x; i < pool_slots * (PPL * W + W); i += NW * 64) {
            const int sl = i / (PPL * W + W), k = i - sl * (PPL * W + W);
            reinterpret_cast<QuadLds<KP, PPL> *>(pool.slots + (size_t)sl * pool.slot_bytes)->th[k] = 0.0;
        }
    }
    if (MLDS) {
        for (int i = threadIdx.x; i < P4 * PPL * W; i += NW * 64) Ml[i] = qa.Mg[i];
        __syncthreads();
    }
    const double *Mp = MLDS ? Ml : qa.Mg;
    const size_t rb_bytes = RLDS ? sizeof(double) * (size_t)NW * a.NTmax * W : 0;
    // QM_RAGGED_REG: the second column's tables of the two-columns-per-pass Gram build live in the wave's
    // staging rows (idle during that build; the launcher checks that they are large enough)
    double *Mown = (MMODE == QM_RAGGED || MMODE == QM_RAGGED_REG) ? qa.Mslot + ((size_t)blockIdx.x * NW + wid) * (size_t)P4 * PPL * W
                 : (MMODE == QM_RAGGED_LDS) ? reinterpret_cast<double *>(smem + off_rb + rb_bytes) +
                                               (size_t)wid * (PQ * PQ + W)
                                            : nullptr;
    constexpr int MRS = (MMODE == QM_RAGGED_LDS) ? PQ : W;
    if (MMODE == QM_RAGGED_LDS) {       // the spill-over reads past the last row must meet finite numbers
        Mown[PQ * PQ + lane] = 0.0;
        wave_sync();
    }
    // residual staging r[q][lane] of the running residual pass: in LDS when the launch found room
    // for NW x NTmax x 64 doubles, in registers (NTR), else in the global scratch (long series)
    double *rb = RLDS ? reinterpret_cast<double *>(smem + off_rb) + (size_t)wid * a.NTmax * W
                      : qa.rbuf + ((size_t)blockIdx.x * NW + wid) * a.NTmax * W;
    GramX *gx = (MMODE == QM_RAGGED_REG && RLDS) ? reinterpret_cast<GramX *>(rb) : nullptr;
    if (RPOOL) {
        for (int i = lane; i < PPL * W; i += W) qw->dl[i] = 0.0;
    } else {
        for (int i = lane; i < PPL * W + W; i += W) wlp->th[i] = 0.0;
    }
    wave_sync();

    for (;;) {
        // Every lane takes part (lane 0 adds 1, the others 0).  With `if (lane == 0) n = atomicAdd`
        // here, the compiler threaded lane 0's path from the lane-0-only epilogue stores of the
        // previous series straight into this block, and lanes 1..63 re-entered the loop (and
        // the readfirstlane below) without lane 0: an endless loop on the hardware.
        int n32 = atomicAdd(qa.counter, lane == 0 ? 1 : 0);
        n32 = __builtin_amdgcn_readfirstlane(n32);
        if (n32 >= a.N) break;
        const int64_t n = a.order ? (int64_t)a.order[n32] : (int64_t)n32;   // (cost hints: longest fits first)
        fit_one_quad<KP, PPL, PQ, RAGGED_K, MRS, HLDS,
                     ((quad_three_waves(MMODE, PPL, HLDS) || MMODE == QM_RAGGED_REG || MMODE == QM_GLOBAL_REG) ? 8 : 16),
                     MMODE == QM_RAGGED_REG || MMODE == QM_GLOBAL_REG, NTR, RPOOL>(qa, wlp, rb, Mp, Mown, n, lanec, hist, gx, qw, &pool);
    }
}


// ---------------------------------------------------------------------------------------
// eval-only kernel of the QUADRATIC form (parity tests: tsf_eval_quadratic): one residual pass at
// theta_ref makes it the reference point (s0, c = Z^T r_ref; M = Z^T Z from gram_build_kernel, copied
// to LDS as the fit kernel does), then ONE gram_eval_q at theta -- the template instance the 12-wave
// fit kernel calls for every trial point of its line searches (MB = 8, pipelined reads of M), so what
// a test sees here is the arithmetic of the headline kernel, per evaluation.
// One wave per workgroup; series n = blockIdx.x, + gridDim.x, ... (the staging rows of steps beyond
// NTR are per workgroup: qa.rbuf).
// ---------------------------------------------------------------------------------------
template <int KP, int PQ, int NTR>
__global__ __launch_bounds__(64) void eval_quad_kernel(QuadArgs qa, const double *theta_ref)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const FitArgs &a = qa.f;
    const int lane = lane_id();
    double *Ml = reinterpret_cast<double *>(smem);
    const size_t m_bytes = sizeof(double) * (size_t)PQ * W;
    double *lanec = reinterpret_cast<double *>(smem + m_bytes);
    QuadLds<KP, 1> &wl = *reinterpret_cast<QuadLds<KP, 1> *>(smem + m_bytes + quad_lanec_bytes<1>());
    for (int i = lane; i < PQ * W; i += W) Ml[i] = qa.Mg[i];
    for (int i = lane; i < 2 * W; i += W) wl.th[i] = 0.0;
    wave_sync();
    double *rb = qa.rbuf + (size_t)blockIdx.x * a.NTmax * W;
    for (int64_t n = blockIdx.x; n < a.N; n += gridDim.x) {
        SeriesView sv;
        make_view_q<KP, 1>(a, n, sv);
        LaneConst<1> lk;
        quad_const_table(lanec + 3 * W, a.opt, qa.recenter_ratio);
        lane_consts<1>(a.sp, sv, lanec, lk);
        lk.ct = lanec + 3 * W;
        double xr[1], gr[1], fr, s0, ztr[1];
        load_theta<1>(a, sv, n, theta_ref, xr);
        const bool bad_ref = resid_eval_q<KP, 1, NTR>(sv, wl, lk, rb, xr, fr, gr, s0, ztr);
        wl.ref[lane] = (lane == 2) ? 0.0 : xr[0];
        wl.cvec[lane] = ztr[0];
        wave_sync();
        double th[1], g[1], f, q2;
        load_theta<1>(a, sv, n, a.theta_in, th);
        const double mreg[1] = {0.0};
        const bool bad = gram_eval_q<1, PQ, W, 8, false, true, (TSF_QUAD_EXPTAB != 0)>(sv, lk, Ml, qa.P4, th, wl.ref, wl.cvec, s0, f, g, q2, wl.th, mreg);
        store_theta<1, false>(a, sv, n, g, a.grad_out);
        if (lane == 0) { a.fval[n] = f; a.status[n] = (bad || bad_ref) ? 1 : 0; }
        wave_sync();
    }
}

}  // namespace tsf
