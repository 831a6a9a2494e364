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

#include "../../include/tsf.h"
#include "tsf_detmath.h"

namespace tsf {

constexpr int W = 64;
constexpr int MAXH = 8;                 // max L-BFGS history
constexpr int NTAB = TSF_MAX_S + 4;     // per-changepoint tables

// Model description resident in device memory (built on the host by tsf_api).
struct DevSpec {
    int32_t growth, n_cp, K, Ka, KP, n_seas, n_extra, n_pairs;
    int32_t max_iter, history;
    double cp_range, tau, init_alpha, tol_obj, tol_rel_obj, tol_grad, tol_rel_grad, tol_param;
    int32_t inv_perm[TSF_MAX_P];        // original column -> internal column
    int32_t perm[TSF_MAX_P];            // internal column -> original column
    double prior[TSF_MAX_P];            // prior scale per internal column
    double pair_period[TSF_MAX_K];      // per (seasonality, harmonic) pair
    double pair_mult[TSF_MAX_K];        // 2.0 * (h + 1)
    int32_t pair_col[TSF_MAX_K];        // original column of the sin term
};

// Per-grid derived tables (one grid per call for aligned panels, one per series for ragged).
struct GridTab {
    tsf_grid_info info;
    int32_t Lj[NTAB];                   // chunk holding the first row with t >= t_change[j]
};

struct SeriesTab {
    double y_scale, cap, k0, m0, floor_;
    int32_t status0;                    // 0 ok, TSF_ST_TOO_FEW / TSF_ST_CAP / TSF_ST_CONSTANT
    int32_t pad_;
};

// ---------------------------------------------------------------------------------------
// cross-lane helpers (wave64)
// ---------------------------------------------------------------------------------------

__device__ __forceinline__ double bfly_sum(double v)
{
#pragma unroll
    for (int off = 1; off < W; off <<= 1) v = v + __shfl_xor(v, off, W);
    return v;
}

__device__ __forceinline__ double bcast(double v, int lane) { return __shfl(v, lane, W); }

__device__ __forceinline__ double uniform_f64(double v)
{
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// dot product over the parameter axis: slot 0 product, slot 1 fma'd in, then butterfly
template <int PPL>
__device__ __forceinline__ double pdot(const double (&a)[PPL], const double (&b)[PPL])
{
    double part = a[0] * b[0];
    if (PPL == 2) part = __builtin_fma(a[PPL - 1], b[PPL - 1], part);
    return bfly_sum(part);
}

__device__ __forceinline__ bool finite_f64(double x)
{
    return (x - x) == 0.0;
}

}  // namespace tsf
