// tsf_interval_kernels.h -- uncertainty intervals of the forecast (yhat_lower / yhat_upper):
// fbprophet's Prophet.predict_uncertainty, which the reference runs inside model.predict(future_df)
// (/root/reference/src/jobs/prophet_scorer.py:70) and then drops (:86).  SURVEY.md 8f-3.
//
// fbprophet draws from numpy's unseeded global generator, so there is no output to match; parity
// is DEFINED by oracle cn_predict_intervals: a counter-based generator keyed by (seed, series key,
// sample, stream), deterministic transcendentals, the sampled changepoint times generated already
// sorted.  These kernels consume the same streams in the same order: bit-identical to the oracle.
//   interval_sample_kernel      one thread per (series, sample): sweeps the future rows, writes
//                               samples[series][row][sample]
//   interval_percentile_kernel  one workgroup per (series, row): bitonic sort of the samples in LDS,
//                               the two percentiles by linear interpolation (np.nanpercentile)
// Non-template __global__ functions: include from exactly one translation unit (tsf_api.hip).
#pragma once
#include "tsf_common.h"

namespace tsf {

__device__ __forceinline__ uint64_t iv_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t iv_key(uint64_t seed, uint64_t series_key, uint64_t sample, uint64_t stream)
{
    return iv_mix64(iv_mix64(iv_mix64(iv_mix64(seed) ^ series_key) ^ sample) ^ stream);
}
__device__ __forceinline__ double iv_u01(uint64_t key, uint64_t ctr)        // in (0, 1)
{
    const uint64_t x = iv_mix64(key + 0x9E3779B97F4A7C15ULL * ctr);
    return ((double)(x >> 11) + 0.5) * 1.1102230246251565e-16;
}
__device__ __forceinline__ int iv_poisson(double lam, uint64_t key, uint64_t &ctr)
{
    int n = 0;
    double rest = lam;
    while (rest > 0.0 && n < 100000) {
        const double piece = rest < 8.0 ? rest : 8.0;
        const double L = dm_exp(-piece);
        double p = 1.0;
        int k = 0;
        do { p = p * iv_u01(key, ctr++); ++k; } while (p > L);
        n += k - 1;
        rest = rest - piece;
    }
    return n;
}

struct IntervalArgs {
    const DevSpec *sp;
    int64_t n0, n_chunk;        // series [n0, n0 + n_chunk) of the call
    int H, theta_stride, n_grids, NS;
    const double *theta, *y_scale;
    const tsf_grid_info *grid;
    const double *floor_, *cap;
    const double *t, *xa, *opm; // [N][H] from predict_kernel: scaled time, additive term * y_scale, 1 + multiplicative term
    const int64_t *series_key;  // [N] or nullptr (then the series index)
    uint64_t seed;
    double lo_frac, hi_frac;    // (1 - width) / 2, (1 + width) / 2
    double *samples;            // [n_chunk][H][NS]
    double *lower, *upper;      // [N][H]
};

__global__ __launch_bounds__(256) void interval_sample_kernel(IntervalArgs a)
{
    const int64_t nl = blockIdx.x, n = a.n0 + nl;
    const int s = (int)(blockIdx.y * blockDim.x + threadIdx.x);
    if (nl >= a.n_chunk || s >= a.NS) return;
    const DevSpec *sp = a.sp;
    const tsf_grid_info &gi = a.grid[a.n_grids == 1 ? 0 : n];
    const int S = gi.S, n_cp = sp->n_cp, H = a.H;
    const double *th = a.theta + (size_t)n * a.theta_stride;
    const double *delta = th + 3;
    (void)n_cp;
    const double ys = a.y_scale[n];
    const double fl = (sp->growth == TSF_GROWTH_LOGISTIC && a.floor_) ? a.floor_[n] : 0.0;
    const double cap_sc = (sp->growth == TSF_GROWTH_LOGISTIC) ? (a.cap[n] - fl) / ys : 0.0;
    const double sigma = dm_exp(th[2]);
    const double *t = a.t + (size_t)n * H, *xa = a.xa + (size_t)n * H, *opm = a.opm + (size_t)n * H;
    double Tm = -__builtin_huge_val();
    for (int h = 0; h < H; ++h) if (t[h] > Tm) Tm = t[h];
    const int S_cp = S > 0 ? S : 1;
    double lam_sum = 0.0;
    for (int j = 0; j < S; ++j) lam_sum = lam_sum + __builtin_fabs(delta[j]);
    const double lambda_ = lam_sum / (double)S_cp + 1e-8;
    const double rate = (Tm > 1.0) ? (double)S_cp * (Tm - 1.0) : 0.0;
    const uint64_t skey = a.series_key ? (uint64_t)a.series_key[n] : (uint64_t)n;
    const uint64_t kcp = iv_key(a.seed, skey, (uint64_t)s, 0), knz = iv_key(a.seed, skey, (uint64_t)s, 1);
    const double INF = __builtin_huge_val();
    double wk = 0.0, wm = 0.0, u_prev = 0.0, next_t = INF, next_delta = 0.0, t_last = -INF;
    int ih = 0, inew = 0, n_new = 0;
    uint64_t ctr = 0;
    double *out = a.samples + (size_t)nl * H * a.NS + s;
    for (int h = 0; h < H; ++h) {
        const double th_ = t[h];
        if (h == 0 || th_ < t_last) {       // (re)start the sweep: the stream is replayed from 0
            wk = th[0]; wm = th[1]; ih = 0; inew = 0; ctr = 0; u_prev = 0.0;
            n_new = (rate > 0.0) ? iv_poisson(rate, kcp, ctr) : 0;
            next_t = INF;
        }
        t_last = th_;
        while (ih < S && th_ >= gi.t_change[ih]) {
            const double dj = delta[ih], kn = wk + dj, tc = gi.t_change[ih];
            if (sp->growth == TSF_GROWTH_LINEAR) wm = wm + ((-tc) * dj);
            else wm = wm + (tc - wm) * (1.0 - wk / kn);
            wk = kn; ++ih;
        }
        for (;;) {
            if (next_t == INF && inew < n_new) {
                const double v = iv_u01(kcp, ctr++);
                const double rem = (double)(n_new - inew);
                const double pw = dm_exp(dm_log(v) / rem);
                u_prev = 1.0 - (1.0 - u_prev) * pw;
                next_t = 1.0 + u_prev * (Tm - 1.0);
                const double ul = iv_u01(kcp, ctr++);
                next_delta = (ul < 0.5) ? lambda_ * dm_log(2.0 * ul) : -(lambda_ * dm_log(2.0 * (1.0 - ul)));
            }
            if (!(next_t <= th_)) break;
            const double dj = next_delta, kn = wk + dj;
            if (sp->growth == TSF_GROWTH_LINEAR) wm = wm + ((-next_t) * dj);
            else wm = wm + (next_t - wm) * (1.0 - wk / kn);
            wk = kn; ++inew; next_t = INF;
        }
        double gtr;
        if (sp->growth == TSF_GROWTH_LINEAR) gtr = __builtin_fma(wk, th_, wm);
        else gtr = cap_sc * (1.0 / (1.0 + dm_exp(-(wk * (th_ - wm)))));
        const double trend = gtr * ys + fl;
        const double u1 = iv_u01(knz, 2 * (uint64_t)h), u2 = iv_u01(knz, 2 * (uint64_t)h + 1);
        double sn, cs;
        dm_sincos(6.283185307179586 * u2, sn, cs);
        const double z = __builtin_sqrt(-2.0 * dm_log(u1)) * cs;
        out[(size_t)h * a.NS] = trend * opm[h] + xa[h] + (z * sigma) * ys;
    }
}

// NSP: NS rounded up to a power of two (<= 4096), the padding sorts to the end as +inf
__global__ __launch_bounds__(256) void interval_percentile_kernel(IntervalArgs a, int NSP)
{
    extern __shared__ __align__(16) unsigned char iv_smem[];
    double *v = reinterpret_cast<double *>(iv_smem);
    const int64_t nl = blockIdx.x / a.H;
    const int h = (int)(blockIdx.x - nl * a.H);
    const double *src = a.samples + ((size_t)nl * a.H + h) * a.NS;
    for (int i = threadIdx.x; i < NSP; i += blockDim.x) v[i] = (i < a.NS) ? src[i] : __builtin_huge_val();
    __syncthreads();
    for (int k = 2; k <= NSP; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < NSP; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool asc = (i & k) == 0;
                    const double x = v[i], y = v[ixj];
                    if ((x > y) == asc) { v[i] = y; v[ixj] = x; }
                }
            }
            __syncthreads();
        }
    if (threadIdx.x < 2) {
        const double pos = (threadIdx.x ? a.hi_frac : a.lo_frac) * (double)(a.NS - 1);
        int lo = (int)__builtin_floor(pos);
        if (lo > a.NS - 1) lo = a.NS - 1;
        const int hi = lo + 1 < a.NS ? lo + 1 : a.NS - 1;
        const double val = v[lo] + (v[hi] - v[lo]) * (pos - (double)lo);
        double *dst = threadIdx.x ? a.upper : a.lower;
        dst[(size_t)(a.n0 + nl) * a.H + h] = val;
    }
}

}  // namespace tsf
