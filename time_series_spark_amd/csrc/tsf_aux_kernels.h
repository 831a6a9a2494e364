// tsf_aux_kernels.h -- setup (scaling, changepoints, design matrix), predict and self-test
// kernels.  Non-template __global__ functions: include from exactly one translation unit
// (tsf_api.hip).
#pragma once
#include "tsf_common.h"

namespace tsf {

// ---------------------------------------------------------------------------------------
// setup kernels
// ---------------------------------------------------------------------------------------

// Sparse indicator columns of a 64-column design table (SP_* in tsf_fit_kernels.h; included before this file by
// tsf_api.hip).  One workgroup of 64 per grid, lane = chunk as in the fit kernels.  Writes the lanes' entry words
// meta[grid][SP_M][64] and the columns' fold programs prog[grid][SP_MAXC]; sets *bad if the grid does not qualify.
__global__ __launch_bounds__(64) void sparse_extra_kernel(const DevSpec *__restrict__ sp, const GridTab *__restrict__ gtab,
                                                          const double *__restrict__ Xw_all, int NTmax,
                                                          uint32_t *__restrict__ meta_all, unsigned long long *__restrict__ prog_all,
                                                          int *__restrict__ bad)
{
    const int g = blockIdx.x, lane = threadIdx.x;
    __shared__ unsigned long long mask[SP_MAXC];
    const int Ks = sp->K - SP_DENSE;
    const GridTab &gt = gtab[g];
    const int NT = gt.info.NT, T = gt.info.T;
    const double *X = Xw_all + (size_t)g * NTmax * 64 * W;
    uint32_t *meta = meta_all + (size_t)g * SP_M * W;
    unsigned long long *prog = prog_all + (size_t)g * SP_MAXC;
    bool ok = Ks > 0 && Ks <= SP_MAXC && NT <= SP_MAX_NT && sp->KP == 64;
    if (lane < SP_MAXC) mask[lane] = 0ull;
    for (int e = 0; e < SP_M; ++e) meta[e * W + lane] = SP_END;
    __syncthreads();
    int rows = T - lane * NT;
    rows = rows < 0 ? 0 : (rows > NT ? NT : rows);
    int ne = 0;
    unsigned long long seen = 0ull;
    // last row first (the order in which the fit kernel's row loop meets them), columns ascending inside a row: the
    // ones of a row then stand in the order of its fma chain
    for (int q = rows - 1; q >= 0 && ok; --q) {
        for (int c = 0; c < Ks; ++c) {
            const double v = X[((size_t)q * 64 + SP_DENSE + c) * W + lane];
            if (v == 0.0) continue;
            if (v != 1.0 || ((seen >> c) & 1ull) || ne == SP_M) { ok = false; break; }
            seen |= 1ull << c;
            meta[ne * W + lane] = (unsigned)q | ((unsigned)c << 7);
            ++ne;
            atomicOr(&mask[c], 1ull << lane);
        }
    }
    ok = __syncthreads_and(ok ? 1 : 0) != 0;
    if (ok && lane < Ks && __popcll(mask[lane]) > SP_E) ok = false;
    ok = __syncthreads_and(ok ? 1 : 0) != 0;
    if (!ok) {
        if (lane == 0) atomicExch(bad, 1);
        return;
    }
    // the slot of every entry: column * SP_E + rank of this lane among the column's lanes
    for (int e = 0; e < ne; ++e) {
        const unsigned m = meta[e * W + lane];
        const int c = (int)((m >> 7) & 63u);
        const int rank = __popcll(mask[c] & ((1ull << lane) - 1ull));
        meta[e * W + lane] = m | ((unsigned)rank << 13);
    }
    // the fold program of column `lane`: the nodes are the column's lanes (slot = rank); the reduction network of
    // column_sums pairs lanes that differ in bit 5, then 4, 0, 1, 2, 3 -- two nodes merge at the stage after which they
    // agree on every bit not yet used
    if (lane < SP_MAXC) {
        unsigned long long pg = 0ull;
        if (lane < Ks) {
            const unsigned long long mk = mask[lane];
            const int k = __popcll(mk);
            int L[SP_E];
            bool alive[SP_E];
            {
                unsigned long long r = mk;
                for (int i = 0; i < SP_E; ++i) {
                    alive[i] = i < k;
                    L[i] = 0;
                    if (i < k) { L[i] = __ffsll((long long)r) - 1; r &= r - 1ull; }
                }
            }
            int nm = 0;
            unsigned used = 0u;
            const int order[6] = {5, 4, 0, 1, 2, 3};
            for (int s = 0; s < 6; ++s) {
                used |= 1u << order[s];
                for (int i = 0; i < SP_E; ++i) {
                    if (!alive[i]) continue;
                    for (int j = i + 1; j < SP_E; ++j) {
                        if (!alive[j]) continue;
                        if (((unsigned)L[i] & ~used) == ((unsigned)L[j] & ~used)) {
                            pg |= ((unsigned long long)i | ((unsigned long long)j << 3)) << (10 + 6 * nm);
                            ++nm;
                            alive[j] = false;
                        }
                    }
                }
            }
            pg |= (unsigned long long)nm | (0ull << 3) | ((unsigned long long)k << 6);
        }
        prog[lane] = pg;
    }
}

// One block per grid.  Derives scaled time, changepoints, segment indices and the design
// matrix (fbprophet setup_dataframe / set_changepoints / make_all_seasonality_features).
// Xw must be zero-filled by the caller (padding columns / rows).
__global__ void setup_grid_kernel(const DevSpec *__restrict__ sp, int n_grids,
                                  const int64_t *__restrict__ offsets, int T_aligned,
                                  const int64_t *__restrict__ ds_all,
                                  const double *__restrict__ extra, int64_t extra_stride,
                                  int NTmax, GridTab *__restrict__ gtab,
                                  double *__restrict__ tw_all, uint16_t *__restrict__ cw_all,
                                  double *__restrict__ Xw_all, int32_t *__restrict__ uw_all,
                                  int64_t lat_base, int64_t lat_step,
                                  const int64_t *__restrict__ grid_rows = nullptr,
                                  double *__restrict__ Bw_all = nullptr)
{
    const int g = blockIdx.x;
    if (g >= n_grids) return;
    // grid_rows (ragged panels whose series share timestamp vectors: FitArgs::grid_of): [n_grids][2] first row and
    // row count of the vector of grid g; else the rows of series g
    const int64_t row0 = grid_rows ? grid_rows[2 * g] : (offsets ? offsets[g] : 0);
    const int T = grid_rows ? (int)grid_rows[2 * g + 1] : (offsets ? (int)(offsets[g + 1] - offsets[g]) : T_aligned);
    const int64_t *ds = ds_all + row0;
    GridTab &gt = gtab[g];
    double *tw = tw_all + (size_t)g * NTmax * W;
    uint16_t *cw = cw_all + (size_t)g * NTmax * W;
    double *Xw = Xw_all + (size_t)g * NTmax * sp->KP * W;
    int32_t *uw = uw_all + (size_t)g * NTmax * W;     // only touched when lat_step > 0
    const int KP = sp->KP;
    __shared__ double tch[NTAB];
    __shared__ int S_sh;
    const int NT = (T + W - 1) / W;
    if (T < 2) {
        if (threadIdx.x == 0) {
            gt.info.T = T; gt.info.S = 0; gt.info.NT = NT > 0 ? NT : 1; gt.info.i1 = 0;
            gt.info.start_ns = T > 0 ? ds[0] : 0; gt.info.t_scale_ns = 0;
        }
        return;
    }
    const int64_t start = ds[0];
    const double tsc = (double)(ds[T - 1] - ds[0]);
    int hist = (int)__builtin_floor((double)T * sp->cp_range);
    int S = sp->n_cp;
    if (S + 1 > hist) S = hist - 1;
    if (S < 0) S = 0;
    const int S_out = S;
    if (threadIdx.x == 0) S_sh = S;
    const double step = (S > 0) ? (double)(hist - 1) / (double)S : 0.0;
    if (S_out == 0) {
        // no changepoints: the dummy changepoint at t = 0 (see GridTab::S_fit)
        S = 1;
        if (threadIdx.x == 0) { tch[0] = 0.0; gt.Lj[0] = 0; gt.info.t_change[0] = 0.0; }
    }
    for (int j = threadIdx.x; j < S_out; j += blockDim.x) {
        const double v = (j + 1 == S) ? (double)(hist - 1) : (double)(j + 1) * step;
        const int idx = (int)__builtin_rint(v);
        tch[j] = (double)(ds[idx] - start) / tsc;
        int fj = idx;
        while (fj > 0 && ds[fj - 1] == ds[idx]) --fj;
        gt.Lj[j] = fj / NT;
        gt.info.t_change[j] = tch[j];
    }
    if (threadIdx.x == 0) {
        int i1 = T - 1;
        while (i1 > 0 && ds[i1 - 1] == ds[T - 1]) --i1;
        gt.info.start_ns = start; gt.info.t_scale_ns = ds[T - 1] - ds[0];
        gt.info.T = T; gt.info.S = S_out; gt.info.i1 = i1; gt.info.NT = NT;
        gt.S_fit = S;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        const int L = i / NT, q = i - L * NT;
        const double ti = (double)(ds[i] - start) / tsc;
        tw[q * W + L] = ti;
        int c = 0;
        while (c < S && ti >= tch[c]) ++c;
        int cp = 0;
        if (i > 0) {
            const double tp = (double)(ds[i - 1] - start) / tsc;
            while (cp < S && tp >= tch[cp]) ++cp;
        }
        cw[q * W + L] = (uint16_t)(c | (cp << 8));
        if (lat_step > 0) uw[q * W + L] = (int32_t)((ds[i] - lat_base) / lat_step);
    }
    if (lat_step > 0) return;       // design rows come from the shared lattice table (setup_lattice_kernel)
    // Fourier columns: one (row, seasonality) per work item -- the base pair from dm_sincos, the harmonics by the
    // recurrence (fourier_harmonics, tsf_common.h).  Bw (where given): the base pairs alone, [step][seasonality][lane][2],
    // the table of the residual-form kernel that expands the harmonics in registers (eval_fg HARM).
    const int n_seas = sp->n_seas;
    double *Bw = Bw_all ? Bw_all + (size_t)g * NTmax * n_seas * 2 * W : nullptr;
    for (int w = threadIdx.x; w < T * n_seas; w += blockDim.x) {
        const int i = w / n_seas, se = w - i * n_seas;
        const int L = i / NT, q = i - L * NT;
        double s1, c1;
        dm_sincos(fourier_base_arg(ds[i], sp->seas_period[se]), s1, c1);
        if (Bw) { Bw[(((size_t)q * n_seas + se) * W + L) * 2] = s1; Bw[(((size_t)q * n_seas + se) * W + L) * 2 + 1] = c1; }
        const int col0 = sp->seas_col[se];
        fourier_harmonics(s1, c1, sp->seas_order[se], [&](int h, double sv, double cv) {
            Xw[((size_t)q * KP + sp->inv_perm[col0 + 2 * (h - 1)]) * W + L] = sv;
            Xw[((size_t)q * KP + sp->inv_perm[col0 + 2 * (h - 1) + 1]) * W + L] = cv;
        });
    }
    const int nf = sp->K - sp->n_extra;
    for (int w = threadIdx.x; w < T * sp->n_extra; w += blockDim.x) {
        const int e = w / T, i = w - e * T;
        const int L = i / NT, q = i - L * NT;
        Xw[((size_t)q * KP + sp->inv_perm[nf + e]) * W + L] = extra[(size_t)e * extra_stride + row0 + i];
    }
}

// Design rows of a timestamp lattice base + u*step, u < U: Xu[u][KP] in internal column order,
// same arithmetic as setup_grid_kernel (explicit columns are not functions of the timestamp:
// lattice tables are only used when the model has none).  Xu must be zero-filled by the caller.
// Bu (where given): the base pairs alone, [u][seasonality][2] (FitArgs::Bu).
__global__ void setup_lattice_kernel(const DevSpec *__restrict__ sp, int64_t U, int64_t lat_base,
                                     int64_t lat_step, double *__restrict__ Xu, double *__restrict__ Bu)
{
    const int n_seas = sp->n_seas, KP = sp->KP;
    const int64_t total = U * n_seas;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t u = w / n_seas;
        const int se = (int)(w - u * n_seas);
        const int64_t ts = lat_base + u * lat_step;
        double s1, c1;
        dm_sincos(fourier_base_arg(ts, sp->seas_period[se]), s1, c1);
        if (Bu) { Bu[((size_t)u * n_seas + se) * 2] = s1; Bu[((size_t)u * n_seas + se) * 2 + 1] = c1; }
        const int col0 = sp->seas_col[se];
        fourier_harmonics(s1, c1, sp->seas_order[se], [&](int h, double sv, double cv) {
            Xu[(size_t)u * KP + sp->inv_perm[col0 + 2 * (h - 1)]] = sv;
            Xu[(size_t)u * KP + sp->inv_perm[col0 + 2 * (h - 1) + 1]] = cv;
        });
    }
}

// One wave per series: y scaling (initialize_scales), step-major copy of scaled y, growth
// init (linear_growth_init / logistic_growth_init).
__global__ __launch_bounds__(64) void setup_series_kernel(
    const DevSpec *__restrict__ sp, int64_t N, const int64_t *__restrict__ offsets, int T_aligned,
    const int64_t *__restrict__ ds_all, const void *__restrict__ y_all, int y_dtype,
    const double *__restrict__ floor_in, const double *__restrict__ cap_in, int NTmax,
    const GridTab *__restrict__ gtab, int aligned, SeriesTab *__restrict__ stab,
    double *__restrict__ yw_all, const int32_t *__restrict__ grid_of = nullptr)
{
    const int64_t n = blockIdx.x;
    if (n >= N) return;
    const int lane = threadIdx.x;
    const GridTab &gt = gtab[aligned ? 0 : (grid_of ? (int64_t)grid_of[n] : n)];
    const int T = gt.info.T, NT = gt.info.NT;
    const int64_t row0 = offsets ? offsets[n] : n * (int64_t)T_aligned;
    const int64_t *ds = ds_all + (offsets ? offsets[n] : 0);
    double *yw = yw_all ? yw_all + (size_t)n * NTmax * W : nullptr;      // null: the fit reads the caller's rows itself (quadratic form)
    SeriesTab &st = stab[n];
    const double fl = (sp->growth == TSF_GROWTH_LOGISTIC && floor_in) ? floor_in[n] : 0.0;
    const double capv = cap_in ? cap_in[n] : 0.0;
    if (T < 2) {
        if (lane == 0) { st.status0 = TSF_ST_TOO_FEW; st.y_scale = 1.0; st.cap = 0; st.k0 = 0; st.m0 = 0; st.floor_ = fl; }
        return;
    }
    double amax = 0.0, ymin = __builtin_huge_val(), ymax = -__builtin_huge_val();
    // Series of up to 16 x 64 rows keep their rows in registers between the scale pass and the scaling pass (round 6):
    // y is read ONCE -- the kernel used to read the panel twice, 114 of the step's 289 MB on the headline panel.
    constexpr int HOLD = 16;
    const bool hold = T <= HOLD * W;
    double held[HOLD];
    if (hold) {
#pragma unroll
        for (int k = 0; k < HOLD; ++k) {
            const int i = lane + k * W;
            held[k] = 0.0;
            if (i < T) {
                const double v = load_y(y_all, y_dtype, row0 + i);
                held[k] = v;
                amax = __builtin_fmax(amax, __builtin_fabs(v - fl));
                ymin = __builtin_fmin(ymin, v);
                ymax = __builtin_fmax(ymax, v);
            }
        }
    } else {
        for (int i = lane; i < T; i += W) {
            const double v = load_y(y_all, y_dtype, row0 + i);
            amax = __builtin_fmax(amax, __builtin_fabs(v - fl));
            ymin = __builtin_fmin(ymin, v);
            ymax = __builtin_fmax(ymax, v);
        }
    }
#pragma unroll
    for (int off = 1; off < W; off <<= 1) {
        amax = __builtin_fmax(amax, __shfl_xor(amax, off, W));
        ymin = __builtin_fmin(ymin, __shfl_xor(ymin, off, W));
        ymax = __builtin_fmax(ymax, __shfl_xor(ymax, off, W));
    }
    const double ys = (amax == 0.0) ? 1.0 : amax;
    if (!yw) {
        // no scaled copy wanted
    } else if (hold) {
#pragma unroll
        for (int k = 0; k < HOLD; ++k) {
            const int i = lane + k * W;
            if (i < T) {
                const int L = i / NT, q = i - L * NT;
                yw[q * W + L] = (held[k] - fl) / ys;
            }
        }
    } else {
        for (int i = lane; i < T; i += W) {
            const int L = i / NT, q = i - L * NT;
            yw[q * W + L] = (load_y(y_all, y_dtype, row0 + i) - fl) / ys;
        }
    }
    if (lane == 0) {
        int status0 = 0;
        double capsc = 0.0, k0 = 0.0, m0 = 0.0;
        const int i0 = 0, i1 = gt.info.i1;
        const double tsc = (double)gt.info.t_scale_ns;
        const double t0 = (double)(ds[i0] - gt.info.start_ns) / tsc;
        const double t1 = (double)(ds[i1] - gt.info.start_ns) / tsc;
        const double y0 = (load_y(y_all, y_dtype, row0 + i0) - fl) / ys;
        const double y1 = (load_y(y_all, y_dtype, row0 + i1) - fl) / ys;
        const double Td = t1 - t0;
        if (sp->growth == TSF_GROWTH_LINEAR) {
            k0 = (y1 - y0) / Td;
            m0 = y0 - k0 * t0;
            if (ymin == ymax) status0 = TSF_ST_CONSTANT;
        } else {
            if (capv <= fl) {
                status0 = TSF_ST_CAP;
            } else {
                capsc = (capv - fl) / ys;
                const double C0 = capsc, C1 = capsc;
                const double yy0 = __builtin_fmax(0.01 * C0, __builtin_fmin(0.99 * C0, y0));
                const double yy1 = __builtin_fmax(0.01 * C1, __builtin_fmin(0.99 * C1, y1));
                double r0 = C0 / yy0;
                const double r1 = C1 / yy1;
                if (__builtin_fabs(r0 - r1) <= 0.01) r0 = 1.05 * r0;
                const double L0 = dm_log(r0 - 1.0), L1 = dm_log(r1 - 1.0);
                m0 = L0 * Td / (L0 - L1);
                k0 = (L0 - L1) / Td;
            }
        }
        st.status0 = status0; st.y_scale = ys; st.cap = capsc; st.k0 = k0; st.m0 = m0; st.floor_ = fl;
    }
}

// ---------------------------------------------------------------------------------------
// predict: one wavefront per series, lanes over the horizon
// ---------------------------------------------------------------------------------------
// (Prophet.predict as the reference calls it, /root/reference/src/jobs/prophet_scorer.py:64-84.)
// Round 2 ran one thread per (series, step) with the design row in a per-thread array indexed
// through the column permutation: the array went to scratch (390 MB of traffic for 10.8 MB of
// output) and every thread recomputed the same 13 sincos of a future grid all series share.  Now:
//   * a shared future grid gets ONE design table Xf[column][step] (future_design_kernel, H x K values,
//     L2-resident), read coalesced over the steps;
//   * per-series future grids compute their Fourier terms in place, column by column in the order of
//     the chain (no array);
//   * the trend's changepoint recurrence (a division per step for logistic growth) runs once per
//     series, not once per (series, step): the wave walks it and parks (ks, mc) per segment in LDS;
//   * the coefficients of a series are wave-uniform.
// Same operations in the same order per forecast as before: bit-identical to oracle cn_predict.

struct PredictArgs {
    const DevSpec *sp;
    int64_t N;
    int H, theta_stride, n_grids, shared_future;
    const double *theta, *y_scale;
    const tsf_grid_info *grid;
    const int64_t *ds_future;
    const double *floor_, *cap, *extra_future;
    double *yhat;
    int32_t *yhat_int;
    // optional per-row pieces for the interval kernels (tsf_interval_kernels.h): scaled time,
    // additive term * y_scale, 1 + multiplicative term
    double *t_out, *xa_out, *opm_out;
    const double *Xf;           // shared future grid: [K][H] design values in ORIGINAL column order
};

// Fourier columns of a shared future grid, original column order: Xf[col][h]
__global__ void future_design_kernel(const DevSpec *sp, int H, const int64_t *ds_future, double *Xf)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= H * sp->n_seas) return;
    const int se = i / H, h = i - se * H;
    double s1, c1;
    dm_sincos(fourier_base_arg(ds_future[h], sp->seas_period[se]), s1, c1);
    const int col0 = sp->seas_col[se];
    fourier_harmonics(s1, c1, sp->seas_order[se], [&](int hh, double sv, double cv) {
        Xf[(size_t)(col0 + 2 * (hh - 1)) * H + h] = sv;
        Xf[(size_t)(col0 + 2 * (hh - 1) + 1) * H + h] = cv;
    });
}

constexpr int PREDICT_WAVES = 4;        // series per workgroup

__global__ __launch_bounds__(PREDICT_WAVES * 64) void predict_kernel(PredictArgs a)
{
    __shared__ double seg_ks[PREDICT_WAVES][TSF_MAX_S + 4], seg_mc[PREDICT_WAVES][TSF_MAX_S + 4];
    const int wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * PREDICT_WAVES + wid;
    if (n >= a.N) return;
    const DevSpec *sp = a.sp;
    const tsf_grid_info &gi = a.grid[a.n_grids == 1 ? 0 : n];
    const int S = gi.S, K = sp->K, Ka = sp->Ka, n_cp = sp->n_cp, H = a.H;
    const double *th = a.theta + (size_t)n * a.theta_stride;
    const double *delta = th + 3, *beta = th + 3 + n_cp;
    const double ys = a.y_scale[n];
    const double fl = (sp->growth == TSF_GROWTH_LOGISTIC && a.floor_) ? a.floor_[n] : 0.0;
    const double fl_clamp = a.floor_ ? a.floor_[n] : 0.0;
    const double capsc = (sp->growth == TSF_GROWTH_LOGISTIC) ? (a.cap[n] - fl) / ys : 0.0;
    // slope / offset of every trend segment: the sequential recurrence, once per series
    // (the changepoints' deltas and times in lanes first: one load each instead of S dependent ones per wave)
    double dl[(TSF_MAX_S + W - 1) / W], tl[(TSF_MAX_S + W - 1) / W];
    {
        double ks = th[0], mc = th[1];
        if (lane == 0) { seg_ks[wid][0] = ks; seg_mc[wid][0] = mc; }
#pragma unroll
        for (int s = 0; s < (TSF_MAX_S + W - 1) / W; ++s) {
            const int c = lane + s * W;
            dl[s] = c < S ? delta[c] : 0.0;
            tl[s] = c < S ? gi.t_change[c] : 0.0;
        }
        for (int c = 0; c < S; ++c) {
            double dj = 0.0, tcj = 0.0;
#pragma unroll
            for (int s = 0; s < (TSF_MAX_S + W - 1) / W; ++s)
                if ((c >> 6) == s) { dj = readlane_f64(dl[s], c & 63); tcj = readlane_f64(tl[s], c & 63); }
            const double ksn = ks + dj;
            if (sp->growth == TSF_GROWTH_LINEAR) {
                mc = mc + ((-tcj) * dj);
            } else {
                const double gamma = (tcj - mc) * (1.0 - ks / ksn);
                mc = mc + gamma;
            }
            ks = ksn;
            if (lane == 0) { seg_ks[wid][c + 1] = ks; seg_mc[wid][c + 1] = mc; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nf = K - sp->n_extra;
    // (the coefficients in chain order in lanes: bl[j mod 64 of slot j / 64] = beta[perm[j]])
    double bl[(TSF_MAX_P + W - 1) / W];
#pragma unroll
    for (int s = 0; s < (TSF_MAX_P + W - 1) / W; ++s) {
        const int j = lane + s * W;
        bl[s] = j < K ? beta[sp->perm[j]] : 0.0;
    }
    for (int h = lane; h < H; h += 64) {
        const int64_t gid = n * (int64_t)H + h;
        const int64_t dsv = a.ds_future[a.shared_future ? h : gid];
        const double t = (double)(dsv - gi.start_ns) / (double)gi.t_scale_ns;
        // (the segment of t: the changepoint times ascend, so the first one beyond t is their count up to t -- taken from
        // the lanes that hold them instead of a dependent load per step)
        int c = 0;
        for (int j = 0; j < S; ++j) {
            double tcj = 0.0;
#pragma unroll
            for (int s = 0; s < (TSF_MAX_S + W - 1) / W; ++s)
                if ((j >> 6) == s) tcj = readlane_f64(tl[s], j & 63);
            c += (c == j && t >= tcj) ? 1 : 0;
        }
        const double ks = seg_ks[wid][c], mc = seg_mc[wid][c];
        // X.beta in the order of the chain: additive columns in original order, then the
        // multiplicative ones (internal column j = original column perm[j])
        double xa = 0.0, xm = 0.0;
        if (a.Xf) {
            for (int j = 0; j < K; ++j) {
                const int col = sp->perm[j];
                const double xv = (col < nf) ? a.Xf[(size_t)col * H + h]
                                             : a.extra_future[a.shared_future ? (size_t)(col - nf) * H + h
                                                                              : ((size_t)n * sp->n_extra + (col - nf)) * H + h];
                double bj = 0.0;
#pragma unroll
                for (int s = 0; s < (TSF_MAX_P + W - 1) / W; ++s)
                    if ((j >> 6) == s) bj = readlane_f64(bl[s], j & 63);
                if (j < Ka) xa = __builtin_fma(xv, bj, xa);
                else xm = __builtin_fma(xv, bj, xm);
            }
        } else {
            for (int pass = 0; pass < 2; ++pass) {
                double acc = 0.0;
                for (int se = 0; se < sp->n_seas; ++se) {
                    const int col0 = sp->seas_col[se];
                    if ((sp->inv_perm[col0] < Ka) != (pass == 0)) continue;       // (a seasonality's columns share one mode)
                    double s1, c1;
                    dm_sincos(fourier_base_arg(dsv, sp->seas_period[se]), s1, c1);
                    fourier_harmonics(s1, c1, sp->seas_order[se], [&](int hh, double sv, double cv) {
                        const int col = col0 + 2 * (hh - 1);
                        acc = __builtin_fma(sv, beta[col], acc);
                        acc = __builtin_fma(cv, beta[col + 1], acc);
                    });
                }
                for (int e = 0; e < sp->n_extra; ++e) {
                    const int col = nf + e;
                    if ((sp->inv_perm[col] < Ka) != (pass == 0)) continue;
                    const size_t off = a.shared_future ? (size_t)e * H + h : ((size_t)n * sp->n_extra + e) * H + h;
                    acc = __builtin_fma(a.extra_future[off], beta[col], acc);
                }
                if (pass == 0) xa = acc; else xm = acc;
            }
        }
        double gtr;
        if (sp->growth == TSF_GROWTH_LINEAR) {
            gtr = __builtin_fma(ks, t, mc);
        } else {
            const double z = ks * (t - mc);
            gtr = capsc * (1.0 / (1.0 + dm_exp(-z)));
        }
        const double trend = gtr * ys + fl;
        const double yh = trend * (1.0 + xm) + xa * ys;
        a.yhat[gid] = yh;
        if (a.t_out) { a.t_out[gid] = t; a.xa_out[gid] = xa * ys; a.opm_out[gid] = 1.0 + xm; }
        if (a.yhat_int) {
            // prophet_scorer.py:73 astype(int) truncates toward zero; :76-84 clamp to floor
            double tr = __builtin_trunc(yh);
            if (!(tr >= -2147483648.0)) tr = -2147483648.0;
            if (tr > 2147483647.0) tr = 2147483647.0;
            int32_t iv = (int32_t)tr;
            if ((double)iv < fl_clamp) iv = (int32_t)fl_clamp;
            a.yhat_int[gid] = iv;
        }
    }
}

// IEEE self test of the primitive operations the canonical order relies on
__global__ void selftest_kernel(int op, int64_t n, const double *a, const double *b, double *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], y = b ? b[i] : 0.0;
    double r = 0.0, s, c;
    switch (op) {
    case 0: r = x / y; break;
    case 1: r = __builtin_sqrt(x); break;
    case 2: r = dm_exp(x); break;
    case 3: r = dm_log(x); break;
    case 4: dm_sincos(x, s, c); r = s; break;
    case 5: dm_sincos(x, s, c); r = c; break;
    case 6: r = __builtin_fma(x, y, x); break;
    default: break;
    }
    out[i] = r;
}

}  // namespace tsf
