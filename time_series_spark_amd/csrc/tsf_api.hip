// tsf_api.hip -- C-ABI of libtsf_amd.so (include/tsf.h): context, device workspace, kernel
// dispatch, host-pointer convenience wrappers.
//
// The entry points stand where the reference calls into fbprophet per series:
//   fit      /root/reference/src/jobs/prophet_modeler.py:56-66
//   predict  /root/reference/src/jobs/prophet_scorer.py:64-84
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <atomic>
#include <vector>

#include "tsf_aux_kernels.h"
#include "tsf_interval_kernels.h"
#include "tsf_fit_kernels.h"
#include "tsf_quad_kernels.h"
#include "tsf_mfma_tabs.h"
#include "tsf_launch.h"

using namespace tsf;

struct tsf_ctx {
    int device;
    std::string err;
    // cached device workspace (grown on demand, never shrunk)
    void *ws;
    size_t ws_bytes;
    DevSpec *d_spec;
    void *nb_ws;            // slot records of the several-series-per-wave Newton kernel (grown on demand)
    size_t nb_ws_bytes;
    double *fut_tab;        // design table of a shared future grid (predict): [K][H]
    size_t fut_tab_bytes;
    void *iv_ws;            // scratch of tsf_predict_intervals_dev (per-row pieces + samples of one chunk; grown on demand, <= ~0.5 GB)
    size_t iv_ws_bytes;
    int32_t *order_dev[2];  // tsf_set_cost_hints: series in order of decreasing expected cost (two buffers, used in
    size_t order_cap[2];    // turn: a fit that is still running on its stream keeps reading the one it was given)
    int order_next;
    int64_t order_n;        // series count the pending hints are for (0: none pending)
    std::vector<int32_t> cost_host;   // the pending hints themselves: a ragged call cut into length classes hands every class its share
    hipEvent_t order_ev[2]; // recorded behind the launch that reads order_dev[b]: the buffer is rewritten only after it
    int order_busy[2];
    int n_cu;               // compute units of the device (persistent kernels: one workgroup each)
    const int *last_sp_flag;  // sparse-column route of the last fit call: its device flag (in the workspace), or null
    int opt[TSF_OPT_COUNT];   // tsf_set_option: route switches of THIS context (-1 = the library's default)
    int profiling;
    hipEvent_t ev0[TSF_PROFILE_RING], ev1[TSF_PROFILE_RING];
    int ev_created;
    long ev_count;          // profiled calls since profiling was enabled
};

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return -2;                                                                       \
        }                                                                                    \
    } while (0)

static int fail(tsf_ctx *ctx, const char *msg)
{
    if (ctx) ctx->err = msg;
    return -1;
}

extern "C" int tsf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int tsf_create(int device_id, tsf_ctx **out)
{
    if (!out) return -1;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return -2;
    if (hipSetDevice(device_id) != hipSuccess) return -2;
    tsf_ctx *c = new tsf_ctx();
    c->device = device_id; c->ws = nullptr; c->ws_bytes = 0; c->d_spec = nullptr;
    c->fut_tab = nullptr; c->fut_tab_bytes = 0;
    c->nb_ws = nullptr; c->nb_ws_bytes = 0;
    c->iv_ws = nullptr; c->iv_ws_bytes = 0;
    c->order_dev[0] = c->order_dev[1] = nullptr; c->order_cap[0] = c->order_cap[1] = 0; c->order_next = 0; c->order_n = 0;
    c->order_ev[0] = c->order_ev[1] = nullptr; c->order_busy[0] = c->order_busy[1] = 0;
    c->profiling = 0; c->ev_created = 0; c->ev_count = 0;
    c->last_sp_flag = nullptr;
    for (int i = 0; i < TSF_OPT_COUNT; ++i) c->opt[i] = -1;
    if (hipMalloc((void **)&c->d_spec, sizeof(DevSpec)) != hipSuccess) { delete c; return -2; }
    {
        hipDeviceProp_t prop;
        c->n_cu = (hipGetDeviceProperties(&prop, device_id) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    // (No hipMallocAsync anywhere in this library, and the device's default memory pool is left as the process
    // set it: round 3 measured intermittently wrong Newton fits after gigabytes of stream-ordered scratch
    // (DESIGN 5b), cause not found; every scratch buffer is a cached hipMalloc block of the context.)
    *out = c;
    return 0;
}

namespace { void pool_trim(int device); }    // host-pointer buffer cache, below

extern "C" void tsf_destroy(tsf_ctx *ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    pool_trim(ctx->device);
    if (ctx->ws) hipFree(ctx->ws);
    if (ctx->d_spec) hipFree(ctx->d_spec);
    if (ctx->fut_tab) hipFree(ctx->fut_tab);
    if (ctx->nb_ws) hipFree(ctx->nb_ws);
    if (ctx->iv_ws) hipFree(ctx->iv_ws);
    for (int b = 0; b < 2; ++b) { if (ctx->order_dev[b]) hipFree(ctx->order_dev[b]); if (ctx->order_ev[b]) hipEventDestroy(ctx->order_ev[b]); }
    if (ctx->ev_created)
        for (int i = 0; i < TSF_PROFILE_RING; ++i) { hipEventDestroy(ctx->ev0[i]); hipEventDestroy(ctx->ev1[i]); }
    delete ctx;
}

extern "C" const char *tsf_last_error(const tsf_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int tsf_set_option(tsf_ctx *ctx, int option, int value)
{
    if (!ctx) return -1;
    if (option < 0 || option >= TSF_OPT_COUNT) return fail(ctx, "unknown option");
    ctx->opt[option] = value;
    return 0;
}

extern "C" int tsf_get_option(const tsf_ctx *ctx, int option)
{
    if (!ctx || option < 0 || option >= TSF_OPT_COUNT) return -1;
    return ctx->opt[option];
}

extern "C" void tsf_spec_default(tsf_spec *s)
{
    memset(s, 0, sizeof(*s));
    s->growth = TSF_GROWTH_LINEAR; s->n_changepoints = 25; s->changepoint_range = 0.8;
    s->changepoint_prior_scale = 0.05;
    s->max_iter = 10000; s->history = 5; s->init_alpha = 1e-3; s->tol_obj = 1e-12;
    s->tol_rel_obj = 1e4; s->tol_grad = 1e-8; s->tol_rel_grad = 1e7; s->tol_param = 1e-8;
    s->eval_form = TSF_EVAL_AUTO;
    s->algorithm = TSF_ALGO_LBFGS;
    s->residual_kernel = TSF_RK_AUTO; s->recenter_every = 128; s->recenter_ratio = 1.0;
    s->coop_after = -1;
    s->converge = TSF_CONVERGE_STAN; s->map_max_iter = 10000; s->map_tol = 1e-7;
}

extern "C" int tsf_spec_size(void) { return (int)sizeof(tsf_spec); }
extern "C" int tsf_grid_info_size(void) { return (int)sizeof(tsf_grid_info); }

extern "C" int tsf_spec_K(const tsf_spec *s)
{
    int K = s->n_extra;
    for (int i = 0; i < s->n_seas; ++i) K += 2 * s->seas_order[i];
    return K;
}

extern "C" int tsf_theta_stride(const tsf_spec *s) { return 3 + s->n_changepoints + tsf_spec_K(s); }

// ---- spec validation + device form ---------------------------------------------------------

// parameters of the FIT: with no changepoints the model is fitted on one dummy changepoint
// (GridTab::S_fit), so there is always at least one delta
static int fit_P(int n_cp, int K) { return 3 + (n_cp > 0 ? n_cp : 1) + K; }

static int pick_KP(int K, int n_cp, int mixed)
{
    const int P = fit_P(n_cp, K);
    if (mixed || P > 64) return 64;
    if (K <= 8) return 8;
    if (K <= 16) return 16;
    if (K <= 28) return 28;
    return 64;
}

static int build_devspec(tsf_ctx *ctx, const tsf_spec *s, DevSpec *d, int *mode_out)
{
    if (!s) return fail(ctx, "spec is NULL");
    if (s->growth != TSF_GROWTH_LINEAR && s->growth != TSF_GROWTH_LOGISTIC) return fail(ctx, "bad growth");
    if (s->n_seas < 0 || s->n_seas > TSF_MAX_SEAS || s->n_extra < 0 || s->n_extra > TSF_MAX_EXTRA)
        return fail(ctx, "too many seasonalities / extra columns");
    if (s->n_changepoints < 0 || s->n_changepoints > TSF_MAX_S) return fail(ctx, "n_changepoints out of range");
    if (!(s->changepoint_range >= 0.0 && s->changepoint_range <= 1.0)) return fail(ctx, "changepoint_range must be in [0,1]");
    if (!(s->changepoint_prior_scale > 0.0)) return fail(ctx, "changepoint_prior_scale must be > 0");
    if (s->history < 1 || s->history > MAXH) return fail(ctx, "history must be in [1,8]");
    if (s->eval_form < TSF_EVAL_AUTO || s->eval_form > TSF_EVAL_QUADRATIC) return fail(ctx, "bad eval_form");
    if (s->algorithm < TSF_ALGO_LBFGS || s->algorithm > TSF_ALGO_AUTO) return fail(ctx, "bad algorithm");
    if (s->residual_kernel < TSF_RK_AUTO || s->residual_kernel > TSF_RK_COOP) return fail(ctx, "bad residual_kernel");
    if (s->converge != TSF_CONVERGE_STAN && s->converge != TSF_CONVERGE_MAP) return fail(ctx, "bad converge");
    if (s->converge == TSF_CONVERGE_MAP && (s->map_max_iter < 1 || !(s->map_tol > 0.0))) return fail(ctx, "map_max_iter must be >= 1 and map_tol > 0");
    if (s->eval_form != TSF_EVAL_RESIDUAL && (s->recenter_every < 1 || !(s->recenter_ratio > 0.0)))
        return fail(ctx, "recenter_every must be >= 1 and recenter_ratio > 0");
    const int K = tsf_spec_K(s);
    if (K < 1) return fail(ctx, "model needs at least one design column (fbprophet adds a zero column; pass one extra column of zeros)");
    if (K > TSF_MAX_K || fit_P(s->n_changepoints, K) > TSF_MAX_P) return fail(ctx, "too many parameters (3+S+K must be <= 128, K <= 64)");
    memset(d, 0, sizeof(*d));
    int mode[TSF_MAX_P];
    double pr[TSF_MAX_P];
    int col = 0, np = 0;
    for (int i = 0; i < s->n_seas; ++i) {
        if (s->seas_order[i] < 1) return fail(ctx, "fourier order must be >= 1");
        if (!(s->seas_prior_scale[i] > 0.0) || !(s->seas_period[i] > 0.0)) return fail(ctx, "bad seasonality prior scale / period");
        d->seas_period[i] = s->seas_period[i]; d->seas_order[i] = s->seas_order[i]; d->seas_col[i] = col;
        for (int h = 0; h < s->seas_order[i]; ++h) {
            np++;
            mode[col] = s->seas_mode[i]; pr[col] = s->seas_prior_scale[i]; col++;
            mode[col] = s->seas_mode[i]; pr[col] = s->seas_prior_scale[i]; col++;
        }
    }
    for (int e = 0; e < s->n_extra; ++e) {
        if (!(s->extra_prior_scale[e] > 0.0)) return fail(ctx, "bad extra prior scale");
        mode[col] = s->extra_mode[e]; pr[col] = s->extra_prior_scale[e]; col++;
    }
    int n = 0;
    for (int j = 0; j < K; ++j) if (mode[j] == TSF_MODE_ADDITIVE) { d->perm[n] = j; d->inv_perm[j] = n; d->prior[n] = pr[j]; n++; }
    const int Ka = n;
    for (int j = 0; j < K; ++j) if (mode[j] != TSF_MODE_ADDITIVE) { d->perm[n] = j; d->inv_perm[j] = n; d->prior[n] = pr[j]; n++; }
    for (int j = K; j < TSF_MAX_P; ++j) d->prior[j] = 1.0;
    const int m = (Ka == K) ? 0 : (Ka == 0 ? 1 : 2);
    d->growth = s->growth; d->n_cp = s->n_changepoints; d->K = K; d->Ka = Ka;
    d->KP = pick_KP(K, s->n_changepoints, m == 2);
    d->n_seas = s->n_seas; d->n_extra = s->n_extra; d->n_pairs = np;
    d->max_iter = s->max_iter; d->history = s->history;
    // harmonic structure (harm_code): one column mode (then the internal column order is fbprophet's: the seasonalities'
    // Fourier columns first, in order) and at most three seasonalities; whether a kernel is compiled for it is the
    // launch function's business (tsf_inst.inc)
    d->harm = 0;
    if (m != 2 && s->n_seas >= 1 && s->n_seas <= 3) {
        bool ok = true;
        for (int i = 0; i < s->n_seas; ++i) ok = ok && s->seas_order[i] <= 255;
        if (ok) d->harm = harm_code(s->seas_order[0], s->n_seas > 1 ? s->seas_order[1] : 0, s->n_seas > 2 ? s->seas_order[2] : 0);
    }
    d->cp_range = s->changepoint_range; d->tau = s->changepoint_prior_scale;
    d->init_alpha = s->init_alpha; d->tol_obj = s->tol_obj; d->tol_rel_obj = s->tol_rel_obj;
    d->tol_grad = s->tol_grad; d->tol_rel_grad = s->tol_rel_grad; d->tol_param = s->tol_param;
    *mode_out = m;
    return 0;
}

// ---- workspace ------------------------------------------------------------------------------

struct WsLayout {
    size_t gtab, stab, tw, cw, Xw, yw, Mg, Mslot, rbuf, counter, uw, Xu, spm, spp, Bw;
    size_t mXF, mXB, mXT, mtq, mcq, mcpof, myq, mhist;      // matrix-core path (tsf_mfma_kernels.h)
    size_t clist, cslots;                                   // cooperative tail (tsf_coop_kernels.h)
    size_t total;
};

// checkpoint slots of the cooperative tail: one per fit that can be suspended at a time -- the blocks
// resident when the launch runs dry (a few thousand), all series of a small call
constexpr size_t TSF_NB_KEEP = (size_t)1 << 30;

// (after: FitArgs::coop_after.  COOP_DIRECT suspends nothing; the tail rule (< 0) suspends at most the fits
// still running when no more of them are left than there are CUs -- 2 n_cu covers the waves racing past the
// test; an explicit hand-over point (>= 0: tests, latency mode) can suspend every running fit)
static int coop_slots_for(int64_t N, int after, int n_cu)
{
    if (after == COOP_DIRECT) return 0;
    const int64_t cap = after >= 0 ? 8192 : 8 * (int64_t)n_cu;      // (hand-over at up to ~4 fits per CU, twice that for the waves racing past the test)
    return (int)(N < cap ? N : cap);
}

// launch plan of the matrix-core residual kernel
struct MfmaPlan { int on, NG, KF, NCB, rr0, blocks; };

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static WsLayout ws_layout(int64_t N, int64_t n_grids, int NTmax, int KP, int quad_P4 = 0,
                          int quad_slots = 0, int quad_ragged = 0, int64_t lat_U = 0,
                          const MfmaPlan *mp = nullptr, int coop_slots = 0, int coop_stride = 0, int64_t quad_pre = 0,
                          bool sparse = false, int bw_ns = 0, bool no_yw = false)
{
    WsLayout l;
    size_t off = 0;
    l.gtab = off; off = align_up(off + sizeof(GridTab) * (size_t)n_grids);
    l.stab = off; off = align_up(off + sizeof(SeriesTab) * (size_t)N);
    l.tw = off; off = align_up(off + sizeof(double) * (size_t)n_grids * NTmax * W);
    l.cw = off; off = align_up(off + sizeof(uint16_t) * (size_t)n_grids * NTmax * W);
    // lattice panels (lat_U > 0) keep one shared table Xu and a row index per series row
    // instead of a design matrix per grid
    l.Xw = off; off = align_up(off + (lat_U > 0 ? 0 : sizeof(double) * (size_t)n_grids * NTmax * KP * W));
    l.yw = off; off = align_up(off + (no_yw ? 0 : sizeof(double) * (size_t)N * NTmax * W));
    // one Gram matrix for an aligned panel; quad_pre of them for a ragged panel whose series share timestamp vectors
    l.Mg = off; off = align_up(off + sizeof(double) * (size_t)quad_P4 * 2 * W * (size_t)(quad_pre > 0 ? quad_pre : 1));
    l.Mslot = off; off = align_up(off + (quad_ragged ? sizeof(double) * (size_t)quad_slots * quad_P4 * 2 * W : 0));
    l.rbuf = off; off = align_up(off + sizeof(double) * (size_t)quad_slots * NTmax * W);
    l.counter = off; off = align_up(off + 256);
    // sparse indicator columns (SP_* in tsf_fit_kernels.h): the lanes' entries and the columns' fold programs per grid
    l.spm = off; off = align_up(off + (sparse ? sizeof(uint32_t) * (size_t)n_grids * SP_M * W : 0));
    l.spp = off; off = align_up(off + (sparse ? sizeof(unsigned long long) * (size_t)n_grids * SP_MAXC : 0));
    // base pairs of the Fourier columns (fit_kernel<..., HARM>): two doubles per seasonality and row
    // (a panel on a timestamp lattice: per lattice POINT, one table for every series -- FitArgs::Bu)
    l.Bw = off; off = align_up(off + sizeof(double) * (lat_U > 0 ? (size_t)lat_U * bw_ns * 2 : (size_t)n_grids * NTmax * bw_ns * 2 * W));
    l.uw = off; off = align_up(off + (lat_U > 0 ? sizeof(int32_t) * (size_t)n_grids * NTmax * W : 0));
    l.Xu = off; off = align_up(off + (lat_U > 0 ? sizeof(double) * (size_t)lat_U * KP : 0));
    const bool mf = mp && mp->on;
    const size_t tiles = mf ? (size_t)W * mp->NG : 0;
    l.mXF = off; off = align_up(off + (mf ? sizeof(double) * tiles * mp->KF * W : 0));
    l.mXB = off; off = align_up(off + (mf ? sizeof(double) * tiles * 4 * mp->NCB * W : 0));
    l.mXT = off; off = align_up(off + (mf ? sizeof(double) * tiles * 4 * W : 0));
    l.mtq = off; off = align_up(off + (mf ? sizeof(double) * tiles * 16 : 0));
    l.mcq = off; off = align_up(off + (mf ? sizeof(uint16_t) * tiles * 16 : 0));
    l.mcpof = off; off = align_up(off + (mf ? (size_t)W * 8 : 0));
    l.myq = off; off = align_up(off + (mf ? sizeof(double) * (size_t)N * tiles * 16 : 0));
    l.mhist = off; off = align_up(off + (mf ? sizeof(double) * (size_t)mp->blocks * MT_NS * 2 * MAXH * W : 0));
    l.clist = off; off = align_up(off + sizeof(int32_t) * (size_t)coop_slots);
    l.cslots = off; off = align_up(off + sizeof(double) * (size_t)coop_slots * coop_stride);
    l.total = off;
    return l;
}

static int ensure_ws(tsf_ctx *ctx, size_t bytes, int64_t N = 0, int NTmax = 0, int ragged = 0)
{
    if (ctx->ws_bytes >= bytes) return 0;
    {
        // a ragged panel pads EVERY series' tables to the longest series of the call: say so instead of
        // failing inside hipMalloc
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes > free_b + ctx->ws_bytes && (ctx->nb_ws || ctx->iv_ws)) {
            // the cached scratch of other entry points (Newton slot records: up to 6 GB; interval samples) gives way
            if (ctx->nb_ws) { (void)hipFree(ctx->nb_ws); ctx->nb_ws = nullptr; ctx->nb_ws_bytes = 0; }
            if (ctx->iv_ws) { (void)hipFree(ctx->iv_ws); ctx->iv_ws = nullptr; ctx->iv_ws_bytes = 0; }
            (void)hipMemGetInfo(&free_b, &total_b);
        }
        if (free_b + total_b > 0 && bytes > free_b + ctx->ws_bytes) {
            char msg[384];
            snprintf(msg, sizeof(msg), "workspace of %.1f GB does not fit the device (%.1f GB free): %lld series, the longest has "
                     "%d rows%s", bytes / 1e9, (free_b + ctx->ws_bytes) / 1e9, (long long)N, NTmax * W,
                     ragged ? " and every series of a ragged call is padded to it -- split the call by series length" : "");
            ctx->err = msg;
            return -2;
        }
        (void)hipGetLastError();
    }
    ctx->last_sp_flag = nullptr;       // (it points into the workspace that goes away here)
    if (ctx->ws) { HIP_TRY(ctx, hipFree(ctx->ws)); ctx->ws = nullptr; ctx->ws_bytes = 0; }
    HIP_TRY(ctx, hipMalloc(&ctx->ws, bytes));
    ctx->ws_bytes = bytes;
    return 0;
}

// grid / LDS plan of the quadratic-form kernel for this device
static int quad_plan(tsf_ctx *ctx, const DevSpec &hs, int64_t N, QuadPlan *qp)
{
    const int P = fit_P(hs.n_cp, hs.K);
    qp->PPL = (hs.KP == 64) ? 2 : 1;
    // rows of M the kernel walks: compile-time 40 / 56 / 64 for the one-slot kernels (zero rows
    // beyond P are bit-neutral), P rounded up to 4 for the two-slot kernel
    qp->P4 = (qp->PPL == 2) ? ((P + 3) & ~3) : (P <= 40 ? 40 : (P <= 56 ? 56 : 64));
    qp->NW = quad_waves_per_block(qp->PPL);       // the most any variant launches: sizes the slots
    qp->n_cu = ctx->n_cu;
    qp->opt = ctx->opt;
    int64_t blocks = ctx->n_cu;                     // persistent: LDS admits one workgroup per CU
    const int64_t need = (N + qp->NW - 1) / qp->NW;
    if (blocks > need) blocks = need;
    if (blocks < 1) blocks = 1;
    qp->blocks = (int)blocks;
    // resident waves over all kernel variants (their workgroups hold 4, 8 or 12 waves; each launcher
    // sizes its own grid): at most n_cu x NW, and no more than one per series rounded up to a workgroup
    int64_t slots = (int64_t)ctx->n_cu * qp->NW;
    if (slots > N + qp->NW) slots = N + qp->NW;
    qp->slots = (int)slots;
    if (qp->slots < qp->P4) qp->slots = qp->P4;
    return 0;
}

typedef int (*launch_t)(int, const FitArgs &, int, hipStream_t);
typedef int (*launch_newton_t)(int, const FitArgs &, int, hipStream_t);
static launch_newton_t pick_newton_launch(int growth, int mode)
{
    static const launch_newton_t tab[2][3] = {{launch_newton_g0m0, launch_newton_g0m1, launch_newton_g0m2},
                                              {launch_newton_g1m0, launch_newton_g1m1, launch_newton_g1m2}};
    return tab[growth][mode];
}

typedef int (*launch_map_t)(int, const FitArgs &, hipStream_t);
static launch_map_t pick_map_launch(int growth, int mode)
{
    static const launch_map_t tab[2][3] = {{launch_map_g0m0, launch_map_g0m1, launch_map_g0m2},
                                           {launch_map_g1m0, launch_map_g1m1, launch_map_g1m2}};
    return tab[growth][mode];
}

static launch_t pick_launch(int growth, int mode)
{
    static const launch_t tab[2][3] = {{launch_g0m0, launch_g0m1, launch_g0m2},
                                       {launch_g1m0, launch_g1m1, launch_g1m2}};
    return tab[growth][mode];
}

// Common driver: setup kernels + fit (or eval-only) kernel on device pointers.
static int run_fit(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int aligned, int32_t T,
                   const int64_t *offsets, int64_t total_rows, int32_t max_T, const int64_t *ds,
                   const void *y, int32_t y_dtype, const double *floor_, const double *cap,
                   const double *extra, tsf_fit_out *out, const double *theta_in,
                   double *grad_out, hipStream_t st, int64_t lat_base = 0, int64_t lat_step = 0,
                   int64_t lat_U = 0, const double *theta_ref = nullptr,
                   const int32_t *grid_of = nullptr, const int64_t *grid_rows = nullptr, int64_t n_distinct = 0,
                   const int32_t *grid_order = nullptr)
{
    if (!ctx) return -1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (N <= 0) return fail(ctx, "N must be > 0");
    if (!ds || !y || !out) return fail(ctx, "NULL input");
    if (y_dtype < TSF_Y_F64 || y_dtype > TSF_Y_I32) return fail(ctx, "bad y_dtype");
    DevSpec hs;
    int mode = 0;
    int rc = build_devspec(ctx, spec, &hs, &mode);
    if (rc) return rc;
    if (hs.n_extra > 0 && !extra) return fail(ctx, "extra columns declared but extra is NULL");
    if (hs.growth == TSF_GROWTH_LOGISTIC && !cap) return fail(ctx, "logistic growth needs cap");
    const int Tm = aligned ? T : max_T;
    if (Tm < 1) return fail(ctx, "no rows");
    if (Tm > TSF_MAX_T) return fail(ctx, "series too long (TSF_MAX_T rows per series)");
    const int NTmax = (Tm + W - 1) / W;
    // ragged panel whose series share timestamp vectors (fit_host found n_distinct < N of them): one set of grid tables
    // per distinct vector
    if (aligned || !grid_of || !grid_rows || n_distinct <= 0 || n_distinct >= N) { grid_of = nullptr; grid_rows = nullptr; n_distinct = 0; }
    const int64_t n_grids = aligned ? 1 : (grid_of ? n_distinct : N);
    // quadratic (Gram) form of the data term: see tsf_quad_kernels.h
    // Stan's Newton optimiser (tsf_newton_kernels.h): explicitly, or by fbprophet's rule on the
    // longest series of the call
    const bool newton = theta_in == nullptr &&
                        (spec->algorithm == TSF_ALGO_NEWTON ||
                         (spec->algorithm == TSF_ALGO_AUTO && Tm < TSF_NEWTON_BELOW_T));
    // (more than 64 parameters, or mixed additive / multiplicative columns: newton_kernel2, two parameters per lane)
    if (newton && fit_P(hs.n_cp, hs.K) > 2 * W)
        return fail(ctx, "Newton needs 3 + n_changepoints + K <= 128");
    const bool quad_ok = hs.growth == TSF_GROWTH_LINEAR && mode == 0 && hs.history == QH &&
                         theta_in == nullptr && !newton;
    if (spec->eval_form == TSF_EVAL_QUADRATIC && !quad_ok && theta_in == nullptr)
        return fail(ctx, "eval_form QUADRATIC needs linear growth, additive columns only and history == 5");
    const bool quad = quad_ok && spec->eval_form != TSF_EVAL_RESIDUAL;
    // Newton on the same models: residual form at the accepted points, quadratic form for the
    // finite-difference and halving evaluations (tsf_newton_quad.h); aligned panels
    const bool newton_quad = newton && hs.growth == TSF_GROWTH_LINEAR && mode == 0 && hs.KP <= 28 &&
                             spec->eval_form != TSF_EVAL_RESIDUAL && NTmax <= 16;
    // tsf_eval_quadratic: one quadratic-form evaluation at theta_in around the reference point theta_ref
    const bool quad_eval = theta_in != nullptr && theta_ref != nullptr;
    if (quad_eval && !(aligned && hs.growth == TSF_GROWTH_LINEAR && mode == 0 && hs.KP != 64))
        return fail(ctx, "the quadratic form needs an aligned panel, linear growth, additive columns only and 3 + n_changepoints + K <= 64");
    // the slot records of the several-series-per-wave Newton kernel are cached between Newton calls, but at most
    // TSF_NB_KEEP bytes of them outlive a call that does not use them
    if (!(newton_quad && aligned) && ctx->nb_ws_bytes > TSF_NB_KEEP) {
        HIP_TRY(ctx, hipFree(ctx->nb_ws));
        ctx->nb_ws = nullptr; ctx->nb_ws_bytes = 0;
    }
    QuadPlan qp;
    memset(&qp, 0, sizeof(qp));
    if (quad || newton_quad || quad_eval) {
        rc = quad_plan(ctx, hs, N, &qp);
        if (rc) return rc;
        if (newton_quad) {          // one-wave workgroups, up to 16 per CU: that many Z^T Z slots (ragged)
            qp.slots = ctx->n_cu * 16;
            if (qp.slots < qp.P4) qp.slots = qp.P4;
        }
    }
    // shared lattice table: ragged panel, residual-form kernel, no explicit columns
    if (aligned || quad || newton || theta_in != nullptr || hs.n_extra > 0 || lat_step <= 0) lat_U = 0;
    {
        // Series that share timestamp vectors (grid_of) have step-major tables per DISTINCT vector: where those fit the
        // caches (<= 64 MB of design tables) every wave reads them coalesced, as on an aligned panel, and the lattice
        // table with its gathered rows (64 cache lines per load instruction) is the slower of the two.
        // tsf_set_option(TSF_OPT_LATTICE, 0): never the lattice table; 1: always where it applies.
        // Round 5: a model whose Fourier columns have a compiled expansion reads 32-byte base pairs per row from its OWN
        // tables (eval_fg HARM) -- faster than the gathered 224-byte rows of the lattice table in the one-wave kernel (ragged
        // bench panel, a grid per series: 93 -> 60 ms) and, above all, in the cooperative tail (gathered rows stream: 11 us
        // per evaluation against 5) -- wherever a table per series is affordable: <= 32 GB of design tables per call
        // (~190 000 series of 730 rows) AND no more than two fifths of the memory this context can have (what is free
        // now + its own cached workspace; the tables are ~4/5 of the layout): several contexts or ranks on one GPU, or
        // a smaller part, keep the shared lattice table -- the route that needs no table per series -- instead of
        // failing in ensure_ws (round-5 advice).
        // Round 6: the base-pair kernels read a lattice panel too (rows of 22 bytes + the lattice POINTS' pairs from one
        // shared table, FitArgs::Bu: as fast or faster than a table per series -- 10 000 / 2 000 series at their own subsets
        // of a daily lattice: 75.9 -> 74.5 / 31.1 -> 27.0 ms -- for 0.44 of the traffic and none of the 2 GB of tables), so
        // such a model keeps the lattice (its MAP continuation too).
        const int el = ctx->opt[TSF_OPT_LATTICE];
        const size_t tab = sizeof(double) * (size_t)n_grids * (size_t)NTmax * hs.KP * W;
        const bool harm_model = ctx->opt[TSF_OPT_HARM] != 0 && mode != 2 &&
                                ((hs.harm == HARM_Y10_W3 && hs.KP == 28) || (hs.harm == HARM_W3_D4 && hs.KP == 16) || (hs.harm == HARM_W3 && hs.KP == 8));
        const bool lat_harm = harm_model && hs.K == harm_kf(hs.harm);
        size_t harm_cap = (size_t)32 << 30;
        if (lat_U > 0 && el < 0 && harm_model && !lat_harm && tab > ((size_t)256 << 20)) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const size_t mine = (free_b + ctx->ws_bytes) / 5 * 2;
                if (mine < harm_cap) harm_cap = mine;
            } else {
                (void)hipGetLastError();
            }
        }
        if (el >= 0 ? el == 0 : ((grid_of != nullptr && tab <= ((size_t)64 << 20)) || (harm_model && !lat_harm && tab <= harm_cap))) lat_U = 0;
    }
    // matrix-core residual kernel (tsf_mfma_kernels.h): aligned panel, L-BFGS, one parameter per lane
    // (KP <= 28 implies one column mode and P <= 64), at most MT_SP changepoints, and an upper bound
    // on the changepoint rows one chunk can hold (the exact count is checked on the device:
    // MfmaTabs::overflow routes the call to the one-wave kernel)
    MfmaPlan mp;
    memset(&mp, 0, sizeof(mp));
    // TSF_RK_AUTO = the one-wave kernel.  Measured (profiles/r02_mfma_vs_wave.txt): both kernels issue
    // the same number of vector instructions per evaluation and are bound by them -- fp64 MFMA has
    // the vector unit's rate, and the 672 multiply-adds it takes over are 16 % of an evaluation --
    // so once the reductions of the optimiser were cut (uniform butterflies) the one-wave kernel,
    // which has no rounds to synchronise, is ahead on every panel measured (100 000 x 730, iteration
    // cap 150: 271 vs 326 ms; reference settings with their stragglers: 1.27 vs 1.62 s).  The
    // matrix-core kernel stays available (TSF_RK_MFMA), bit-identical.
    const bool want_mfma = spec->residual_kernel == TSF_RK_MFMA;
    if (aligned && !quad && !newton && theta_in == nullptr && hs.KP <= 28 && want_mfma) {
        const int hist_rows = (int)floor((double)Tm * hs.cp_range);
        int S = hs.n_cp;
        if (S + 1 > hist_rows) S = hist_rows - 1;
        if (S < 1) S = 1;
        const double step = (hs.n_cp > 0 && S > 0) ? (double)(hist_rows - 1) / (double)S : 1e30;
        const int per_chunk = (int)ceil((double)NTmax / (step > 1.0 ? step : 1.0)) + 2;
        if (S <= MT_SP && per_chunk <= MT_MAXCP && mfma_lds_bytes(hs.KP) <= 160 * 1024) {
            mp.on = 1;
            mp.NG = (NTmax + 15) / 16;
            mp.KF = hs.KP / 4;
            mp.NCB = (hs.KP + 15) / 16;
            mp.rr0 = (16 * mp.NG - NTmax) / 4;
            int64_t blocks = (N + MT_NS - 1) / MT_NS;
            if (blocks > ctx->n_cu) blocks = ctx->n_cu;
            mp.blocks = (int)blocks;
        }
    }
    if (spec->residual_kernel == TSF_RK_MFMA && !mp.on && theta_in == nullptr && !quad && !newton)
        return fail(ctx, "residual_kernel MFMA needs an aligned panel, K <= 28 columns of one mode, 3+S+K <= 64 and S <= 28");
    // cooperative tail of the one-wave residual kernel: series of at most COOP_MAX_NT steps per chunk
    // (the rows of one evaluation are staged in the workgroup's LDS)
    const bool coop_ok = !quad && !newton && theta_in == nullptr && !mp.on && NTmax <= COOP_MAX_NT;
    if (spec->residual_kernel == TSF_RK_COOP && !coop_ok && theta_in == nullptr)
        return fail(ctx, "residual_kernel COOP needs a residual-form L-BFGS fit of series of at most 4096 rows");
    const bool coop = coop_ok && spec->residual_kernel != TSF_RK_WAVE;
    // where fits are handed over (FitArgs::coop_after).  TSF_RK_COOP: the cooperative kernel runs every fit from
    // its first evaluation (no one-wave phase, no checkpoints).  AUTO does the same for the models whose one-wave
    // kernel holds two parameters per lane (KP = 64) on series too long for the grouped column sums: that kernel
    // needs > 256 registers (one wave per SIMD) and measures 9.6 M evaluations/s on 50 000 x 730 with 56 columns,
    // the workgroup kernel 11.4+ M.
    // Round 3: on series of <= 768 rows the one-wave kernel of these models takes its per-column sums in groups
    // of 8 (eval_fg GNTR: 187 instead of 379 registers, two waves per SIMD) and allocates only the history pairs
    // it uses (wave_lds_bytes: eight blocks per CU instead of six): 17.0 M evaluations/s on cfg4 against the
    // workgroup kernel's 15.4 M (2.27 against 2.50 s) -- so AUTO runs it, with the workgroup kernel for the tail
    // as for the narrower models.  tsf_set_option(TSF_OPT_FIT_GROUPED, 0): every series on the workgroup kernel, as before.
    // Wide models (64-column tables) whose columns from the 29th on are explicit columns -- holidays: 0 / 1 indicators,
    // almost all 0 -- are tried on the sparse-column form of the 28-column kernel (eval_fg<..., SPARSE>, tsf_fit_kernels.h):
    // sparse_extra_kernel decides on the device whether every grid qualifies, the dense route is launched behind it
    // with the opposite guard (series of <= 768 rows: the grouped one-wave kernel; up to 4 096 rows: the workgroup
    // kernel from the first evaluation; longer: the ungrouped one-wave kernel).  tsf_set_option(TSF_OPT_SPARSE_EXTRA, 0): never.
    const bool sparse_try = !quad && !newton && !mp.on && theta_in == nullptr && lat_U == 0 && hs.KP == 64 && mode != 2 &&
                            spec->residual_kernel == TSF_RK_AUTO && hs.K > SP_DENSE && hs.K <= SP_DENSE + SP_MAXC &&
                            hs.K - hs.n_extra <= SP_DENSE && NTmax <= SP_MAX_NT && ctx->opt[TSF_OPT_SPARSE_EXTRA] != 0;
    int coop_after = spec->residual_kernel == TSF_RK_COOP ? COOP_DIRECT : spec->coop_after;
    if (coop) {
        const bool grouped = NTmax <= 12 && ctx->opt[TSF_OPT_FIT_GROUPED] != 0 && lat_U == 0;
        // (with sparse_try the slots of the tail exist either way: the dense route of longer series is then launched
        // in direct mode by the launch function itself)
        if (spec->residual_kernel == TSF_RK_AUTO && spec->coop_after < 0 && hs.KP == 64 && !grouped && !sparse_try) coop_after = COOP_DIRECT;
    }
    const int coop_slots = coop ? coop_slots_for(N, coop_after, ctx->n_cu) : 0;
    const int coop_stride = coop ? coop_slot_doubles(hs.KP == 64 ? 2 : 1) : 0;
    // ragged panel, quadratic form, series sharing timestamp vectors: Z^T Z once per distinct vector (gram_grids_kernel)
    // instead of once per series inside the fit kernel -- when that at least halves the builds (TSF_OPT_GRAM_SHARE 0: never)
    const int64_t quad_pre = (quad && !aligned && grid_of && n_grids * 2 <= N && ctx->opt[TSF_OPT_GRAM_SHARE] != 0) ? n_grids : 0;
    // Fourier columns expanded from the rows' base pairs (eval_fg HARM): the residual-form one-wave kernel of models
    // whose harmonic structure has a compiled kernel -- yearly 10 + weekly 3 on the 28-column kernel and its
    // sparse-column form, weekly 3 + daily 4 (16 columns), weekly 3 (8 columns); never with the lattice table or the
    // matrix-core / workgroup-from-the-start routes.  tsf_set_option(TSF_OPT_HARM, 0): never.
    // Round 6: WITH the lattice table where the model is exactly one of those (nothing behind the Fourier block): a row keeps t, y, its segment word and its lattice point (22 bytes instead of 50) and the base pairs
    // are the point's, from one table of 32 bytes per point that every series shares (FitArgs::Bu) -- no table per series.
    int harm = 0;
    if (!quad && !newton && !mp.on && theta_in == nullptr && lat_U == 0 && ctx->opt[TSF_OPT_HARM] != 0) {
        if ((hs.harm == HARM_Y10_W3 && (hs.KP == 28 || sparse_try)) || (hs.harm == HARM_W3_D4 && hs.KP == 16) ||
            (hs.harm == HARM_W3 && hs.KP == 8))
            harm = hs.harm;
    } else if (!quad && !newton && !mp.on && theta_in == nullptr && lat_U > 0 && ctx->opt[TSF_OPT_HARM] != 0 && mode != 2 &&
               hs.K == harm_kf(hs.harm) &&
               ((hs.harm == HARM_Y10_W3 && hs.KP == 28) || (hs.harm == HARM_W3_D4 && hs.KP == 16) || (hs.harm == HARM_W3 && hs.KP == 8))) {
        harm = hs.harm;
    }
    // The same base pairs for the Gram build of a ragged quadratic-form panel whose series have a calendar each (the M-in-
    // registers kernel builds Z^T Z per series: gram_columns_harm, tsf_quad_kernels.h); shared calendars build once per
    // calendar (quad_pre) and keep reading the tables.
    int gram_harm = 0;
    if (quad && !aligned && !quad_pre && lat_U == 0 && hs.harm == HARM_Y10_W3 && hs.KP == 28 && ctx->opt[TSF_OPT_HARM] != 0)
        gram_harm = hs.harm;
    // ... and for the MAP continuation (tsf_map_kernels.h): its evaluator reads base-pair rows wherever the model has a
    // compiled expansion, whatever kernel ran the Stan-rule fit (the quadratic-form route builds the table for it)
    int map_harm = 0;
    if (spec->converge == TSF_CONVERGE_MAP && theta_in == nullptr && mode != 2 && ctx->opt[TSF_OPT_HARM] != 0 &&
        hs.K == harm_kf(hs.harm) &&
        ((hs.harm == HARM_Y10_W3 && hs.KP == 28) || (hs.harm == HARM_W3_D4 && hs.KP == 16) || (hs.harm == HARM_W3 && hs.KP == 8)))
        map_harm = hs.harm;
    const int bw_ns = (harm || gram_harm || map_harm) ? hs.n_seas : 0;
    // Quadratic-form L-BFGS fits read the caller's y rows themselves (FitArgs::y_raw) instead of a scaled step-major copy
    // that setup_series_kernel would write and they would read back -- a second f64 panel on the device and half of the
    // step's HBM bytes.  Not when the MAP continuation follows (its evaluator reads yw).
    const bool raw_y = quad && spec->converge != TSF_CONVERGE_MAP && ctx->opt[TSF_OPT_QUAD_RAW_Y] != 0;
    const WsLayout l = ws_layout(N, n_grids, NTmax, hs.KP, qp.P4, qp.slots, (quad || newton_quad) && !aligned, lat_U, &mp,
                                 coop_slots, coop_stride, quad_pre, sparse_try, bw_ns, raw_y);
    rc = ensure_ws(ctx, l.total, N, NTmax, !aligned);
    if (rc) return rc;
    char *ws = (char *)ctx->ws;
    GridTab *gtab = (GridTab *)(ws + l.gtab);
    SeriesTab *stab = (SeriesTab *)(ws + l.stab);
    double *tw = (double *)(ws + l.tw);
    uint16_t *cw = (uint16_t *)(ws + l.cw);
    double *Xw = (double *)(ws + l.Xw);
    double *yw = raw_y ? (double *)nullptr : (double *)(ws + l.yw);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_spec, &hs, sizeof(hs), hipMemcpyHostToDevice, st));
    if (lat_U > 0) {
        double *Xu = (double *)(ws + l.Xu);
        HIP_TRY(ctx, hipMemsetAsync(Xu, 0, sizeof(double) * (size_t)lat_U * hs.KP, st));
        // (rows a lane's chunk does not have point at lattice point 0: the base-pair kernels gather before they mask)
        HIP_TRY(ctx, hipMemsetAsync(ws + l.uw, 0, sizeof(int32_t) * (size_t)n_grids * NTmax * W, st));
        const int64_t work = lat_U * (hs.n_seas > 0 ? hs.n_seas : 1);
        hipLaunchKernelGGL(setup_lattice_kernel, dim3((unsigned)((work + 255) / 256 > 65535 ? 65535 : (work + 255) / 256)),
                           dim3(256), 0, st, ctx->d_spec, lat_U, lat_base, lat_step, Xu, bw_ns ? (double *)(ws + l.Bw) : (double *)nullptr);
        HIP_TRY(ctx, hipGetLastError());
    } else {
        HIP_TRY(ctx, hipMemsetAsync(Xw, 0, sizeof(double) * (size_t)n_grids * NTmax * hs.KP * W, st));
        // (the base pairs too: rows past a chunk's end are (0, 0) -- every harmonic 0 --, never stale bits that could be
        // NaN or Inf under an `fma(x, 0, acc)` that relies on finite x; round-5 advice)
        if (bw_ns) HIP_TRY(ctx, hipMemsetAsync(ws + l.Bw, 0, sizeof(double) * (size_t)n_grids * NTmax * bw_ns * 2 * W, st));
    }
    HIP_TRY(ctx, hipMemsetAsync(gtab, 0, sizeof(GridTab) * (size_t)n_grids, st));
    hipLaunchKernelGGL(setup_grid_kernel, dim3((unsigned)n_grids), dim3(256), 0, st, ctx->d_spec,
                       (int)n_grids, aligned ? nullptr : offsets, T, ds, extra,
                       aligned ? (int64_t)T : total_rows, NTmax, gtab, tw, cw, Xw,
                       (int32_t *)(ws + l.uw), lat_base, lat_U > 0 ? lat_step : (int64_t)0, grid_rows,
                       (bw_ns && lat_U == 0) ? (double *)(ws + l.Bw) : (double *)nullptr);
    HIP_TRY(ctx, hipGetLastError());
    if (sparse_try) {
        int *sp_bad = (int *)(ws + l.counter) + 8;
        HIP_TRY(ctx, hipMemsetAsync(sp_bad, 0, sizeof(int), st));
        hipLaunchKernelGGL(sparse_extra_kernel, dim3((unsigned)n_grids), dim3(64), 0, st, ctx->d_spec, gtab, Xw, NTmax,
                           (uint32_t *)(ws + l.spm), (unsigned long long *)(ws + l.spp), sp_bad);
        HIP_TRY(ctx, hipGetLastError());
    }
    // (raw_y: the quadratic-form fit kernel derives the scale and the initial values from the caller's rows itself --
    // series_tab_wave, tsf_quad_kernels.h -- and no kernel of that route reads SeriesTab or yw)
    if (!raw_y) {
        hipLaunchKernelGGL(setup_series_kernel, dim3((unsigned)N), dim3(64), 0, st, ctx->d_spec, N,
                           aligned ? nullptr : offsets, T, ds, y, y_dtype, floor_, cap, NTmax, gtab,
                           aligned, stab, yw, grid_of);
        HIP_TRY(ctx, hipGetLastError());
    }
    FitArgs a;
    memset(&a, 0, sizeof(a));
    a.sp = ctx->d_spec; a.N = N; a.aligned = aligned; a.NTmax = NTmax;
    a.theta_stride = tsf_theta_stride(spec);
    {
        const double eps = 2.220446049250313e-16;
        a.opt.init_alpha = hs.init_alpha; a.opt.tol_obj = hs.tol_obj;
        a.opt.tol_rel_obj_eps = hs.tol_rel_obj * eps; a.opt.tol_grad = hs.tol_grad;
        a.opt.tol_rel_grad_eps = hs.tol_rel_grad * eps; a.opt.tol_param = hs.tol_param;
        a.opt.max_iter = hs.max_iter; a.opt.history = hs.history;
    }
    a.gtab = gtab; a.stab = stab; a.tw = tw; a.yw = yw; a.Xw = Xw; a.cw = cw;
    a.theta = out->theta; a.y_scale = out->y_scale; a.fval = out->fval; a.status = out->status;
    a.n_iter = out->n_iter; a.n_eval = out->n_eval; a.grid_out = out->grid;
    a.theta_in = theta_in; a.grad_out = grad_out;
    if (raw_y) { a.y_raw = y; a.y_raw_dtype = y_dtype; a.y_offsets = aligned ? nullptr : offsets; a.y_T = T; }
    a.uw = (const int32_t *)(ws + l.uw); a.Xu = (const double *)(ws + l.Xu); a.xidx = lat_U > 0 ? 1 : 0;
    a.grid_of = grid_of;
    a.Bw = (bw_ns && lat_U == 0) ? (const double *)(ws + l.Bw) : nullptr; a.bw_ns = bw_ns; a.harm = harm;
    a.Bu = (bw_ns && lat_U > 0) ? (const double *)(ws + l.Bw) : nullptr;
    a.coop_harm = (harm != 0 && hs.K == harm_kf(harm) && mode != 2) ? 1 : 0;
    {
        // rows of the one-wave base-pair kernel from HBM?  (tables per grid: segment words, t, base pairs; y per series)
        const size_t row_tabs = (size_t)n_grids * NTmax * W * (sizeof(double) + sizeof(uint16_t) + sizeof(double) * 2 * (size_t)bw_ns) +
                                (size_t)N * NTmax * W * sizeof(double);
        const int oh = ctx->opt[TSF_OPT_HARM];
        a.harm_pf = (harm != 0 && (oh == 2 || (oh != 1 && !aligned && row_tabs > ((size_t)256 << 20)))) ? 1 : 0;
        if (harm != 0 && lat_U > 0) a.harm_pf = 1;         // (the lattice form exists with the row prefetch only)
    }
    a.opt_coop_sparse = ctx->opt[TSF_OPT_SPARSE_EXTRA] != 2 ? 1 : 0;     // (2: the sparse fit kernel with the 64-column tail, for A/B runs)
    ctx->last_sp_flag = nullptr;
    if (sparse_try) {
        a.sp_meta = (const uint32_t *)(ws + l.spm); a.sp_prog = (const unsigned long long *)(ws + l.spp);
        a.sp_flag = (int *)(ws + l.counter) + 8;
        ctx->last_sp_flag = a.sp_flag;
    }
    // scheduling hints of tsf_set_cost_hints: for this call if they were given for this many series; used once
    const int order_buf = (ctx->order_n == N && !theta_in) ? (ctx->order_next ^ 1) : -1;
    if (order_buf >= 0) a.order = ctx->order_dev[order_buf];
    else if (grid_order && ctx->opt[TSF_OPT_GRID_ORDER] != 0) a.order = grid_order;       // series grouped by shared grid (fit_host_one)
    ctx->order_n = 0;
    const int slot = (int)(ctx->ev_count % TSF_PROFILE_RING);
    if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(ctx->ev0[slot], st));
    int lrc;
    bool map_done = false;
    if (newton_quad) {
        QuadArgs qa;
        memset(&qa, 0, sizeof(qa));
        qa.f = a; qa.Mg = (const double *)(ws + l.Mg); qa.Mslot = (double *)(ws + l.Mslot);
        qa.rbuf = (double *)(ws + l.rbuf);
        qa.counter = (int *)(ws + l.counter); qa.P4 = qp.P4;
        if (aligned && !(ctx->opt[TSF_OPT_DEBUG_ASYNC_SCRATCH] > 0 && (ctx->opt[TSF_OPT_DEBUG_ASYNC_SCRATCH] & 1))) {
            const size_t need = newton_batch_scratch_bytes(hs.KP, fit_P(hs.n_cp, hs.K) | 1, N, NTmax, ctx->n_cu, ctx->opt);
            if (need > ctx->nb_ws_bytes) {
                if (ctx->nb_ws) { HIP_TRY(ctx, hipFree(ctx->nb_ws)); ctx->nb_ws = nullptr; ctx->nb_ws_bytes = 0; }
                if (hipMalloc(&ctx->nb_ws, need) == hipSuccess) ctx->nb_ws_bytes = need;
                else { ctx->nb_ws = nullptr; (void)hipGetLastError(); }       // no room: the one-series-per-wave kernel
            }
            qa.nb_buf = ctx->nb_ws; qa.nb_bytes = ctx->nb_ws_bytes;
        }
        HIP_TRY(ctx, hipMemsetAsync(qa.counter, 0, sizeof(int), st));
        lrc = launch_newton_quad(hs.KP, qp, qa, (double *)(ws + l.Mg), fit_P(hs.n_cp, hs.K) | 1, ctx->n_cu, st);
        // (-1: this shape does not fit the quadratic-form Newton kernel's LDS -- the residual-form kernel has no such limit)
        if (lrc == -1) lrc = pick_newton_launch(hs.growth, mode)(hs.KP, a, fit_P(hs.n_cp, hs.K), st);
    } else if (newton) {
        lrc = pick_newton_launch(hs.growth, mode)(hs.KP, a, fit_P(hs.n_cp, hs.K), st);
    } else if (quad_eval) {
        QuadArgs qa;
        memset(&qa, 0, sizeof(qa));
        qa.f = a; qa.Mg = (const double *)(ws + l.Mg); qa.rbuf = (double *)(ws + l.rbuf);
        qa.counter = (int *)(ws + l.counter); qa.P4 = qp.P4;
        lrc = launch_eval_quad(hs.KP, qp, qa, (double *)(ws + l.Mg), theta_ref, st);
    } else if (quad) {
        QuadArgs qa;
        memset(&qa, 0, sizeof(qa));
        qa.f = a; qa.Mg = (const double *)(ws + l.Mg); qa.Mslot = (double *)(ws + l.Mslot);
        qa.rbuf = (double *)(ws + l.rbuf);
        qa.counter = (int *)(ws + l.counter); qa.P4 = qp.P4;
        qa.recenter_every = spec->recenter_every; qa.recenter_ratio = spec->recenter_ratio;
        qa.dbg = nullptr; qa.nb_buf = nullptr; qa.nb_bytes = 0;
        qa.Mpre = quad_pre ? qa.Mg : nullptr; qa.n_pre = quad_pre;
        qa.gram_harm = gram_harm;
        HIP_TRY(ctx, hipMemsetAsync(qa.counter, 0, sizeof(int), st));
        lrc = -1;
        if (spec->converge == TSF_CONVERGE_MAP && qp.PPL == 1 && ctx->opt[TSF_OPT_MAP_DIRECT] != 0) {
            // converge = MAP where the posterior is a quadratic form in everything but sigma: the estimate itself by
            // alternating exact minimisations (tsf_map_quad.h) -- no L-BFGS trajectory, no continuation
            qa.f.map_max_iter = spec->map_max_iter; qa.f.map_tol = spec->map_tol;
            lrc = launch_map_quad(hs.KP, qp, qa, (double *)(ws + l.Mg), fit_P(hs.n_cp, hs.K) | 1, st);
            if (lrc == 0) map_done = true;
        }
        if (lrc == -1) lrc = launch_quad(hs.KP, qp, qa, (double *)(ws + l.Mg), st);
    } else if (mp.on) {
        MfmaTabs mt;
        memset(&mt, 0, sizeof(mt));
        int *flags = (int *)(ws + l.counter);        // [0] work queue head, [1] changepoint-row overflow
        HIP_TRY(ctx, hipMemsetAsync(flags, 0, 2 * sizeof(int), st));
        mt.XF = (const double *)(ws + l.mXF); mt.XB = (const double *)(ws + l.mXB);
        mt.XT = (const double *)(ws + l.mXT); mt.tq = (const double *)(ws + l.mtq);
        mt.cq = (const uint16_t *)(ws + l.mcq); mt.cpof = (const int8_t *)(ws + l.mcpof);
        mt.yq = (const double *)(ws + l.myq); mt.hist = (double *)(ws + l.mhist);
        mt.counter = flags; mt.overflow = flags + 1;
        mt.NG = mp.NG; mt.KF = mp.KF; mt.NCB = mp.NCB; mt.rr0 = mp.rr0; mt.run_if_overflow = 0;
        lrc = launch_mfma_layout(a, hs.KP, mt, (double *)(ws + l.mXF), (double *)(ws + l.mXB),
                                 (double *)(ws + l.mXT), (double *)(ws + l.mtq), (uint16_t *)(ws + l.mcq),
                                 (int8_t *)(ws + l.mcpof), (double *)(ws + l.myq), flags + 1, st);
        if (lrc == 0) {
            if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(ctx->ev0[slot], st));   // the fit kernel proper
            lrc = launch_mfma(hs.KP, hs.growth, mode, a, mt, mp.blocks, st);
        }
        if (lrc == 0) {
            // fallback, decided on the device: runs only if the layout kernel found a chunk with more
            // changepoint rows than the trend block holds (long runs of equal timestamps)
            FitArgs fb = a;
            fb.run_flag = flags + 1; fb.run_if = 1;
            lrc = pick_launch(hs.growth, mode)(hs.KP, fb, 0, st);
        }
    } else {
        if (coop) {
            a.coop_ctl = (int *)(ws + l.counter);
            a.coop_list = (int32_t *)(ws + l.clist);
            a.coop_slots = (double *)(ws + l.cslots);
            a.coop_max = coop_slots; a.coop_stride = coop_stride;
            a.coop_after = coop_after;
            a.coop_blocks = ctx->n_cu;
            {
                // Hand-over point of the tail rule: the one-wave kernel's waves run alone on their SIMDs by then (13.6 us per
                // evaluation) and a cooperative workgroup needs 5.05: up to ~2.7 fits per CU the cooperative kernel has
                // the higher aggregate rate even though they queue for its workgroups.  Default 2 per CU.
                const int pct = ctx->opt[TSF_OPT_COOP_TAIL] > 0 ? ctx->opt[TSF_OPT_COOP_TAIL] : 200;
                a.coop_tail_at = (int)((int64_t)ctx->n_cu * (pct > 400 ? 400 : pct) / 100);
                if (a.coop_tail_at < 1) a.coop_tail_at = 1;
            }
            HIP_TRY(ctx, hipMemsetAsync(a.coop_ctl, 0, 4 * sizeof(int), st));
        }
        lrc = pick_launch(hs.growth, mode)(hs.KP, a, theta_in != nullptr, st);
    }
    // converge = MAP: from where the optimiser's own tests stopped every fit on to the maximum a posteriori estimate
    // (tsf_map_kernels.h), on the same stream behind whichever kernels ran the fit; inside the profiled interval
    if (lrc == 0 && spec->converge == TSF_CONVERGE_MAP && theta_in == nullptr && !map_done) {
        a.map_max_iter = spec->map_max_iter; a.map_tol = spec->map_tol; a.map_harm = map_harm;
        if (map_harm && lat_U == 0) a.Bw = (const double *)(ws + l.Bw);
        if (map_harm && lat_U > 0) a.Bu = (const double *)(ws + l.Bw);
        a.order = nullptr; a.run_flag = nullptr;
        lrc = pick_map_launch(hs.growth, mode)(hs.KP, a, st);
    }
    if (ctx->profiling) { HIP_TRY(ctx, hipEventRecord(ctx->ev1[slot], st)); ctx->ev_count++; }
    if (order_buf >= 0) {       // the launch above reads the hint buffer: tsf_set_cost_hints waits for this before rewriting it
        if (!ctx->order_ev[order_buf]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->order_ev[order_buf], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventRecord(ctx->order_ev[order_buf], st));
        ctx->order_busy[order_buf] = 1;
    }
    if (lrc != 0) {
        ctx->err = std::string("kernel launch failed: ") + (lrc > 0 ? hipGetErrorString((hipError_t)lrc) : "no kernel for this shape");
        return -2;
    }
    return 0;
}

extern "C" int tsf_fit_aligned_dev(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int32_t T,
                                   const int64_t *ds, const void *y, int32_t y_dtype,
                                   const double *floor_, const double *cap, const double *extra,
                                   tsf_fit_out *out, void *stream)
{
    if (!ctx) return -1;
    if (!out || !out->theta || !out->y_scale || !out->fval || !out->status || !out->n_iter ||
        !out->n_eval || !out->grid)
        return fail(ctx, "tsf_fit_out has NULL members");
    return run_fit(ctx, spec, N, 1, T, nullptr, 0, T, ds, y, y_dtype, floor_, cap, extra, out,
                   nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int tsf_fit_ragged_dev(tsf_ctx *ctx, const tsf_spec *spec, int64_t N,
                                  const int64_t *offsets, int64_t total_rows, int32_t max_T,
                                  const int64_t *ds, const void *y, int32_t y_dtype,
                                  const double *floor_, const double *cap, const double *extra,
                                  tsf_fit_out *out, void *stream)
{
    if (!ctx) return -1;
    if (!offsets) return fail(ctx, "offsets is NULL");
    if (!out || !out->theta || !out->y_scale || !out->fval || !out->status || !out->n_iter ||
        !out->n_eval || !out->grid)
        return fail(ctx, "tsf_fit_out has NULL members");
    return run_fit(ctx, spec, N, 0, 0, offsets, total_rows, max_T, ds, y, y_dtype, floor_, cap,
                   extra, out, nullptr, nullptr, (hipStream_t)stream);
}

// ---- host-pointer wrappers --------------------------------------------------------------------

namespace {
// Device buffers of the host-pointer entry points come from a small per-process cache
// (power-of-two size classes, per device, at most 2 GiB kept): the DataFrame layer calls these
// entry points once per group of series, and a hipMalloc + hipFree (which synchronises the
// device) per buffer per call costs more than a small group's fit.
struct PoolEntry { void *p; size_t bytes; int device; };
std::mutex g_pool_mu;
std::vector<PoolEntry> g_pool;
size_t g_pool_bytes = 0;
constexpr size_t POOL_LIMIT = (size_t)2 << 30;

size_t pool_class(size_t n)
{
    size_t c = 256;
    while (c < n) c <<= 1;
    return c;
}

void pool_trim(int device)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size();) {
        if (g_pool[i].device == device) {
            hipFree(g_pool[i].p);
            g_pool_bytes -= g_pool[i].bytes;
            g_pool[i] = g_pool.back();
            g_pool.pop_back();
        } else {
            ++i;
        }
    }
}

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int device = 0;
    ~DevBuf()
    {
        if (!p) return;
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (g_pool_bytes + bytes <= POOL_LIMIT) { g_pool.push_back({p, bytes, device}); g_pool_bytes += bytes; }
        else hipFree(p);
    }
    hipError_t alloc(size_t n)
    {
        bytes = pool_class(n ? n : 8);
        hipGetDevice(&device);
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            for (size_t i = 0; i < g_pool.size(); ++i) {
                if (g_pool[i].device == device && g_pool[i].bytes == bytes) {
                    p = g_pool[i].p;
                    g_pool_bytes -= bytes;
                    g_pool[i] = g_pool.back();
                    g_pool.pop_back();
                    return hipSuccess;
                }
            }
        }
        return hipMalloc(&p, bytes);
    }
    template <class T> T *as() { return (T *)p; }
};

size_t ysize(int dt) { return dt == TSF_Y_F64 ? 8 : 4; }
}  // namespace

static int fit_host_one(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int aligned, int32_t T,
                        const int64_t *offsets, const int64_t *ds, const void *y, int32_t y_dtype,
                        const double *floor_, const double *cap, const double *extra,
                        tsf_fit_out *out, const double *theta_in, double *f_out, double *grad_out,
                        const double *theta_ref = nullptr)
{
    if (!ctx) return -1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (N <= 0) return fail(ctx, "N must be > 0");
    if (!spec) return fail(ctx, "spec is NULL");
    if (y_dtype < TSF_Y_F64 || y_dtype > TSF_Y_I32) return fail(ctx, "bad y_dtype");
    const int stride = tsf_theta_stride(spec);
    int64_t total = 0;
    int32_t max_T = 0;
    if (aligned) {
        total = (int64_t)N * T; max_T = T;
    } else {
        if (!offsets) return fail(ctx, "offsets is NULL");
        total = offsets[N] - offsets[0];
        if (offsets[0] != 0) return fail(ctx, "offsets[0] must be 0");
        for (int64_t n = 0; n < N; ++n) {
            const int64_t len = offsets[n + 1] - offsets[n];
            if (len < 0 || len > (int64_t)TSF_MAX_T) return fail(ctx, "series length out of range (0 .. TSF_MAX_T rows)");
            if (len > max_T) max_T = (int32_t)len;
        }
    }
    if (max_T < 1) return fail(ctx, "no rows");
    if (max_T > TSF_MAX_T) return fail(ctx, "series too long (TSF_MAX_T rows per series)");
    const int64_t n_ds = aligned ? T : total;
    const int64_t n_grids = aligned ? 1 : N;
    // TSF_HOST_TIMING=1 (dev): wall-clock of the phases of this entry point on stderr
    static const bool host_timing = getenv("TSF_HOST_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = host_timing ? now() : 0.0;
    double t_h2d = 0.0, t_fit = 0.0;
    DevBuf d_ds, d_y, d_off, d_floor, d_cap, d_extra, d_theta, d_ys, d_f, d_st, d_it, d_ev, d_grid, d_thin, d_grad, d_thref;
    HIP_TRY(ctx, d_ds.alloc(8 * n_ds));
    HIP_TRY(ctx, d_y.alloc(ysize(y_dtype) * total));
    HIP_TRY(ctx, hipMemcpy(d_ds.p, ds, 8 * n_ds, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(d_y.p, y, ysize(y_dtype) * total, hipMemcpyHostToDevice));
    if (!aligned) {
        HIP_TRY(ctx, d_off.alloc(8 * (N + 1)));
        HIP_TRY(ctx, hipMemcpy(d_off.p, offsets, 8 * (N + 1), hipMemcpyHostToDevice));
    }
    if (floor_) { HIP_TRY(ctx, d_floor.alloc(8 * N)); HIP_TRY(ctx, hipMemcpy(d_floor.p, floor_, 8 * N, hipMemcpyHostToDevice)); }
    if (cap) { HIP_TRY(ctx, d_cap.alloc(8 * N)); HIP_TRY(ctx, hipMemcpy(d_cap.p, cap, 8 * N, hipMemcpyHostToDevice)); }
    if (spec->n_extra > 0) {
        if (!extra) return fail(ctx, "extra columns declared but extra is NULL");
        const size_t nb = 8 * (size_t)spec->n_extra * n_ds;
        HIP_TRY(ctx, d_extra.alloc(nb));
        HIP_TRY(ctx, hipMemcpy(d_extra.p, extra, nb, hipMemcpyHostToDevice));
    }
    HIP_TRY(ctx, d_theta.alloc(8 * (size_t)N * stride));
    HIP_TRY(ctx, d_ys.alloc(8 * N)); HIP_TRY(ctx, d_f.alloc(8 * N));
    HIP_TRY(ctx, d_st.alloc(4 * N)); HIP_TRY(ctx, d_it.alloc(4 * N)); HIP_TRY(ctx, d_ev.alloc(4 * N));
    HIP_TRY(ctx, d_grid.alloc(sizeof(tsf_grid_info) * n_grids));
    HIP_TRY(ctx, hipMemset(d_st.p, 0, 4 * N)); HIP_TRY(ctx, hipMemset(d_it.p, 0, 4 * N));
    HIP_TRY(ctx, hipMemset(d_ev.p, 0, 4 * N));
    HIP_TRY(ctx, hipMemset(d_grid.p, 0, sizeof(tsf_grid_info) * n_grids));
    tsf_fit_out dout;
    dout.theta = d_theta.as<double>(); dout.y_scale = d_ys.as<double>(); dout.fval = d_f.as<double>();
    dout.status = d_st.as<int32_t>(); dout.n_iter = d_it.as<int32_t>(); dout.n_eval = d_ev.as<int32_t>();
    dout.grid = d_grid.as<tsf_grid_info>();
    if (theta_in) {
        HIP_TRY(ctx, d_thin.alloc(8 * (size_t)N * stride));
        HIP_TRY(ctx, hipMemcpy(d_thin.p, theta_in, 8 * (size_t)N * stride, hipMemcpyHostToDevice));
        HIP_TRY(ctx, d_grad.alloc(8 * (size_t)N * stride));
        if (theta_ref) {
            HIP_TRY(ctx, d_thref.alloc(8 * (size_t)N * stride));
            HIP_TRY(ctx, hipMemcpy(d_thref.p, theta_ref, 8 * (size_t)N * stride, hipMemcpyHostToDevice));
        }
    }
    // ragged panel: do all timestamps lie on one lattice base + u*step (the usual case: one
    // sampling grid, different start dates / lengths)?  Then the design rows are computed once
    // for the lattice and shared by every series instead of being stored per series.
    int64_t lat_base = 0, lat_step = 0, lat_U = 0;
    if (!aligned && !theta_in && spec->n_extra == 0 && total > 0) {
        int64_t lo = ds[0], hi = ds[0];
        for (int64_t i = 1; i < total; ++i) { lo = ds[i] < lo ? ds[i] : lo; hi = ds[i] > hi ? ds[i] : hi; }
        uint64_t g = 0;
        for (int64_t i = 0; i < total && g != 1; ++i) {
            uint64_t v = (uint64_t)(ds[i] - lo);
            while (v) { const uint64_t r = g % v; g = v; v = r; }      // gcd(g, v), gcd(0, v) = v
        }
        if (g > 0) {
            const uint64_t U = (uint64_t)(hi - lo) / g + 1;
            if (U <= (uint64_t)1 << 18) { lat_base = lo; lat_step = (int64_t)g; lat_U = (int64_t)U; }
        }
    }
    // ragged panel: which series SHARE a timestamp vector (the usual case: many series observed on a few calendars)?
    // Their grid tables -- scaled time, changepoint segments, the design matrix: 172 KB for two years of daily data --
    // are then built once per distinct vector and shared as on an aligned panel (FitArgs::grid_of), and the
    // quadratic-form path builds Z^T Z once per vector instead of once per series.  Identical vectors only (hash of
    // the bytes, then memcmp against the class's first member); models with explicit columns are left alone (their
    // columns are per row, not per timestamp).  TSF_GRID_SHARE=0 turns it off (tests compare both).
    DevBuf d_gof, d_grows, d_gord;
    int64_t n_distinct = 0;
    {
        if (!aligned && !theta_in && spec->n_extra == 0 && N >= 2 && ctx->opt[TSF_OPT_GRID_SHARE] != 0) {
            struct Key { int64_t len; uint64_t h; int64_t n; };
            std::vector<Key> keys((size_t)N);
            {
                int hw = (int)std::thread::hardware_concurrency();
                const int nt = hw < 1 ? 1 : (hw > 16 ? 16 : hw);
                std::atomic<int64_t> next(0);
                auto work = [&]() {
                    for (;;) {
                        const int64_t b = next.fetch_add(256);
                        if (b >= N) break;
                        for (int64_t n = b; n < N && n < b + 256; ++n) {
                            const int64_t len = offsets[n + 1] - offsets[n];
                            const uint64_t *p = (const uint64_t *)(ds + offsets[n]);
                            uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)len;
                            for (int64_t i = 0; i < len; ++i) { h ^= p[i]; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; }
                            keys[(size_t)n] = Key{len, h, n};
                        }
                    }
                };
                if (nt <= 1 || N < 1024) work();
                else {
                    std::vector<std::thread> th;
                    for (int i = 0; i < nt; ++i) th.emplace_back(work);
                    for (auto &x : th) x.join();
                }
            }
            std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
                return a.len != b.len ? a.len < b.len : (a.h != b.h ? a.h < b.h : a.n < b.n);
            });
            std::vector<int64_t> rep((size_t)N);            // first member of the class of series n
            for (size_t i = 0; i < keys.size();) {
                size_t j = i;
                while (j < keys.size() && keys[j].len == keys[i].len && keys[j].h == keys[i].h) ++j;
                // members of one (length, hash) run: classes by memcmp against the representatives found so far
                std::vector<int64_t> reps;
                for (size_t k = i; k < j; ++k) {
                    const int64_t n = keys[k].n;
                    int64_t r = -1;
                    for (int64_t c : reps)
                        if (memcmp(ds + offsets[c], ds + offsets[n], (size_t)keys[k].len * 8) == 0) { r = c; break; }
                    if (r < 0) { reps.push_back(n); r = n; }
                    rep[(size_t)n] = r;
                }
                i = j;
            }
            std::vector<int32_t> gof((size_t)N), id_of((size_t)N, -1);
            std::vector<int64_t> grows;
            for (int64_t n = 0; n < N; ++n) {
                const int64_t r = rep[(size_t)n];
                if (id_of[(size_t)r] < 0) {
                    id_of[(size_t)r] = (int32_t)(grows.size() / 2);
                    grows.push_back(offsets[r]);
                    grows.push_back(offsets[r + 1] - offsets[r]);
                }
                gof[(size_t)n] = id_of[(size_t)r];
            }
            n_distinct = (int64_t)(grows.size() / 2);
            if (n_distinct < N) {
                HIP_TRY(ctx, d_gof.alloc(4 * (size_t)N));
                HIP_TRY(ctx, d_grows.alloc(8 * grows.size()));
                HIP_TRY(ctx, hipMemcpy(d_gof.p, gof.data(), 4 * (size_t)N, hipMemcpyHostToDevice));
                HIP_TRY(ctx, hipMemcpy(d_grows.p, grows.data(), 8 * grows.size(), hipMemcpyHostToDevice));
                // The order in which the launch starts the series: grouped by grid, and the grouped list cut into 8
                // segments dealt out one position each in turn -- block b of a one-wave launch runs on XCD b % 8
                // (observed, MI355X_MICROARCH: for speed only), so every XCD works its way through ITS segment and its
                // 4 MB of L2 hold the two or three grids its resident blocks are on, instead of a share of all of them
                // (10 000 series on 91 grids, the reference's model: 205 -> 172 ms, tools/bench_ragged.py).  Used
                // when the caller gave no cost hints; results do not depend on the order.
                std::vector<int32_t> byg((size_t)N), ord((size_t)N);
                {
                    std::vector<int64_t> start((size_t)n_distinct + 1, 0);
                    for (int64_t n = 0; n < N; ++n) start[(size_t)gof[(size_t)n] + 1]++;
                    for (int64_t g = 0; g < n_distinct; ++g) start[(size_t)g + 1] += start[(size_t)g];
                    for (int64_t n = 0; n < N; ++n) byg[(size_t)start[(size_t)gof[(size_t)n]]++] = (int32_t)n;
                    const int64_t seg = (N + 7) / 8;
                    int64_t q = 0;
                    for (int64_t i = 0; i < seg; ++i)
                        for (int x = 0; x < 8; ++x) { const int64_t k = (int64_t)x * seg + i; if (k < N) ord[(size_t)q++] = byg[(size_t)k]; }
                }
                HIP_TRY(ctx, d_gord.alloc(4 * (size_t)N));
                HIP_TRY(ctx, hipMemcpy(d_gord.p, ord.data(), 4 * (size_t)N, hipMemcpyHostToDevice));
            } else {
                n_distinct = 0;
            }
        }
    }
    if (host_timing) t_h2d = now();
    int rc = run_fit(ctx, spec, N, aligned, T, d_off.as<int64_t>(), total, max_T, d_ds.as<int64_t>(),
                     d_y.p, y_dtype, floor_ ? d_floor.as<double>() : nullptr,
                     cap ? d_cap.as<double>() : nullptr,
                     spec->n_extra > 0 ? d_extra.as<double>() : nullptr, &dout,
                     theta_in ? d_thin.as<double>() : nullptr,
                     theta_in ? d_grad.as<double>() : nullptr, nullptr, lat_base, lat_step, lat_U,
                     (theta_in && theta_ref) ? d_thref.as<double>() : nullptr,
                     n_distinct > 0 ? d_gof.as<int32_t>() : nullptr, n_distinct > 0 ? d_grows.as<int64_t>() : nullptr, n_distinct,
                     n_distinct > 0 ? d_gord.as<int32_t>() : nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (host_timing) t_fit = now();
    if (theta_in) {
        HIP_TRY(ctx, hipMemcpy(f_out, d_f.p, 8 * N, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(grad_out, d_grad.p, 8 * (size_t)N * stride, hipMemcpyDeviceToHost));
        return 0;
    }
    HIP_TRY(ctx, hipMemcpy(out->theta, d_theta.p, 8 * (size_t)N * stride, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out->y_scale, d_ys.p, 8 * N, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out->fval, d_f.p, 8 * N, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out->status, d_st.p, 4 * N, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out->n_iter, d_it.p, 4 * N, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out->n_eval, d_ev.p, 4 * N, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out->grid, d_grid.p, sizeof(tsf_grid_info) * n_grids, hipMemcpyDeviceToHost));
    if (host_timing)
        fprintf(stderr, "[host-timing] N %lld: alloc + H2D + clears %.3f ms, fit (launch .. device idle) %.3f ms, D2H %.3f ms\n",
                (long long)N, t_h2d - t_start, t_fit - t_h2d, now() - t_fit);
    return 0;
}


// Ragged panels by length.  The device tables of a ragged call are laid out [series][NTmax] (NTmax = steps of
// the LONGEST series of the call: every other series' rows are padding), so one 100 000-row series among
// 10 000 series of 700 rows would ask for ~224 GB.  When the padding is more than half of the layout (and the
// layout is not small anyway) the host entry point therefore cuts the call into length classes -- the longest
// series and every series at least half as long, then the same again for the rest: at most ~11 classes, each
// padded by < 2 x -- and runs one fit call per class on the gathered rows: workspace proportional to the rows
// that exist.  Every series is fitted by itself, so the results do not depend on the grouping (GPU test:
// bit-identical to the single call).  Scheduling hints (tsf_set_cost_hints) given for the whole call are handed to
// the classes series by series.  tsf_fit_ragged_dev (device pointers: the lengths are not on the host) leaves
// this to its caller.
static int fit_host(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int aligned, int32_t T,
                    const int64_t *offsets, const int64_t *ds, const void *y, int32_t y_dtype,
                    const double *floor_, const double *cap, const double *extra,
                    tsf_fit_out *out, const double *theta_in, double *f_out, double *grad_out,
                    const double *theta_ref = nullptr)
{
    if (!ctx) return -1;
    // (anything fit_host_one rejects -- NULL inputs, offsets that do not start at 0 -- goes to it for the error: the
    // split below indexes ds / y / extra with the caller's offsets; round-4 advice)
    if (aligned || theta_in || !offsets || !spec || N < 2 || y_dtype < TSF_Y_F64 || y_dtype > TSF_Y_I32 ||
        !ds || !y || !out || offsets[0] != 0 || (spec->n_extra > 0 && !extra))
        return fit_host_one(ctx, spec, N, aligned, T, offsets, ds, y, y_dtype, floor_, cap, extra, out, theta_in,
                            f_out, grad_out, theta_ref);
    int64_t sum_nt = 0, nt_max = 0;
    for (int64_t n = 0; n < N; ++n) {
        const int64_t len = offsets[n + 1] - offsets[n];
        if (len < 0 || len > (int64_t)TSF_MAX_T)
            return fit_host_one(ctx, spec, N, aligned, T, offsets, ds, y, y_dtype, floor_, cap, extra, out,
                                theta_in, f_out, grad_out, theta_ref);      // (reports the error)
        const int64_t nt = (len + W - 1) / W;
        sum_nt += nt;
        if (nt > nt_max) nt_max = nt;
    }
    const int force = ctx->opt[TSF_OPT_RAGGED_SPLIT];       // 0: never, 1: whenever the padding rule says so (tests)
    const double padded = (double)N * (double)nt_max;
    const bool small = padded * 512.0 * (3 + tsf_spec_K(spec)) < 256e6;
    if (force == 0 || padded <= 2.0 * (double)sum_nt || (small && force != 1))
        return fit_host_one(ctx, spec, N, aligned, T, offsets, ds, y, y_dtype, floor_, cap, extra, out, theta_in,
                            f_out, grad_out, theta_ref);
    std::vector<int64_t> order((size_t)N);
    for (int64_t n = 0; n < N; ++n) order[(size_t)n] = n;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        return offsets[a + 1] - offsets[a] > offsets[b + 1] - offsets[b];
    });
    const int stride = tsf_theta_stride(spec);
    const size_t ys = (y_dtype == TSF_Y_F64) ? 8 : 4;
    const int64_t total = offsets[N] - offsets[0];
    // scheduling hints given for this call (tsf_set_cost_hints with N entries): every class gets its series' share
    // (round-4 advice: they used to be dropped on this path without a word)
    std::vector<int32_t> hints;
    if (ctx->order_n == N && (int64_t)ctx->cost_host.size() == N) hints = ctx->cost_host;
    ctx->order_n = 0;
    std::vector<int32_t> hb;
    for (int64_t b0 = 0; b0 < N;) {
        const int64_t top = (offsets[order[(size_t)b0] + 1] - offsets[order[(size_t)b0]] + W - 1) / W;
        int64_t b1 = b0;
        while (b1 < N && 2 * ((offsets[order[(size_t)b1] + 1] - offsets[order[(size_t)b1]] + W - 1) / W) >= top) ++b1;
        const int64_t nb = b1 - b0;
        // the class in the caller's order (stable: neighbours stay neighbours)
        std::vector<int64_t> idx(order.begin() + b0, order.begin() + b1);
        std::sort(idx.begin(), idx.end());
        std::vector<int64_t> off((size_t)nb + 1, 0);
        for (int64_t i = 0; i < nb; ++i) off[(size_t)i + 1] = off[(size_t)i] + (offsets[idx[(size_t)i] + 1] - offsets[idx[(size_t)i]]);
        const int64_t rows = off[(size_t)nb];
        std::vector<int64_t> dsb((size_t)rows);
        std::vector<char> yb((size_t)rows * ys);
        std::vector<double> flb, cpb, exb;
        if (floor_) flb.resize((size_t)nb);
        if (cap) cpb.resize((size_t)nb);
        if (spec->n_extra > 0 && extra) exb.resize((size_t)spec->n_extra * (size_t)rows);
        for (int64_t i = 0; i < nb; ++i) {
            const int64_t n = idx[(size_t)i], r0 = offsets[n], len = offsets[n + 1] - r0;
            if (len > 0) {
                memcpy(dsb.data() + off[(size_t)i], ds + r0, (size_t)len * 8);
                memcpy(yb.data() + (size_t)off[(size_t)i] * ys, (const char *)y + (size_t)r0 * ys, (size_t)len * ys);
                for (int e = 0; e < spec->n_extra && !exb.empty(); ++e)
                    memcpy(exb.data() + (size_t)e * rows + off[(size_t)i], extra + (size_t)e * total + r0, (size_t)len * 8);
            }
            if (floor_) flb[(size_t)i] = floor_[n];
            if (cap) cpb[(size_t)i] = cap[n];
        }
        std::vector<double> th((size_t)nb * stride), ysc((size_t)nb), fv((size_t)nb);
        std::vector<int32_t> st((size_t)nb), it((size_t)nb), ev((size_t)nb);
        std::vector<tsf_grid_info> gr((size_t)nb);
        tsf_fit_out ob;
        ob.theta = th.data(); ob.y_scale = ysc.data(); ob.fval = fv.data(); ob.status = st.data();
        ob.n_iter = it.data(); ob.n_eval = ev.data(); ob.grid = gr.data();
        if (!hints.empty()) {
            hb.resize((size_t)nb);
            for (int64_t i = 0; i < nb; ++i) hb[(size_t)i] = hints[(size_t)idx[(size_t)i]];
            const int hrc = tsf_set_cost_hints(ctx, hb.data(), nb);
            if (hrc) return hrc;
        }
        const int rc = fit_host_one(ctx, spec, nb, 0, 0, off.data(), dsb.data(), yb.data(), y_dtype,
                                    floor_ ? flb.data() : nullptr, cap ? cpb.data() : nullptr,
                                    exb.empty() ? (spec->n_extra > 0 ? extra : nullptr) : exb.data(), &ob, nullptr, nullptr, nullptr);
        if (rc) return rc;
        for (int64_t i = 0; i < nb; ++i) {
            const int64_t n = idx[(size_t)i];
            memcpy(out->theta + (size_t)n * stride, th.data() + (size_t)i * stride, sizeof(double) * stride);
            out->y_scale[n] = ysc[(size_t)i]; out->fval[n] = fv[(size_t)i]; out->status[n] = st[(size_t)i];
            out->n_iter[n] = it[(size_t)i]; out->n_eval[n] = ev[(size_t)i]; out->grid[n] = gr[(size_t)i];
        }
        b0 = b1;
    }
    return 0;
}

extern "C" int tsf_fit_aligned(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int32_t T,
                               const int64_t *ds, const void *y, int32_t y_dtype,
                               const double *floor_, const double *cap, const double *extra,
                               tsf_fit_out *out)
{
    if (!ctx) return -1;
    if (!ds || !y || !out) return fail(ctx, "NULL input");
    return fit_host(ctx, spec, N, 1, T, nullptr, ds, y, y_dtype, floor_, cap, extra, out, nullptr,
                    nullptr, nullptr);
}

extern "C" int tsf_fit_ragged(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, const int64_t *offsets,
                              const int64_t *ds, const void *y, int32_t y_dtype,
                              const double *floor_, const double *cap, const double *extra,
                              tsf_fit_out *out)
{
    if (!ctx) return -1;
    if (!ds || !y || !out) return fail(ctx, "NULL input");
    return fit_host(ctx, spec, N, 0, 0, offsets, ds, y, y_dtype, floor_, cap, extra, out, nullptr,
                    nullptr, nullptr);
}

extern "C" int tsf_eval(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int32_t T, const int64_t *ds,
                        const void *y, int32_t y_dtype, const double *floor_, const double *cap,
                        const double *extra, const double *theta, double *f_out, double *grad_out)
{
    if (!ctx) return -1;
    if (!ds || !y || !theta || !f_out || !grad_out) return fail(ctx, "NULL input");
    tsf_fit_out dummy;
    memset(&dummy, 0, sizeof(dummy));
    return fit_host(ctx, spec, N, 1, T, nullptr, ds, y, y_dtype, floor_, cap, extra, &dummy, theta,
                    f_out, grad_out);
}

extern "C" int tsf_eval_quadratic(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int32_t T, const int64_t *ds,
                                  const void *y, int32_t y_dtype, const double *extra, const double *theta_ref,
                                  const double *theta, double *f_out, double *grad_out)
{
    if (!ctx) return -1;
    if (!ds || !y || !theta || !theta_ref || !f_out || !grad_out) return fail(ctx, "NULL input");
    tsf_fit_out dummy;
    memset(&dummy, 0, sizeof(dummy));
    return fit_host(ctx, spec, N, 1, T, nullptr, ds, y, y_dtype, nullptr, nullptr, extra, &dummy, theta,
                    f_out, grad_out, theta_ref);
}

// ---- predict ----------------------------------------------------------------------------------

// predict_kernel on the series [n0, n0 + n) of the caller's arrays.  A shared future grid gets its
// design table built first (once per call: tab_ready), in the context's own buffer.
static int launch_predict(tsf_ctx *ctx, const DevSpec &hs, PredictArgs a, int64_t n0, int64_t n, bool *tab_ready,
                          hipStream_t st)
{
    if (a.shared_future) {
        const size_t need = sizeof(double) * (size_t)(hs.K > 0 ? hs.K : 1) * a.H;
        if (ctx->fut_tab_bytes < need) {
            // (grows only: a buffer an earlier call on another stream may still read is never freed here
            // without the device being idle -- hipFree synchronises)
            if (ctx->fut_tab) { HIP_TRY(ctx, hipFree(ctx->fut_tab)); ctx->fut_tab = nullptr; ctx->fut_tab_bytes = 0; }
            HIP_TRY(ctx, hipMalloc((void **)&ctx->fut_tab, need));
            ctx->fut_tab_bytes = need;
        }
        if (!*tab_ready && hs.n_seas > 0) {
            const int work = a.H * hs.n_seas;
            hipLaunchKernelGGL(future_design_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st,
                               ctx->d_spec, a.H, a.ds_future, ctx->fut_tab);
            HIP_TRY(ctx, hipGetLastError());
        }
        *tab_ready = true;
        a.Xf = ctx->fut_tab;
    }
    const int64_t g = a.n_grids == 1 ? 0 : n0;
    a.N = n;
    a.theta += (size_t)n0 * a.theta_stride; a.y_scale += n0; a.grid += g;
    if (!a.shared_future) a.ds_future += (size_t)n0 * a.H;
    if (a.floor_) a.floor_ += n0;
    if (a.cap) a.cap += n0;
    if (a.extra_future && !a.shared_future) a.extra_future += (size_t)n0 * hs.n_extra * a.H;
    a.yhat += (size_t)n0 * a.H;
    if (a.yhat_int) a.yhat_int += (size_t)n0 * a.H;
    hipLaunchKernelGGL(predict_kernel, dim3((unsigned)((n + PREDICT_WAVES - 1) / PREDICT_WAVES)), dim3(PREDICT_WAVES * 64), 0, st, a);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

extern "C" int tsf_predict_dev(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int32_t H,
                               const double *theta, const double *y_scale,
                               const tsf_grid_info *grid, int32_t n_grids, const int64_t *ds_future,
                               int32_t shared_future, const double *floor_, const double *cap,
                               const double *extra_future, double *yhat, int32_t *yhat_int,
                               void *stream)
{
    if (!ctx) return -1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (N <= 0 || H <= 0) return fail(ctx, "N and H must be > 0");
    if (!theta || !y_scale || !grid || !ds_future || !yhat) return fail(ctx, "NULL input");
    if (n_grids != 1 && n_grids != N) return fail(ctx, "n_grids must be 1 or N");
    DevSpec hs;
    int mode = 0;
    int rc = build_devspec(ctx, spec, &hs, &mode);
    if (rc) return rc;
    if (hs.n_extra > 0 && !extra_future) return fail(ctx, "extra_future is NULL");
    if (hs.growth == TSF_GROWTH_LOGISTIC && !cap) return fail(ctx, "logistic growth needs cap");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_spec, &hs, sizeof(hs), hipMemcpyHostToDevice, st));
    PredictArgs a;
    memset(&a, 0, sizeof(a));
    a.sp = ctx->d_spec; a.N = N; a.H = H; a.theta_stride = tsf_theta_stride(spec);
    a.n_grids = n_grids; a.shared_future = shared_future; a.theta = theta; a.y_scale = y_scale;
    a.grid = grid; a.ds_future = ds_future; a.floor_ = floor_; a.cap = cap;
    a.extra_future = extra_future; a.yhat = yhat; a.yhat_int = yhat_int;
    bool tab_ready = false;
    return launch_predict(ctx, hs, a, 0, N, &tab_ready, st);
}

extern "C" int tsf_predict(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int32_t H,
                           const double *theta, const double *y_scale, const tsf_grid_info *grid,
                           int32_t n_grids, const int64_t *ds_future, int32_t shared_future,
                           const double *floor_, const double *cap, const double *extra_future,
                           double *yhat, int32_t *yhat_int)
{
    if (!ctx) return -1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (N <= 0 || H <= 0) return fail(ctx, "N and H must be > 0");
    if (!spec || !theta || !y_scale || !grid || !ds_future || !yhat) return fail(ctx, "NULL input");
    if (n_grids != 1 && n_grids != N) return fail(ctx, "n_grids must be 1 or N");
    const int stride = tsf_theta_stride(spec);
    const size_t nfut = shared_future ? (size_t)H : (size_t)N * H;
    DevBuf d_th, d_ys, d_grid, d_ds, d_fl, d_cap, d_ex, d_yh, d_yi;
    HIP_TRY(ctx, d_th.alloc(8 * (size_t)N * stride));
    HIP_TRY(ctx, hipMemcpy(d_th.p, theta, 8 * (size_t)N * stride, hipMemcpyHostToDevice));
    HIP_TRY(ctx, d_ys.alloc(8 * N));
    HIP_TRY(ctx, hipMemcpy(d_ys.p, y_scale, 8 * N, hipMemcpyHostToDevice));
    HIP_TRY(ctx, d_grid.alloc(sizeof(tsf_grid_info) * n_grids));
    HIP_TRY(ctx, hipMemcpy(d_grid.p, grid, sizeof(tsf_grid_info) * n_grids, hipMemcpyHostToDevice));
    HIP_TRY(ctx, d_ds.alloc(8 * nfut));
    HIP_TRY(ctx, hipMemcpy(d_ds.p, ds_future, 8 * nfut, hipMemcpyHostToDevice));
    if (floor_) { HIP_TRY(ctx, d_fl.alloc(8 * N)); HIP_TRY(ctx, hipMemcpy(d_fl.p, floor_, 8 * N, hipMemcpyHostToDevice)); }
    if (cap) { HIP_TRY(ctx, d_cap.alloc(8 * N)); HIP_TRY(ctx, hipMemcpy(d_cap.p, cap, 8 * N, hipMemcpyHostToDevice)); }
    if (spec->n_extra > 0) {
        if (!extra_future) return fail(ctx, "extra_future is NULL");
        const size_t nb = 8 * (size_t)spec->n_extra * nfut;
        HIP_TRY(ctx, d_ex.alloc(nb));
        HIP_TRY(ctx, hipMemcpy(d_ex.p, extra_future, nb, hipMemcpyHostToDevice));
    }
    HIP_TRY(ctx, d_yh.alloc(8 * (size_t)N * H));
    if (yhat_int) HIP_TRY(ctx, d_yi.alloc(4 * (size_t)N * H));
    int rc = tsf_predict_dev(ctx, spec, N, H, d_th.as<double>(), d_ys.as<double>(),
                             d_grid.as<tsf_grid_info>(), n_grids, d_ds.as<int64_t>(), shared_future,
                             floor_ ? d_fl.as<double>() : nullptr, cap ? d_cap.as<double>() : nullptr,
                             spec->n_extra > 0 ? d_ex.as<double>() : nullptr, d_yh.as<double>(),
                             yhat_int ? d_yi.as<int32_t>() : nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpy(yhat, d_yh.p, 8 * (size_t)N * H, hipMemcpyDeviceToHost));
    if (yhat_int) HIP_TRY(ctx, hipMemcpy(yhat_int, d_yi.p, 4 * (size_t)N * H, hipMemcpyDeviceToHost));
    return 0;
}

// ---- uncertainty intervals ------------------------------------------------------------------------

extern "C" int tsf_predict_intervals_dev(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int32_t H,
                                         const double *theta, const double *y_scale,
                                         const tsf_grid_info *grid, int32_t n_grids,
                                         const int64_t *ds_future, int32_t shared_future,
                                         const double *floor_, const double *cap,
                                         const double *extra_future, const int64_t *series_key,
                                         int32_t n_samples, double interval_width, uint64_t seed,
                                         double *yhat, double *yhat_lower, double *yhat_upper, void *stream)
{
    if (!ctx) return -1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (N <= 0 || H <= 0) return fail(ctx, "N and H must be > 0");
    if (!spec || !theta || !y_scale || !grid || !ds_future || !yhat || !yhat_lower || !yhat_upper) return fail(ctx, "NULL input");
    if (n_grids != 1 && n_grids != N) return fail(ctx, "n_grids must be 1 or N");
    if (n_samples < 2 || n_samples > 4096) return fail(ctx, "n_samples must be in [2, 4096]");
    if (!(interval_width > 0.0 && interval_width < 1.0)) return fail(ctx, "interval_width must be in (0, 1)");
    DevSpec hs;
    int mode = 0;
    int rc = build_devspec(ctx, spec, &hs, &mode);
    if (rc) return rc;
    if (hs.n_extra > 0 && !extra_future) return fail(ctx, "extra_future is NULL");
    if (hs.growth == TSF_GROWTH_LOGISTIC && !cap) return fail(ctx, "logistic growth needs cap");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_spec, &hs, sizeof(hs), hipMemcpyHostToDevice, st));
    // Series are processed in chunks sized so that ALL scratch of a chunk -- the per-row pieces of the
    // point forecast (t, additive term, 1 + multiplicative term) and the samples -- stays within
    // 512 MB.  The scratch is ONE cached block of the context (like the fit workspace: grown on demand by a
    // plain hipMalloc, which waits for the device only on the call that grows it); the chunks of a call and
    // successive calls reuse it in stream order, so in the steady state nothing here waits for the device, as
    // the `_dev` contract (include/tsf.h) promises.  (Round 3 used hipMallocAsync / hipFreeAsync here; see
    // tsf_create.)  As with the workspace: one stream at a time per context.
    const size_t per_series = (size_t)H * 8 * (3 + (size_t)n_samples);
    int64_t chunk = (int64_t)(((size_t)512 << 20) / per_series);
    if (chunk < 1) chunk = 1;
    if (chunk > N) chunk = N;
    const size_t nh = (size_t)chunk * H;
    {
        const size_t need = 8 * nh * (3 + (size_t)n_samples);
        if (ctx->iv_ws_bytes < need) {
            if (ctx->iv_ws) { HIP_TRY(ctx, hipFree(ctx->iv_ws)); ctx->iv_ws = nullptr; ctx->iv_ws_bytes = 0; }
            HIP_TRY(ctx, hipMalloc(&ctx->iv_ws, need));
            ctx->iv_ws_bytes = need;
        }
    }
    double *d_t = (double *)ctx->iv_ws, *d_xa = d_t + nh, *d_opm = d_xa + nh, *d_samp = d_opm + nh;
    auto release = [&]() {};
#define HIP_TRY_REL(expr)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                     \
            release();                                                                        \
            return -2;                                                                        \
        }                                                                                     \
    } while (0)
    PredictArgs p;
    memset(&p, 0, sizeof(p));
    p.sp = ctx->d_spec; p.N = N; p.H = H; p.theta_stride = tsf_theta_stride(spec);
    p.n_grids = n_grids; p.shared_future = shared_future; p.theta = theta; p.y_scale = y_scale;
    p.grid = grid; p.ds_future = ds_future; p.floor_ = floor_; p.cap = cap;
    p.extra_future = extra_future; p.yhat = yhat; p.yhat_int = nullptr;
    p.t_out = d_t; p.xa_out = d_xa; p.opm_out = d_opm;
    int NSP = 2;
    while (NSP < n_samples) NSP <<= 1;
    IntervalArgs a;
    memset(&a, 0, sizeof(a));
    a.sp = ctx->d_spec; a.H = H; a.theta_stride = tsf_theta_stride(spec); a.n_grids = n_grids; a.NS = n_samples;
    a.theta = theta; a.y_scale = y_scale; a.grid = grid; a.floor_ = floor_; a.cap = cap;
    a.series_key = series_key; a.seed = seed;
    a.lo_frac = (1.0 - interval_width) / 2.0; a.hi_frac = (1.0 + interval_width) / 2.0;
    a.samples = d_samp; a.lower = yhat_lower; a.upper = yhat_upper;
    bool tab_ready = false;
    for (int64_t n0 = 0; n0 < N; n0 += chunk) {
        const int64_t nc = (N - n0 < chunk) ? N - n0 : chunk;
        // the point forecast of the chunk and its per-row pieces (chunk-local scratch: rows of
        // series n0 + i at [i][H])
        {
            const int prc = launch_predict(ctx, hs, p, n0, nc, &tab_ready, st);
            if (prc) { release(); return prc; }
        }
        a.n0 = n0; a.n_chunk = nc;
        a.t = d_t - (size_t)n0 * H; a.xa = d_xa - (size_t)n0 * H; a.opm = d_opm - (size_t)n0 * H;
        hipLaunchKernelGGL(interval_sample_kernel, dim3((unsigned)a.n_chunk, (unsigned)((n_samples + 255) / 256)),
                           dim3(256), 0, st, a);
        HIP_TRY_REL(hipGetLastError());
        hipLaunchKernelGGL(interval_percentile_kernel, dim3((unsigned)(a.n_chunk * H)), dim3(256),
                           sizeof(double) * NSP, st, a, NSP);
        HIP_TRY_REL(hipGetLastError());
    }
#undef HIP_TRY_REL
    return 0;
}

extern "C" int tsf_predict_intervals(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int32_t H,
                                     const double *theta, const double *y_scale,
                                     const tsf_grid_info *grid, int32_t n_grids,
                                     const int64_t *ds_future, int32_t shared_future,
                                     const double *floor_, const double *cap,
                                     const double *extra_future, const int64_t *series_key,
                                     int32_t n_samples, double interval_width, uint64_t seed,
                                     double *yhat, double *yhat_lower, double *yhat_upper)
{
    if (!ctx) return -1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (N <= 0 || H <= 0) return fail(ctx, "N and H must be > 0");
    if (!spec || !theta || !y_scale || !grid || !ds_future || !yhat || !yhat_lower || !yhat_upper) return fail(ctx, "NULL input");
    if (n_grids != 1 && n_grids != N) return fail(ctx, "n_grids must be 1 or N");
    const int stride = tsf_theta_stride(spec);
    const size_t nfut = shared_future ? (size_t)H : (size_t)N * H;
    DevBuf d_th, d_ys, d_grid, d_ds, d_fl, d_cap, d_ex, d_key, d_yh, d_lo, d_hi;
    HIP_TRY(ctx, d_th.alloc(8 * (size_t)N * stride));
    HIP_TRY(ctx, hipMemcpy(d_th.p, theta, 8 * (size_t)N * stride, hipMemcpyHostToDevice));
    HIP_TRY(ctx, d_ys.alloc(8 * N));
    HIP_TRY(ctx, hipMemcpy(d_ys.p, y_scale, 8 * N, hipMemcpyHostToDevice));
    HIP_TRY(ctx, d_grid.alloc(sizeof(tsf_grid_info) * n_grids));
    HIP_TRY(ctx, hipMemcpy(d_grid.p, grid, sizeof(tsf_grid_info) * n_grids, hipMemcpyHostToDevice));
    HIP_TRY(ctx, d_ds.alloc(8 * nfut));
    HIP_TRY(ctx, hipMemcpy(d_ds.p, ds_future, 8 * nfut, hipMemcpyHostToDevice));
    if (floor_) { HIP_TRY(ctx, d_fl.alloc(8 * N)); HIP_TRY(ctx, hipMemcpy(d_fl.p, floor_, 8 * N, hipMemcpyHostToDevice)); }
    if (cap) { HIP_TRY(ctx, d_cap.alloc(8 * N)); HIP_TRY(ctx, hipMemcpy(d_cap.p, cap, 8 * N, hipMemcpyHostToDevice)); }
    if (series_key) { HIP_TRY(ctx, d_key.alloc(8 * N)); HIP_TRY(ctx, hipMemcpy(d_key.p, series_key, 8 * N, hipMemcpyHostToDevice)); }
    if (spec->n_extra > 0) {
        if (!extra_future) return fail(ctx, "extra_future is NULL");
        const size_t nb = 8 * (size_t)spec->n_extra * nfut;
        HIP_TRY(ctx, d_ex.alloc(nb));
        HIP_TRY(ctx, hipMemcpy(d_ex.p, extra_future, nb, hipMemcpyHostToDevice));
    }
    const size_t nh = 8 * (size_t)N * H;
    HIP_TRY(ctx, d_yh.alloc(nh)); HIP_TRY(ctx, d_lo.alloc(nh)); HIP_TRY(ctx, d_hi.alloc(nh));
    int rc = tsf_predict_intervals_dev(ctx, spec, N, H, d_th.as<double>(), d_ys.as<double>(),
                                       d_grid.as<tsf_grid_info>(), n_grids, d_ds.as<int64_t>(), shared_future,
                                       floor_ ? d_fl.as<double>() : nullptr, cap ? d_cap.as<double>() : nullptr,
                                       spec->n_extra > 0 ? d_ex.as<double>() : nullptr,
                                       series_key ? d_key.as<int64_t>() : nullptr, n_samples, interval_width, seed,
                                       d_yh.as<double>(), d_lo.as<double>(), d_hi.as<double>(), nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpy(yhat, d_yh.p, nh, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(yhat_lower, d_lo.p, nh, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(yhat_upper, d_hi.p, nh, hipMemcpyDeviceToHost));
    return 0;
}

// ---- diagnostics --------------------------------------------------------------------------------

extern "C" int tsf_design(tsf_ctx *ctx, const tsf_spec *spec, int32_t T, const int64_t *ds,
                          const double *extra, double *X_out, double *t_out,
                          tsf_grid_info *grid_out)
{
    if (!ctx) return -1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!spec || !ds || T < 2) return fail(ctx, "bad input");
    DevSpec hs;
    int mode = 0;
    int rc = build_devspec(ctx, spec, &hs, &mode);
    if (rc) return rc;
    if (hs.n_extra > 0 && !extra) return fail(ctx, "extra is NULL");
    const int NT = (T + W - 1) / W;
    const WsLayout l = ws_layout(1, 1, NT, hs.KP);
    rc = ensure_ws(ctx, l.total);
    if (rc) return rc;
    char *ws = (char *)ctx->ws;
    DevBuf d_ds, d_ex;
    HIP_TRY(ctx, d_ds.alloc(8 * (size_t)T));
    HIP_TRY(ctx, hipMemcpy(d_ds.p, ds, 8 * (size_t)T, hipMemcpyHostToDevice));
    if (hs.n_extra > 0) {
        HIP_TRY(ctx, d_ex.alloc(8 * (size_t)hs.n_extra * T));
        HIP_TRY(ctx, hipMemcpy(d_ex.p, extra, 8 * (size_t)hs.n_extra * T, hipMemcpyHostToDevice));
    }
    HIP_TRY(ctx, hipMemcpy(ctx->d_spec, &hs, sizeof(hs), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemset(ws + l.Xw, 0, sizeof(double) * (size_t)NT * hs.KP * W));
    HIP_TRY(ctx, hipMemset(ws + l.gtab, 0, sizeof(GridTab)));
    hipLaunchKernelGGL(setup_grid_kernel, dim3(1), dim3(256), 0, nullptr, ctx->d_spec, 1, nullptr, T,
                       d_ds.as<int64_t>(), d_ex.as<double>(), (int64_t)T, NT, (GridTab *)(ws + l.gtab),
                       (double *)(ws + l.tw), (uint16_t *)(ws + l.cw), (double *)(ws + l.Xw), nullptr,
                       (int64_t)0, (int64_t)0);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipDeviceSynchronize());
    std::vector<double> Xw((size_t)NT * hs.KP * W), tw((size_t)NT * W);
    GridTab gt;
    HIP_TRY(ctx, hipMemcpy(Xw.data(), ws + l.Xw, Xw.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(tw.data(), ws + l.tw, tw.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(&gt, ws + l.gtab, sizeof(gt), hipMemcpyDeviceToHost));
    for (int i = 0; i < T; ++i) {
        const int L = i / NT, q = i - L * NT;
        if (t_out) t_out[i] = tw[(size_t)q * W + L];
        if (X_out)
            for (int j = 0; j < hs.K; ++j)
                X_out[(size_t)i * hs.K + hs.perm[j]] = Xw[((size_t)q * hs.KP + j) * W + L];
    }
    if (grid_out) *grid_out = gt.info;
    return 0;
}

extern "C" int tsf_selftest_math(tsf_ctx *ctx, int32_t op, int64_t n, const double *a,
                                 const double *b, double *out)
{
    if (!ctx) return -1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n <= 0 || !a || !out) return fail(ctx, "bad input");
    DevBuf da, db, dout;
    HIP_TRY(ctx, da.alloc(8 * n)); HIP_TRY(ctx, dout.alloc(8 * n));
    HIP_TRY(ctx, hipMemcpy(da.p, a, 8 * n, hipMemcpyHostToDevice));
    if (b) { HIP_TRY(ctx, db.alloc(8 * n)); HIP_TRY(ctx, hipMemcpy(db.p, b, 8 * n, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, op, n,
                       da.as<double>(), b ? db.as<double>() : nullptr, dout.as<double>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpy(out, dout.p, 8 * n, hipMemcpyDeviceToHost));
    return 0;
}

// ---- measurement hooks --------------------------------------------------------------------------

extern "C" int tsf_set_cost_hints(tsf_ctx *ctx, const int32_t *cost, int64_t n)
{
    if (!ctx) return -1;
    ctx->order_n = 0;
    if (!cost || n <= 0) return 0;
    if (n > (int64_t)INT32_MAX) return fail(ctx, "tsf_set_cost_hints: too many series");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<int32_t> order((size_t)n);
    for (int64_t i = 0; i < n; ++i) order[(size_t)i] = (int32_t)i;
    std::stable_sort(order.begin(), order.end(), [cost](int32_t x, int32_t y) { return cost[x] > cost[y]; });
    const int b = ctx->order_next;
    if (ctx->order_busy[b]) {   // a fit two calls back may still be reading this buffer on its stream
        HIP_TRY(ctx, hipEventSynchronize(ctx->order_ev[b]));
        ctx->order_busy[b] = 0;
    }
    const size_t need = sizeof(int32_t) * (size_t)n;
    if (ctx->order_cap[b] < need) {
        if (ctx->order_dev[b]) { HIP_TRY(ctx, hipFree(ctx->order_dev[b])); ctx->order_dev[b] = nullptr; ctx->order_cap[b] = 0; }
        HIP_TRY(ctx, hipMalloc((void **)&ctx->order_dev[b], need));
        ctx->order_cap[b] = need;
    }
    HIP_TRY(ctx, hipMemcpy(ctx->order_dev[b], order.data(), need, hipMemcpyHostToDevice));
    ctx->order_next = b ^ 1;
    ctx->order_n = n;
    if (cost != ctx->cost_host.data()) ctx->cost_host.assign(cost, cost + n);
    return 0;
}

extern "C" int tsf_set_profiling(tsf_ctx *ctx, int32_t enable)
{
    if (!ctx) return -1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (enable && !ctx->ev_created) {
        for (int i = 0; i < TSF_PROFILE_RING; ++i) {
            HIP_TRY(ctx, hipEventCreate(&ctx->ev0[i]));
            HIP_TRY(ctx, hipEventCreate(&ctx->ev1[i]));
        }
        ctx->ev_created = 1;
    }
    ctx->profiling = enable ? 1 : 0;
    ctx->ev_count = 0;
    return 0;
}

extern "C" int tsf_profile_read(tsf_ctx *ctx, float *ms_out, int32_t max_n, int32_t *n_out)
{
    if (!ctx) return -1;
    if (!ms_out || !n_out) return fail(ctx, "NULL output");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    long n = ctx->ev_count < TSF_PROFILE_RING ? ctx->ev_count : TSF_PROFILE_RING;
    if (n > max_n) n = max_n;
    const long first = ctx->ev_count - n;
    for (long i = 0; i < n; ++i) {
        const int slot = (int)((first + i) % TSF_PROFILE_RING);
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev1[slot]));
        HIP_TRY(ctx, hipEventElapsedTime(&ms_out[i], ctx->ev0[slot], ctx->ev1[slot]));
    }
    *n_out = (int32_t)n;
    return 0;
}

extern "C" int tsf_last_fit_route(tsf_ctx *ctx, int32_t *sparse_columns)
{
    if (!ctx || !sparse_columns) return -1;
    *sparse_columns = 0;
    if (!ctx->last_sp_flag) return 0;
    int bad = 1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpy(&bad, ctx->last_sp_flag, sizeof(int), hipMemcpyDeviceToHost));
    *sparse_columns = bad ? 0 : 1;
    return 0;
}

extern "C" int tsf_last_fit_kernel_ms(tsf_ctx *ctx, float *ms_out)
{
    if (!ctx) return -1;
    if (!ms_out || ctx->ev_count == 0) return fail(ctx, "no profiled fit call recorded");
    int32_t n = 0;
    float tmp[TSF_PROFILE_RING];
    int rc = tsf_profile_read(ctx, tmp, TSF_PROFILE_RING, &n);
    if (rc) return rc;
    *ms_out = tmp[n - 1];
    return 0;
}
