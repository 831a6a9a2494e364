// tsf_inst_quad.hip -- instantiates the quadratic-form fit path (tsf_quad_kernels.h): ragged panels,
// the two-slot kernel, Newton.  Built with -mllvm -disable-machine-licm like tsf_inst_quad3.hip (the
// reason is at the top of that file: hoisted fp64 literals end up in scratch).
#include "tsf_quad_launch.h"
#include "tsf_newton_quad.h"
#include "tsf_newton_batch.h"

namespace tsf {

// the most waves per workgroup any variant for this PPL launches (workspace slots: tsf_api.hip quad_plan)
int quad_waves_per_block(int PPL) { return PPL == 2 ? (TSF_QUAD_NW2G > TSF_QUAD_NW2 ? TSF_QUAD_NW2G : TSF_QUAD_NW2) : TSF_QUAD_NW4; }
static_assert(TSF_QUAD_NW4 >= TSF_QUAD_NW3 && TSF_QUAD_NW4 >= TSF_QUAD_NW, "workspace slots are sized for the widest workgroup");

// ragged panel whose series share timestamp vectors (QuadArgs::Mpre): Z^T Z once per distinct vector, ahead of the
// fit kernel, for the kernels that read it (M in registers, M in global memory)
template <int KP, int PPL>
static int prebuild_grams(const QuadPlan &qp, const QuadArgs &qa, hipStream_t st)
{
    if (!qa.Mpre) return 0;
    const int64_t per = qp.slots / qp.P4 > 0 ? qp.slots / qp.P4 : 1;        // staging rows: one rbuf slot per (grid, column)
    for (int64_t g0 = 0; g0 < qa.n_pre; g0 += per) {
        const int64_t cnt = qa.n_pre - g0 < per ? qa.n_pre - g0 : per;
        hipLaunchKernelGGL((gram_grids_kernel<KP, PPL>), dim3((unsigned)qp.P4, (unsigned)cnt), dim3(64), 0, st,
                           qa, const_cast<double *>(qa.Mpre), g0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

// ragged panel, Z^T Z of every resident wave in LDS: NWR waves per workgroup (tsf_quad_kernels.h
// QM_RAGGED_LDS); -2 when it does not fit (the caller falls back to M in global memory)
template <int KP, int PQ>
static int launch_quad_ragged_lds(const QuadPlan &qp, const QuadArgs &qa, hipStream_t st)
{
    constexpr int NWR = 4;
    if (qp.P4 != PQ) return -2;
    const size_t lds = (quad_lanec_bytes<1>() + sizeof(QuadLds<KP, 1>) + sizeof(double) * ((size_t)qa.f.NTmax * W + (size_t)(PQ * PQ + W))) * NWR;
    if (lds > 160 * 1024) return -2;
    hipFuncSetAttribute((const void *)fit_quad_kernel<KP, 1, NWR, QM_RAGGED_LDS, PQ, true>,
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int64_t blocks = qp.n_cu;       // one workgroup per CU
    const int64_t need = (qa.f.N + NWR - 1) / NWR;
    if (blocks > need) blocks = need;
    hipLaunchKernelGGL((fit_quad_kernel<KP, 1, NWR, QM_RAGGED_LDS, PQ, true>), dim3((unsigned)blocks), dim3(NWR * 64), lds, st, qa, 0, 0);
    return (int)hipGetLastError();
}

// ragged panel, Z^T Z of the running series in the wave's registers (QM_RAGGED_REG): 8 waves per
// workgroup (two per SIMD, 256 registers each), history ring and residual staging in LDS; -2 when the
// staging does not fit (long series)
template <int KP, int PQ>
static int launch_quad_ragged_reg(const QuadPlan &qp, const QuadArgs &qa_, hipStream_t st)
{
    constexpr int NWR = 8;
    if (qp.P4 != PQ) return -2;
    QuadArgs qa = qa_;
    // the base-pair build keeps three columns' trend tables: two GramX in the staging rows, else the two-column build
    if (sizeof(double) * (size_t)qa.f.NTmax * W < 2 * sizeof(GramX) || !qa.f.Bw || qa.Mpre) qa.gram_harm = 0;
    const size_t lds = (quad_lanec_bytes<1>() + sizeof(QuadLds<KP, 1>) + quad_hist_bytes<1>(true) +
                        sizeof(double) * (size_t)qa.f.NTmax * W) * NWR;
    if (lds > 160 * 1024) return -2;
    if (sizeof(double) * (size_t)qa.f.NTmax * W < sizeof(GramX)) return -2;   // the Gram build borrows the staging rows
    if (int e = prebuild_grams<KP, 1>(qp, qa, st)) return e;
    hipFuncSetAttribute((const void *)fit_quad_kernel<KP, 1, NWR, QM_RAGGED_REG, PQ, true, true>,
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int64_t blocks = qp.n_cu;       // one workgroup per CU
    const int64_t need = (qa.f.N + NWR - 1) / NWR;
    if (blocks > need) blocks = need;
#ifdef TSF_QUAD_TIMING      // dev build: per-phase cycle counts (s_memtime), mean per series
    {
        QuadArgs qb = qa;
        const size_t nb = sizeof(long long) * 8 * (size_t)qa.f.N;
        hipMalloc((void **)&qb.dbg, nb);
        hipMemsetAsync(qb.dbg, 0, nb, st);
        hipLaunchKernelGGL((fit_quad_kernel<KP, 1, NWR, QM_RAGGED_REG, PQ, true, true>), dim3((unsigned)blocks), dim3(NWR * 64), lds, st, qb, 0, 0);
        hipStreamSynchronize(st);
        std::vector<long long> h(8 * (size_t)qa.f.N);
        hipMemcpy(h.data(), qb.dbg, nb, hipMemcpyDeviceToHost);
        double sum[8] = {0};
        for (int64_t i = 0; i < qa.f.N; ++i) for (int k = 0; k < 8; ++k) sum[k] += (double)h[i * 8 + k];
        fprintf(stderr, "[quad-timing ragged-reg] N %lld mean cycles/series: misc %.0f post %.0f ls %.0f resid %.0f eval %.0f gram-build %.0f | total %.0f\n",
                (long long)qa.f.N, sum[0] / qa.f.N, sum[1] / qa.f.N, sum[2] / qa.f.N, sum[3] / qa.f.N, sum[4] / qa.f.N, sum[5] / qa.f.N, sum[7] / qa.f.N);
        hipFree(qb.dbg);
        return (int)hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((fit_quad_kernel<KP, 1, NWR, QM_RAGGED_REG, PQ, true, true>), dim3((unsigned)blocks), dim3(NWR * 64), lds, st, qa, 0, 0);
    return (int)hipGetLastError();
}

// aligned panels share one M (in LDS when it fits: the one-slot kernels); ragged panels build one
// per series
template <int KP, int PPL, int PQ>
static int launch_quad_one(const QuadPlan &qp, const QuadArgs &qa, double *Mg, hipStream_t st)
{
    if (!qa.f.aligned) {
        if constexpr (PPL == 1 && PQ > 0) {
            int rc = launch_quad_ragged_reg<KP, PQ>(qp, qa, st);
            if (rc != -2) return rc;
            rc = launch_quad_ragged_lds<KP, PQ>(qp, qa, st);
            if (rc != -2) return rc;
        }
        if (int e = prebuild_grams<KP, PPL>(qp, qa, st)) return e;
        return launch_quad_mm<KP, PPL, QM_RAGGED, PQ>(qp, qa, Mg, st);
    }
    if constexpr (PPL == 1) {
        // Two kernels for aligned panels with P <= 64.  Z^T Z in LDS, 12 waves per CU (tsf_inst_quad3.hip): the
        // higher throughput.  Z^T Z in registers, 8 waves per CU (tsf_inst_quad4.hip): the faster lone wave (3.7
        // against 4.5 M cycles per cfg2 series) and no staging traffic.  A launch with few series per wave slot is
        // bound by its longest series, not by throughput -- measured (cfg2 model, ms per fit + forecast, LDS /
        // registers): 1 250 series 3.99 / 3.31, 2 500: 4.74 / 4.55, 5 000: 6.79 / 6.66, 10 000: 9.36 / 9.85 -- so up
        // to three series per slot of the register kernel it takes the call.  This is also what a rank of a
        // strong-scaled 10 000-series panel sees (1 250 series on each of 8 GPUs).  tsf_set_option(TSF_OPT_QUAD_REG, 0 / 1) forces.
        const int e = qp.opt ? qp.opt[TSF_OPT_QUAD_REG] : -1;
        const bool use_reg = e >= 0 ? e != 0 : qa.f.N <= (int64_t)3 * 8 * qp.n_cu;
        if (use_reg) {
            const int rc = launch_quad_aligned_reg(KP, qp, qa, Mg, st);      // tsf_inst_quad4.hip (-2: no such variant)
            if (rc != -2) return rc;
        }
        return launch_quad_aligned1(KP, qp, qa, Mg, st);   // tsf_inst_quad3.hip
    }
    else {
        // Two parameters per lane (64 < P <= 128).  Z^T Z is [P4][2][64] doubles: up to P4 = 88 it fits in LDS beside
        // the state of the kernel's four waves (cfg2 + 30 holiday columns, P4 = 84: 86 KB + 4 x 17 KB), and an
        // evaluation then reads its 86 KB from LDS instead of L2.  TSF_OPT_QUAD_M2_LDS 0: from L2 as before.
        const int e = qp.opt ? qp.opt[TSF_OPT_QUAD_M2_LDS] : -1;
        constexpr int NW = QuadShape<PPL, QM_LDS>::NW;
        const size_t lds = sizeof(double) * (size_t)qp.P4 * PPL * W + quad_lanec_bytes<PPL>() +
                           (sizeof(QuadLds<KP, PPL>) + quad_hist_bytes<PPL>(true)) * NW;
        if (lds <= 160 * 1024 && e != 0) return launch_quad_mm<KP, PPL, QM_LDS, PQ>(qp, qa, Mg, st);
        return launch_quad_mm<KP, PPL, QM_GLOBAL, PQ>(qp, qa, Mg, st);
    }
}

int launch_quad(int KP, const QuadPlan &qp, const QuadArgs &qa, double *Mg, hipStream_t st)
{
    // qp.P4 is the M row count the plan chose: 40 / 56 / 64 (compile-time, P <= 64) or any
    // multiple of 4 for the two-slot kernel
    switch (KP * 100 + (KP == 64 ? 0 : qp.P4)) {
    case 840: return launch_quad_one<8, 1, 40>(qp, qa, Mg, st);
    case 856: return launch_quad_one<8, 1, 56>(qp, qa, Mg, st);
    case 864: return launch_quad_one<8, 1, 64>(qp, qa, Mg, st);
    case 1640: return launch_quad_one<16, 1, 40>(qp, qa, Mg, st);
    case 1656: return launch_quad_one<16, 1, 56>(qp, qa, Mg, st);
    case 1664: return launch_quad_one<16, 1, 64>(qp, qa, Mg, st);
    case 2840: return launch_quad_one<28, 1, 40>(qp, qa, Mg, st);
    case 2856: return launch_quad_one<28, 1, 56>(qp, qa, Mg, st);
    case 2864: return launch_quad_one<28, 1, 64>(qp, qa, Mg, st);
    case 6400: return launch_quad_one<64, 2, 0>(qp, qa, Mg, st);
    default: return -1;
    }
}

// Newton with quadratic-form evaluations (tsf_newton_quad.h): Z^T Z once per grid, then one wave
// per series.  Aligned panels, one parameter per lane.
template <int KP, bool RAGGED>
static int launch_newton_quad_rg(const QuadPlan &qp, const QuadArgs &qa, double *Mg, int PM, int n_cu, hipStream_t st)
{
    if (!RAGGED) {
        hipLaunchKernelGGL((gram_build_kernel<KP, 1>), dim3((unsigned)qp.P4), dim3(64), 0, st, qa, Mg);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    const size_t lds = newton_quad_lds_bytes<KP>(PM, qa.f.NTmax);
    if (lds > 160 * 1024) return -1;
    hipFuncSetAttribute((const void *)newton_quad_kernel<KP, RAGGED>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int per_cu = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, newton_quad_kernel<KP, RAGGED>, 64, lds) != hipSuccess || per_cu < 1)
        per_cu = 1;
    int64_t blocks = (int64_t)per_cu * n_cu;
    if (blocks > qp.slots) blocks = qp.slots;         // one Z^T Z slot per resident block (ragged)
    if (blocks > qa.f.N) blocks = qa.f.N;
#ifdef TSF_QUAD_TIMING      // dev build: per-phase cycle counts (s_memtime), mean per series
    {
        QuadArgs qb = qa;
        const size_t nb = sizeof(long long) * 8 * (size_t)qa.f.N;
        hipMalloc((void **)&qb.dbg, nb);
        hipMemsetAsync(qb.dbg, 0, nb, st);
        hipLaunchKernelGGL((newton_quad_kernel<KP, RAGGED>), dim3((unsigned)blocks), dim3(64), lds, st, qb, PM);
        hipStreamSynchronize(st);
        std::vector<long long> h(8 * (size_t)qa.f.N);
        hipMemcpy(h.data(), qb.dbg, nb, hipMemcpyDeviceToHost);
        double sum[8] = {0};
        for (int64_t i = 0; i < qa.f.N; ++i) for (int k = 0; k < 8; ++k) sum[k] += (double)h[i * 8 + k];
        fprintf(stderr, "[newton-timing] N %lld waves/CU %d mean cycles/series: resid %.0f fd %.0f symm %.0f eig %.0f proj %.0f halving %.0f rest %.0f | total %.0f\n",
                (long long)qa.f.N, per_cu, sum[0] / qa.f.N, sum[1] / qa.f.N, sum[2] / qa.f.N, sum[3] / qa.f.N, sum[4] / qa.f.N, sum[5] / qa.f.N, sum[6] / qa.f.N, sum[7] / qa.f.N);
        hipFree(qb.dbg);
        return (int)hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((newton_quad_kernel<KP, RAGGED>), dim3((unsigned)blocks), dim3(64), lds, st, qa, PM);
    return (int)hipGetLastError();
}

// Aligned panels with enough series: several series per wave, the QL chains of a wave's slots run side by
// side with lane = slot (tsf_newton_batch.h).  Shape of such a launch: waves, slots per wave, list capacity,
// bytes of slot records (the caller provides them: QuadArgs::nb_buf).  false: this call does not take that kernel.
template <int KP>
static bool newton_batch_shape(int PM, int64_t N, int NTmax, int n_cu, NewtonBatchArgs &nb, int64_t &blocks, size_t &lds,
                               int &per_cu, size_t &rec_bytes, size_t &idx_bytes, const int *opt)
{
    // TSF_OPT_NEWTON_BATCH: 0 never, 2 whenever there are two series per wave (tests); default: from twelve series per
    // resident wave on -- measured on MI355X (profiles/r03_newton): 20 000 x 90 0.72 s here against 0.64 s on the
    // one-series-per-wave kernel (few rounds, and the longest series advances one iteration per round of NS
    // slots), 100 000: 3.16 against 3.69 s, 1 000 000: 15.1 against 26.0 s
    auto optv = [opt](int k) { return opt ? opt[k] : -1; };
    const int mode = optv(TSF_OPT_NEWTON_BATCH) >= 0 ? optv(TSF_OPT_NEWTON_BATCH) : 1;
    if (mode == 0) return false;
    lds = newton_batch_lds_bytes<KP>(PM, NTmax);
    if (lds > 160 * 1024) return false;
    hipFuncSetAttribute((const void *)newton_batch_kernel<KP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    per_cu = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, newton_batch_kernel<KP>, 64, lds) != hipSuccess || per_cu < 1)
        per_cu = 1;
    blocks = (int64_t)per_cu * n_cu;
    if (N < (mode == 2 ? 2 : 12) * blocks) return false;
    nb.NS = (int)((N + blocks - 1) / blocks);
    if (nb.NS > NB_MAX_SLOTS) nb.NS = NB_MAX_SLOTS;
    nb.flags = optv(TSF_OPT_NEWTON_FLAGS) >= 0 ? optv(TSF_OPT_NEWTON_FLAGS) : 0;
    { const int v = optv(TSF_OPT_NEWTON_NS); if (v >= 1 && v <= NB_MAX_SLOTS) nb.NS = v; }   // dev
    const int P = PM & ~1;      // PM = P | 1
    nb.LCAP = nb_lcap(P > 0 ? P : 1);
    {       // tests: a list too short for any decomposition -> the in-wave chain
        const int v = optv(TSF_OPT_NEWTON_LCAP);
        if (v >= 2 && v < nb.LCAP) nb.LCAP = v & ~1;
    }
    nb.rec_stride = (nb_rec_doubles(PM, nb.LCAP) + 1) & ~1LL;
    const size_t per_slot = sizeof(double) * (size_t)nb.rec_stride + sizeof(int) * (size_t)nb.LCAP;
    while (nb.NS > 2 && per_slot * (size_t)blocks * nb.NS > ((size_t)6 << 30)) --nb.NS;      // at most 6 GB of records
    rec_bytes = sizeof(double) * (size_t)nb.rec_stride * blocks * nb.NS;
    idx_bytes = sizeof(int) * (size_t)nb.LCAP * blocks * nb.NS;
    return true;
}

size_t newton_batch_scratch_bytes(int KP, int PM, int64_t N, int NTmax, int n_cu, const int *opt)
{
    NewtonBatchArgs nb;
    int64_t blocks = 0;
    size_t lds = 0, rb = 0, ib = 0;
    int per_cu = 0;
    bool ok = false;
    switch (KP) {
    case 8: ok = newton_batch_shape<8>(PM, N, NTmax, n_cu, nb, blocks, lds, per_cu, rb, ib, opt); break;
    case 16: ok = newton_batch_shape<16>(PM, N, NTmax, n_cu, nb, blocks, lds, per_cu, rb, ib, opt); break;
    case 28: ok = newton_batch_shape<28>(PM, N, NTmax, n_cu, nb, blocks, lds, per_cu, rb, ib, opt); break;
    default: break;
    }
    return ok ? rb + ib : 0;
}

// -2: not used (the one-series-per-wave kernel takes the call)
template <int KP>
static int launch_newton_batch(const QuadPlan &qp, const QuadArgs &qa, double *Mg, int PM, int n_cu, hipStream_t st)
{
    NewtonBatchArgs nb;
    int64_t blocks = 0;
    size_t lds = 0, rec_bytes = 0, idx_bytes = 0;
    int per_cu = 0;
    if (!newton_batch_shape<KP>(PM, qa.f.N, qa.f.NTmax, n_cu, nb, blocks, lds, per_cu, rec_bytes, idx_bytes, qp.opt)) return -2;
    // The records come from the caller's cached workspace, NOT from hipMallocAsync: with stream-ordered
    // scratch of this size (gigabytes, above the pool's release threshold) fits went wrong intermittently
    // after a large call on this ROCm (tools/dev/nb_debug.py; plain hipMalloc: never)
    // TSF_OPT_DEBUG_ASYNC_SCRATCH (dev): round 3's stream-ordered scratch again, with the instruments of the round-4
    // review around it -- canary pages, a synchronisation before the free, a pool that never releases
    const int dbg_async = (qp.opt && qp.opt[TSF_OPT_DEBUG_ASYNC_SCRATCH] > 0) ? qp.opt[TSF_OPT_DEBUG_ASYNC_SCRATCH] : 0;
    constexpr size_t CANARY = (size_t)1 << 20;
    void *async_base = nullptr;
    void *buf = qa.nb_buf;
    if (dbg_async & 1) {
        if (dbg_async & 8) {
            hipMemPool_t pool; int dev = 0; hipGetDevice(&dev);
            if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) { uint64_t thr = UINT64_MAX; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr); }
        }
        const size_t pad = (dbg_async & 4) ? CANARY : 0;
        if (hipMallocAsync(&async_base, rec_bytes + idx_bytes + 2 * pad, st) != hipSuccess) { (void)hipGetLastError(); return -2; }
        buf = (char *)async_base + pad;
        if (pad) { hipMemsetAsync(async_base, 0x5C, pad, st); hipMemsetAsync((char *)buf + rec_bytes + idx_bytes, 0x5C, pad, st); }
    } else if (!qa.nb_buf || qa.nb_bytes < rec_bytes + idx_bytes) {
        return -2;
    }
    nb.rec = (double *)buf;
    nb.rot_idx = (int *)((char *)buf + rec_bytes);
    if (qp.opt && qp.opt[TSF_OPT_NEWTON_FILL] >= 0) hipMemsetAsync(buf, qp.opt[TSF_OPT_NEWTON_FILL], rec_bytes + idx_bytes, st);   // dev: 255 = NaN everywhere
    hipLaunchKernelGGL((gram_build_kernel<KP, 1>), dim3((unsigned)qp.P4), dim3(64), 0, st, qa, Mg);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
#ifdef TSF_QUAD_TIMING
    {
        QuadArgs qb = qa;
        const size_t nbytes = sizeof(long long) * 9 * (size_t)blocks;
        hipMalloc((void **)&qb.dbg, nbytes);
        hipMemsetAsync(qb.dbg, 0, nbytes, st);
        hipLaunchKernelGGL((newton_batch_kernel<KP>), dim3((unsigned)blocks), dim3(64), lds, st, qb, PM, nb);
        hipStreamSynchronize(st);
        std::vector<long long> h(9 * (size_t)blocks);
        hipMemcpy(h.data(), qb.dbg, nbytes, hipMemcpyDeviceToHost);
        double sum[9] = {0};
        for (int64_t i = 0; i < blocks; ++i) for (int k = 0; k < 9; ++k) sum[k] += (double)h[i * 9 + k];
        const double N = (double)qa.f.N;
        fprintf(stderr, "[newton-batch-timing] N %lld waves/CU %d slots %d mean cycles/series: symm+tridiag+Q %.0f chain %.0f apply %.0f proj %.0f halving %.0f resid %.0f fd %.0f store %.0f | total %.0f\n",
                (long long)qa.f.N, per_cu, nb.NS, sum[0] / N, sum[1] / N, sum[2] / N, sum[3] / N, sum[4] / N, sum[5] / N, sum[6] / N, sum[7] / N, sum[8] / N);
        hipFree(qb.dbg);
        return (int)hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((newton_batch_kernel<KP>), dim3((unsigned)blocks), dim3(64), lds, st, qa, PM, nb);
    const int lrc = (int)hipGetLastError();
    if (async_base) {
        if (dbg_async & 4) {
            std::vector<unsigned char> h(2 * CANARY);
            hipMemcpyAsync(h.data(), async_base, CANARY, hipMemcpyDeviceToHost, st);
            hipMemcpyAsync(h.data() + CANARY, (char *)buf + rec_bytes + idx_bytes, CANARY, hipMemcpyDeviceToHost, st);
            hipStreamSynchronize(st);
            size_t bad = 0;
            for (unsigned char c : h) bad += c != 0x5C;
            fprintf(stderr, "[async-scratch] %zu bytes of records, canary bytes overwritten: %zu\n", rec_bytes + idx_bytes, bad);
        }
        if (dbg_async & 2) hipStreamSynchronize(st);
        hipFreeAsync(async_base, st);
    }
    return lrc;
}

template <int KP>
static int launch_newton_quad_one(const QuadPlan &qp, const QuadArgs &qa, double *Mg, int PM, int n_cu, hipStream_t st)
{
    if (qa.f.aligned) {
        const int rc = launch_newton_batch<KP>(qp, qa, Mg, PM, n_cu, st);
        if (rc != -2) return rc;
    }
    if (qa.f.aligned) return launch_newton_quad_rg<KP, false>(qp, qa, Mg, PM, n_cu, st);
    return launch_newton_quad_rg<KP, true>(qp, qa, Mg, PM, n_cu, st);
}

int launch_newton_quad(int KP, const QuadPlan &qp, const QuadArgs &qa, double *Mg, int PM, int n_cu, hipStream_t st)
{
    switch (KP) {
    case 8: return launch_newton_quad_one<8>(qp, qa, Mg, PM, n_cu, st);
    case 16: return launch_newton_quad_one<16>(qp, qa, Mg, PM, n_cu, st);
    case 28: return launch_newton_quad_one<28>(qp, qa, Mg, PM, n_cu, st);
    default: return -1;
    }
}

}  // namespace tsf
