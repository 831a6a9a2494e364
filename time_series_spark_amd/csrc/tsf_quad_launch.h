// tsf_quad_launch.h -- launch templates of fit_quad_kernel shared by tsf_inst_quad.hip (ragged
// panels, two-slot kernel) and tsf_inst_quad3.hip (aligned panels, P <= 64: the kernel compiled for
// three waves per SIMD).
#pragma once
#include "tsf_quad_kernels.h"
#include "tsf_launch.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifndef TSF_QUAD_NW
#define TSF_QUAD_NW 8       // two waves per SIMD
#endif
#ifndef TSF_QUAD_NW3
#define TSF_QUAD_NW3 12     // three waves per SIMD (history in LDS: fit_one_quad HLDS)
#endif
#ifndef TSF_QUAD_NTR
#define TSF_QUAD_NTR 12     // steps per lane up to which a residual pass keeps its weights in registers (T <= 768)
#endif
#ifndef TSF_QUAD_NW4
#define TSF_QUAD_NW4 16     // four waves per SIMD (trend tables pooled: fit_quad_kernel RPOOL)
#endif
#ifndef TSF_QUAD_POOL_MIN
#define TSF_QUAD_POOL_MIN 3 // fewest shared table copies the 16-wave kernel is launched with
#endif
#ifndef TSF_QUAD_POOL_RB_MIN
#define TSF_QUAD_POOL_RB_MIN 4      // ... and the fewest with which a slot also carries the staging rows of r
#endif
#ifndef TSF_QUAD_W4_MIN_PER_SLOT
#define TSF_QUAD_W4_MIN_PER_SLOT 8  // series per wave slot from which the 16-wave kernel takes an aligned panel
#endif
#ifndef TSF_QUAD_W4_DEFAULT
#define TSF_QUAD_W4_DEFAULT -1      // -1: the 16-wave kernel wherever it fits; 0: never
#endif
#ifndef TSF_QUAD_NW2G
#define TSF_QUAD_NW2G 8     // two-slot kernel, Z^T Z read from L2
#endif
#ifndef TSF_QUAD_NW2
#define TSF_QUAD_NW2 4      // two-slot kernel (P > 64): twice the per-wave LDS
#endif

namespace tsf {

// waves per workgroup of the kernel variant: the shared-M one-slot kernel runs 12 waves per CU (three
// per SIMD); the others (two per SIMD) 8, the two-slot kernel 4
template <int PPL, int MMODE>
struct QuadShape {
    static constexpr bool HL = true;        // L-BFGS history in LDS (these variants have the room)
    // (two-slot kernel with Z^T Z in L2: no matrix in LDS, room for eight waves -- two per SIMD at 215 registers)
    static constexpr int NW = (PPL == 2) ? (MMODE == QM_GLOBAL ? TSF_QUAD_NW2G : TSF_QUAD_NW2) : (quad_three_waves(MMODE, PPL, HL) ? TSF_QUAD_NW3 : TSF_QUAD_NW);
};

// RPOOL: the 16-waves-per-CU kernel (four per SIMD) with pool_slots shared copies of the trend tables
// LDS of the RPOOL kernel besides its pool slots: M, lane constants, QuadWave + history ring per wave, locks
template <int PPL>
static size_t quad_pool_base_bytes(int P4, int nw)
{
    return sizeof(double) * (size_t)P4 * PPL * W + quad_lanec_bytes<PPL>() +
           (sizeof(QuadWave<PPL>) + quad_hist_bytes<PPL>(true)) * (size_t)nw + QUAD_POOL_LOCK_BYTES;
}

template <int KP, int PPL, int MMODE, int PQ, bool RLDS, int NTR = 0, bool RPOOL = false>
static int launch_quad_rl(const QuadPlan &qp, const QuadArgs &qa_in, double *Mg, hipStream_t st, int pool_slots = 0, int pool_slot_bytes = 0)
{
    constexpr int NW = RPOOL ? TSF_QUAD_NW4 : QuadShape<PPL, MMODE>::NW;
    constexpr bool MLDS = MMODE == QM_LDS;
    constexpr bool HL = QuadShape<PPL, MMODE>::HL;
    const QuadArgs &qa = qa_in;
    int64_t blocks = qp.n_cu;                       // persistent: LDS admits one workgroup per CU
    if (blocks > (qa.f.N + NW - 1) / NW) blocks = (qa.f.N + NW - 1) / NW;
    if (blocks < 1) blocks = 1;
    if (MMODE != QM_RAGGED && MMODE != QM_RAGGED_REG) {       // aligned panel: one M for the whole call
        hipLaunchKernelGGL((gram_build_kernel<KP, PPL>), dim3((unsigned)qp.P4), dim3(64), 0, st, qa, Mg);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    const size_t lds = RPOOL ? quad_pool_base_bytes<PPL>(qp.P4, NW) + (size_t)pool_slot_bytes * (size_t)pool_slots
                     : (MLDS ? sizeof(double) * (size_t)qp.P4 * PPL * W : 0) + quad_lanec_bytes<PPL>() * ((MMODE == QM_RAGGED || MMODE == QM_RAGGED_REG) ? NW : 1) +
                       (sizeof(QuadLds<KP, PPL>) + quad_hist_bytes<PPL>(HL)) * NW +
                       (RLDS ? sizeof(double) * (size_t)NW * qa.f.NTmax * W : 0);
    // per launch: the attribute is per device, and a process may drive several GPUs
    hipFuncSetAttribute((const void *)fit_quad_kernel<KP, PPL, NW, MMODE, PQ, RLDS, HL, NTR, RPOOL>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#ifdef TSF_QUAD_TIMING      // dev build: per-phase cycle counts (s_memtime) summed per series
    {
        QuadArgs qb = qa;
        const size_t nb = sizeof(long long) * 8 * (size_t)qa.f.N;
        hipMalloc((void **)&qb.dbg, nb);
        hipMemsetAsync(qb.dbg, 0, nb, st);
        hipLaunchKernelGGL((fit_quad_kernel<KP, PPL, NW, MMODE, PQ, RLDS, HL, NTR, RPOOL>), dim3((unsigned)blocks), dim3(NW * 64), lds, st, qb, pool_slots, pool_slot_bytes);
        hipStreamSynchronize(st);
        std::vector<long long> h(8 * (size_t)qa.f.N);
        hipMemcpy(h.data(), qb.dbg, nb, hipMemcpyDeviceToHost);
        double sum[8] = {0};
        long long mx = 0;
        for (int64_t i = 0; i < qa.f.N; ++i) { for (int k = 0; k < 8; ++k) sum[k] += (double)h[i * 8 + k]; if (h[i * 8 + 7] > mx) mx = h[i * 8 + 7]; }
        fprintf(stderr, "[quad-timing] N %lld waves/CU %d pool %d mean cycles/series: misc %.0f post %.0f (+ two-loop %.0f) ls %.0f resid %.0f gram %.0f newDFp %.0f | total %.0f max %lld\n",
                (long long)qa.f.N, NW, pool_slots, sum[0] / qa.f.N, sum[1] / qa.f.N, sum[6] / qa.f.N, sum[2] / qa.f.N, sum[3] / qa.f.N, sum[4] / qa.f.N, sum[5] / qa.f.N, sum[7] / qa.f.N, mx);
        hipFree(qb.dbg);
        return (int)hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((fit_quad_kernel<KP, PPL, NW, MMODE, PQ, RLDS, HL, NTR, RPOOL>), dim3((unsigned)blocks), dim3(NW * 64), lds, st, qa, pool_slots, pool_slot_bytes);
    return (int)hipGetLastError();
}

// residual staging in LDS when M + per-wave state + NW x NTmax x 64 doubles fit in 160 KB
template <int KP, int PPL, int MMODE, int PQ>
static int launch_quad_mm(const QuadPlan &qp, const QuadArgs &qa, double *Mg, hipStream_t st)
{
    constexpr int NW = QuadShape<PPL, MMODE>::NW;
    constexpr bool HL = QuadShape<PPL, MMODE>::HL;
    if constexpr (MMODE == QM_LDS && PPL == 1 && PQ > 0) {
        // Sixteen waves per CU (four per SIMD, <= 128 registers) for LARGE panels, when at least
        // TSF_QUAD_POOL_MIN shared copies of the trend tables fit next to M, the history rings and the per-wave
        // vectors (tsf_quad_kernels.h QuadPool).  Measured (profiles/r03_w4): a fourth wave per SIMD makes every
        // wave 23 % slower (5.85 against 4.77 M cycles per cfg2 series) for a third more waves: + 8.5 % on 160 000 x 730
        // (104.5 -> 96.3 ms), + 11 % on 1 000 000 x 90 (600 -> 541 ms), + 2 % on 40 000 series -- and - 7 % on
        // 10 000 (9.7 -> 10.4 ms), where the launch is its longest fits and not throughput.  Hence from
        // TSF_QUAD_W4_MIN_PER_SLOT series per wave slot on.  The weights of a residual pass ride in the pool slot
        // when they are short (cfg5), else they go through the global scratch.
        // TSF_OPT_QUAD_W4 0: never; n > 0: always, with at most n copies (tests, measurements).
        const int e = qp.opt ? qp.opt[TSF_OPT_QUAD_W4] : -1;
        const int forced = e >= 0 ? e : TSF_QUAD_W4_DEFAULT;
        const size_t pbase = quad_pool_base_bytes<PPL>(qp.P4, TSF_QUAD_NW4);
        const size_t avail = pbase < 160 * 1024 ? 160 * 1024 - pbase : 0;
        // short series: the staging rows ride in the slot when that still leaves TSF_QUAD_POOL_RB_MIN copies
        size_t slot = sizeof(QuadLds<KP, PPL>) + sizeof(double) * (size_t)qa.f.NTmax * W;
        if (avail / slot < TSF_QUAD_POOL_RB_MIN) slot = sizeof(QuadLds<KP, PPL>);
        int ns = (int)(avail / slot);
        if (ns > TSF_QUAD_NW4) ns = TSF_QUAD_NW4;
        if (forced > 0 && forced < ns) ns = forced;
        if (forced != 0 && ns >= (forced > 0 ? 1 : TSF_QUAD_POOL_MIN) &&
            (forced > 0 || qa.f.N >= (int64_t)TSF_QUAD_W4_MIN_PER_SLOT * TSF_QUAD_NW4 * qp.n_cu))
            return launch_quad_rl<KP, PPL, MMODE, PQ, false, 0, true>(qp, qa, Mg, st, ns, (int)slot);
    }
    const size_t base = (MMODE == QM_LDS ? sizeof(double) * (size_t)qp.P4 * PPL * W : 0) + quad_lanec_bytes<PPL>() * ((MMODE == QM_RAGGED || MMODE == QM_RAGGED_REG) ? NW : 1) +
                        (sizeof(QuadLds<KP, PPL>) + quad_hist_bytes<PPL>(HL)) * NW;
    const size_t rbytes = sizeof(double) * (size_t)NW * qa.f.NTmax * W;
    if (base + rbytes <= 160 * 1024) return launch_quad_rl<KP, PPL, MMODE, PQ, true>(qp, qa, Mg, st);
    // no room in LDS (the 12-waves-per-CU kernel on series of >= ~600 rows): the weights of the first
    // TSF_QUAD_NTR steps of a pass stay in registers (cfg2: all 12), those of later steps go through the
    // global scratch (cfg3, 1 095 rows: 6 of 18).  TSF_OPT_QUAD_RREG 0: all of them (round 2's route; tests).
    if constexpr (MMODE == QM_LDS && PPL == 1 && PQ > 0) {
        if (!(qp.opt && qp.opt[TSF_OPT_QUAD_RREG] == 0))
            return launch_quad_rl<KP, PPL, MMODE, PQ, false, TSF_QUAD_NTR>(qp, qa, Mg, st);
    }
    return launch_quad_rl<KP, PPL, MMODE, PQ, false>(qp, qa, Mg, st);
}

}  // namespace tsf
