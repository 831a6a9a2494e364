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
#ifndef TSF_QUAD_NW2
#define TSF_QUAD_NW2 4      // two-slot kernel (P > 64): twice the per-wave LDS
#endif

namespace tsf {

// waves per workgroup of the kernel variant: the shared-M one-slot kernel runs 12 waves per CU (three
// per SIMD); the others (two per SIMD) 8, the two-slot kernel 4
template <int PPL, int MMODE>
struct QuadShape {
    static constexpr bool HL = true;        // L-BFGS history in LDS (these variants have the room)
    static constexpr int NW = (PPL == 2) ? TSF_QUAD_NW2 : (quad_three_waves(MMODE, PPL, HL) ? TSF_QUAD_NW3 : TSF_QUAD_NW);
};

template <int KP, int PPL, int MMODE, int PQ, bool RLDS>
static int launch_quad_rl(const QuadPlan &qp, const QuadArgs &qa, double *Mg, hipStream_t st)
{
    constexpr int NW = QuadShape<PPL, MMODE>::NW;
    constexpr bool MLDS = MMODE == QM_LDS;
    constexpr bool HL = QuadShape<PPL, MMODE>::HL;
    int64_t blocks = qp.n_cu;                       // persistent: LDS admits one workgroup per CU
    if (blocks > (qa.f.N + NW - 1) / NW) blocks = (qa.f.N + NW - 1) / NW;
    if (blocks < 1) blocks = 1;
    if (MMODE != QM_RAGGED && MMODE != QM_RAGGED_REG) {       // aligned panel: one M for the whole call
        hipLaunchKernelGGL((gram_build_kernel<KP, PPL>), dim3((unsigned)qp.P4), dim3(64), 0, st, qa, Mg);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    const size_t lds = (MLDS ? sizeof(double) * (size_t)qp.P4 * PPL * W : 0) + quad_lanec_bytes<PPL>() * ((MMODE == QM_RAGGED || MMODE == QM_RAGGED_REG) ? NW : 1) +
                       (sizeof(QuadLds<KP, PPL>) + quad_hist_bytes<PPL>(HL)) * NW +
                       (RLDS ? sizeof(double) * (size_t)NW * qa.f.NTmax * W : 0);
    // per launch: the attribute is per device, and a process may drive several GPUs
    hipFuncSetAttribute((const void *)fit_quad_kernel<KP, PPL, NW, MMODE, PQ, RLDS, HL>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#ifdef TSF_QUAD_TIMING      // dev build: per-phase cycle counts (s_memtime) summed per series
    {
        QuadArgs qb = qa;
        const size_t nb = sizeof(long long) * 8 * (size_t)qa.f.N;
        hipMalloc((void **)&qb.dbg, nb);
        hipMemsetAsync(qb.dbg, 0, nb, st);
        hipLaunchKernelGGL((fit_quad_kernel<KP, PPL, NW, MMODE, PQ, RLDS, HL>), dim3((unsigned)blocks), dim3(NW * 64), lds, st, qb);
        hipStreamSynchronize(st);
        std::vector<long long> h(8 * (size_t)qa.f.N);
        hipMemcpy(h.data(), qb.dbg, nb, hipMemcpyDeviceToHost);
        double sum[8] = {0};
        long long mx = 0;
        for (int64_t i = 0; i < qa.f.N; ++i) { for (int k = 0; k < 8; ++k) sum[k] += (double)h[i * 8 + k]; if (h[i * 8 + 7] > mx) mx = h[i * 8 + 7]; }
        fprintf(stderr, "[quad-timing] N %lld mean cycles/series: misc %.0f post %.0f ls %.0f resid %.0f gram %.0f | total %.0f max %lld\n",
                (long long)qa.f.N, sum[0] / qa.f.N, sum[1] / qa.f.N, sum[2] / qa.f.N, sum[3] / qa.f.N, sum[4] / qa.f.N, sum[7] / qa.f.N, mx);
        hipFree(qb.dbg);
        return (int)hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((fit_quad_kernel<KP, PPL, NW, MMODE, PQ, RLDS, HL>), dim3((unsigned)blocks), dim3(NW * 64), lds, st, qa);
    return (int)hipGetLastError();
}

// residual staging in LDS when M + per-wave state + NW x NTmax x 64 doubles fit in 160 KB
template <int KP, int PPL, int MMODE, int PQ>
static int launch_quad_mm(const QuadPlan &qp, const QuadArgs &qa, double *Mg, hipStream_t st)
{
    constexpr int NW = QuadShape<PPL, MMODE>::NW;
    constexpr bool HL = QuadShape<PPL, MMODE>::HL;
    const size_t base = (MMODE == QM_LDS ? sizeof(double) * (size_t)qp.P4 * PPL * W : 0) + quad_lanec_bytes<PPL>() * ((MMODE == QM_RAGGED || MMODE == QM_RAGGED_REG) ? NW : 1) +
                        (sizeof(QuadLds<KP, PPL>) + quad_hist_bytes<PPL>(HL)) * NW;
    const size_t rbytes = sizeof(double) * (size_t)NW * qa.f.NTmax * W;
    if (base + rbytes <= 160 * 1024) return launch_quad_rl<KP, PPL, MMODE, PQ, true>(qp, qa, Mg, st);
    return launch_quad_rl<KP, PPL, MMODE, PQ, false>(qp, qa, Mg, st);
}

}  // namespace tsf
