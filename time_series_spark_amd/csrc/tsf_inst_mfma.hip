// tsf_inst_mfma.hip -- instantiates the matrix-core residual-form fit path (tsf_mfma_kernels.h).
#include "tsf_mfma_kernels.h"
#include "tsf_launch.h"
#include <cstdio>
#include <vector>

namespace tsf {

size_t mfma_lds_bytes(int KP)
{
    return KP == 8 ? MtLayout<8>::total : (KP == 16 ? MtLayout<16>::total : MtLayout<28>::total);
}

int launch_mfma_layout(const FitArgs &a, int KP, const MfmaTabs &mt, double *XF, double *XB,
                       double *XT, double *tq, uint16_t *cq, int8_t *cpof, double *yq, int *overflow,
                       hipStream_t st)
{
    hipLaunchKernelGGL(mfma_layout_kernel, dim3(W), dim3(256), 0, st, a.gtab, a.tw, a.cw, a.Xw, KP, mt.NG,
                       mt.KF, mt.NCB, XF, XB, XT, tq, cq, cpof, overflow);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mfma_y_kernel, dim3((unsigned)a.N), dim3(256), 0, st, a.gtab, a.yw, a.NTmax, mt.NG, yq);
    return (int)hipGetLastError();
}

template <int KP, int G, int M>
static int launch_mfma_one(const FitArgs &a, const MfmaTabs &mt, int blocks, hipStream_t st)
{
    // per launch: the attribute is per device, and a process may drive several GPUs
    hipFuncSetAttribute((const void *)fit_mfma_kernel<KP, G, M>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
#ifdef TSF_MFMA_TIMING
    {
        MfmaTabs m2 = mt;
        const size_t nb = sizeof(long long) * 6 * (size_t)blocks * MT_NW;
        hipMalloc((void **)&m2.dbg, nb);
        hipMemsetAsync(m2.dbg, 0, nb, st);
        hipLaunchKernelGGL((fit_mfma_kernel<KP, G, M>), dim3((unsigned)blocks), dim3(MT_NW * W),
                           MtLayout<KP>::total, st, a, m2);
        hipStreamSynchronize(st);
        std::vector<long long> h(6 * (size_t)blocks * MT_NW);
        hipMemcpy(h.data(), m2.dbg, nb, hipMemcpyDeviceToHost);
        double sum[6] = {0, 0, 0, 0, 0, 0};
        for (size_t i = 0; i < (size_t)blocks * MT_NW; ++i) for (int k = 0; k < 6; ++k) sum[k] += (double)h[i * 6 + k];
        const double rounds = sum[5] > 0 ? sum[5] : 1;
        fprintf(stderr, "[mfma-timing] blocks %d: rounds/wave %.0f; cycles per round per wave: owner %.0f  wait1 %.0f  eval %.0f  wait2 %.0f  reduce+wait3 %.0f\n",
                blocks, rounds / (blocks * MT_NW), sum[0] / rounds, sum[1] / rounds, sum[2] / rounds, sum[3] / rounds, sum[4] / rounds);
        hipFree(m2.dbg);
        return (int)hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((fit_mfma_kernel<KP, G, M>), dim3((unsigned)blocks), dim3(MT_NW * W),
                       MtLayout<KP>::total, st, a, mt);
    return (int)hipGetLastError();
}

template <int G, int M>
static int launch_mfma_gm(int KP, const FitArgs &a, const MfmaTabs &mt, int blocks, hipStream_t st)
{
    switch (KP) {
    case 8: return launch_mfma_one<8, G, M>(a, mt, blocks, st);
    case 16: return launch_mfma_one<16, G, M>(a, mt, blocks, st);
    case 28: return launch_mfma_one<28, G, M>(a, mt, blocks, st);
    default: return -1;
    }
}

int launch_mfma(int KP, int growth, int mode, const FitArgs &a, const MfmaTabs &mt, int blocks,
                hipStream_t st)
{
    if (growth == 0 && mode == 0) return launch_mfma_gm<0, 0>(KP, a, mt, blocks, st);
    if (growth == 0 && mode == 1) return launch_mfma_gm<0, 1>(KP, a, mt, blocks, st);
    if (growth == 1 && mode == 0) return launch_mfma_gm<1, 0>(KP, a, mt, blocks, st);
    if (growth == 1 && mode == 1) return launch_mfma_gm<1, 1>(KP, a, mt, blocks, st);
    return -1;
}

}  // namespace tsf
