// tsf_inst_quad.hip -- instantiates the quadratic-form fit path (tsf_quad_kernels.h).
#include "tsf_quad_kernels.h"
#include "tsf_launch.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>

#ifndef TSF_QUAD_NW
#define TSF_QUAD_NW 8
#endif

namespace tsf {

int quad_waves_per_block() { return TSF_QUAD_NW; }

template <int KP, int PPL, bool MLDS>
static int launch_quad_one(const QuadPlan &qp, const QuadArgs &qa, double *Mg, hipStream_t st)
{
    constexpr int NW = TSF_QUAD_NW;
    const char *dbg = getenv("TSF_QUAD_DEBUG");
    hipLaunchKernelGGL((gram_build_kernel<KP, PPL>), dim3((unsigned)qp.P4), dim3(64), 0, st, qa, Mg);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (dbg) {
        e = hipStreamSynchronize(st);
        fprintf(stderr, "[quad] gram_build done: %s (P4 %d blocks %d slots %d)\n", hipGetErrorString(e), qp.P4, qp.blocks, qp.slots);
        if (dbg[0] == '1') return (int)e;
    }
    const size_t lds = (MLDS ? sizeof(double) * (size_t)qp.P4 * PPL * W : 0) + sizeof(QuadLds<KP, PPL>) * NW;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void *)fit_quad_kernel<KP, PPL, NW, MLDS>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    QuadArgs qb = qa;
    long long *hdbg = nullptr;
    if (dbg && dbg[0] == '3') {
        hipHostMalloc((void **)&hdbg, sizeof(long long) * 8 * qp.blocks * NW, hipHostMallocCoherent);
        memset(hdbg, 0, sizeof(long long) * 8 * qp.blocks * NW);
        qb.dbg = hdbg;
    }
    hipLaunchKernelGGL((fit_quad_kernel<KP, PPL, NW, MLDS>), dim3((unsigned)qp.blocks), dim3(NW * 64), lds, st, qb);
    if (hdbg) {
        e = hipGetLastError();
        fprintf(stderr, "[quad] fit launched: %s lds %zu\n", hipGetErrorString(e), lds);
        for (int it = 0; it < 50; ++it) {
            if (hipStreamQuery(st) == hipSuccess) { fprintf(stderr, "[quad] finished after %d polls\n", it); break; }
            usleep(100000);
        }
        for (int w = 0; w < qp.blocks * NW && w < 32; ++w)
            fprintf(stderr, "[quad] wave %d: m0 %lld m1 %lld m2 %lld m3 %lld m7 %lld\n", w, hdbg[w * 8], hdbg[w * 8 + 1], hdbg[w * 8 + 2], hdbg[w * 8 + 3], hdbg[w * 8 + 7]);
        fflush(stderr);
        if (hipStreamQuery(st) != hipSuccess) _exit(3);
        return 0;
    }
    if (dbg) {
        e = hipGetLastError();
        fprintf(stderr, "[quad] fit launched: %s lds %zu\n", hipGetErrorString(e), lds);
        e = hipStreamSynchronize(st);
        fprintf(stderr, "[quad] fit done: %s\n", hipGetErrorString(e));
        return (int)e;
    }
    return (int)hipGetLastError();
}

int launch_quad(int KP, const QuadPlan &qp, const QuadArgs &qa, double *Mg, hipStream_t st)
{
    switch (KP) {
    case 8: return launch_quad_one<8, 1, true>(qp, qa, Mg, st);
    case 16: return launch_quad_one<16, 1, true>(qp, qa, Mg, st);
    case 28: return launch_quad_one<28, 1, true>(qp, qa, Mg, st);
    case 64: return launch_quad_one<64, 2, false>(qp, qa, Mg, st);
    default: return -1;
    }
}

}  // namespace tsf
