// tsf_inst_quad3.hip -- the quadratic-form fit kernel for aligned panels with P <= 64 (shared
// Z^T Z in LDS): the variant compiled for three waves per SIMD (L-BFGS history in LDS, <= 168
// VGPRs, 12 waves per CU).  Its own translation unit because it is built with
// -mllvm -disable-machine-licm (build.py): with machine LICM on, the compiler hoists the fp64
// literals of the exp / cubic-interpolation code into ~40 VGPRs for the whole kernel and then
// spills them to scratch, reloading them in the middle of the dependent chains.
#include "tsf_quad_launch.h"
#include "tsf_map_quad.h"

namespace tsf {

int launch_quad_aligned1(int KP, const QuadPlan &qp, const QuadArgs &qa, double *Mg, hipStream_t st)
{
    switch (KP * 100 + qp.P4) {
    case 840: return launch_quad_mm<8, 1, QM_LDS, 40>(qp, qa, Mg, st);
    case 856: return launch_quad_mm<8, 1, QM_LDS, 56>(qp, qa, Mg, st);
    case 864: return launch_quad_mm<8, 1, QM_LDS, 64>(qp, qa, Mg, st);
    case 1640: return launch_quad_mm<16, 1, QM_LDS, 40>(qp, qa, Mg, st);
    case 1656: return launch_quad_mm<16, 1, QM_LDS, 56>(qp, qa, Mg, st);
    case 1664: return launch_quad_mm<16, 1, QM_LDS, 64>(qp, qa, Mg, st);
    case 2840: return launch_quad_mm<28, 1, QM_LDS, 40>(qp, qa, Mg, st);
    case 2856: return launch_quad_mm<28, 1, QM_LDS, 56>(qp, qa, Mg, st);
    case 2864: return launch_quad_mm<28, 1, QM_LDS, 64>(qp, qa, Mg, st);
    default: return -1;
    }
}

// tsf_eval_quadratic: gram_build_kernel + eval_quad_kernel (one wave per workgroup, at most qp.slots of them:
// the staging rows of long series are per workgroup)
template <int KP, int PQ>
static int launch_eval_quad_one(const QuadPlan &qp, const QuadArgs &qa, double *Mg, const double *theta_ref, hipStream_t st)
{
    hipLaunchKernelGGL((gram_build_kernel<KP, 1>), dim3((unsigned)qp.P4), dim3(64), 0, st, qa, Mg);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const size_t lds = sizeof(double) * (size_t)PQ * W + quad_lanec_bytes<1>() + sizeof(QuadLds<KP, 1>);
    hipFuncSetAttribute((const void *)eval_quad_kernel<KP, PQ, TSF_QUAD_NTR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int64_t blocks = qa.f.N < qp.slots ? qa.f.N : qp.slots;
    hipLaunchKernelGGL((eval_quad_kernel<KP, PQ, TSF_QUAD_NTR>), dim3((unsigned)blocks), dim3(64), lds, st, qa, theta_ref);
    return (int)hipGetLastError();
}

// converge = MAP on an aligned linear / additive panel: gram_build_kernel + map_quad_kernel (tsf_map_quad.h), one wave per
// workgroup, at most qp.slots of them
template <int KP>
static int launch_map_quad_one(const QuadPlan &qp, const QuadArgs &qa, double *Mg, int PM, hipStream_t st)
{
    const size_t lds = map_quad_lds_bytes<KP>(PM);
    if (lds > 160 * 1024) return -1;
    int64_t blocks = qa.f.N < qp.slots ? qa.f.N : qp.slots;
    if (!qa.f.aligned) {
        // ragged: Z^T Z per distinct calendar ahead of the kernel where calendars are shared (as tsf_inst_quad.hip does for
        // the fit kernels: one rbuf slot per (grid, column)), else per series inside it
        if (qa.Mpre) {
            const int64_t per = qp.slots / qp.P4 > 0 ? qp.slots / qp.P4 : 1;
            for (int64_t g0 = 0; g0 < qa.n_pre; g0 += per) {
                const int64_t cnt = qa.n_pre - g0 < per ? qa.n_pre - g0 : per;
                hipLaunchKernelGGL((gram_grids_kernel<KP, 1>), dim3((unsigned)qp.P4, (unsigned)cnt), dim3(64), 0, st, qa, const_cast<double *>(qa.Mpre), g0);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) return (int)e;
            }
        }
        hipFuncSetAttribute((const void *)map_quad_kernel<KP, TSF_QUAD_NTR, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((map_quad_kernel<KP, TSF_QUAD_NTR, true>), dim3((unsigned)blocks), dim3(64), lds, st, qa);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL((gram_build_kernel<KP, 1>), dim3((unsigned)qp.P4), dim3(64), 0, st, qa, Mg);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipFuncSetAttribute((const void *)map_quad_kernel<KP, TSF_QUAD_NTR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((map_quad_kernel<KP, TSF_QUAD_NTR>), dim3((unsigned)blocks), dim3(64), lds, st, qa);
    return (int)hipGetLastError();
}

int launch_map_quad(int KP, const QuadPlan &qp, const QuadArgs &qa, double *Mg, int PM, hipStream_t st)
{
    switch (KP) {
    case 8: return launch_map_quad_one<8>(qp, qa, Mg, PM, st);
    case 16: return launch_map_quad_one<16>(qp, qa, Mg, PM, st);
    case 28: return launch_map_quad_one<28>(qp, qa, Mg, PM, st);
    default: return -1;
    }
}

int launch_eval_quad(int KP, const QuadPlan &qp, const QuadArgs &qa, double *Mg, const double *theta_ref, hipStream_t st)
{
    switch (KP * 100 + qp.P4) {
    case 840: return launch_eval_quad_one<8, 40>(qp, qa, Mg, theta_ref, st);
    case 856: return launch_eval_quad_one<8, 56>(qp, qa, Mg, theta_ref, st);
    case 864: return launch_eval_quad_one<8, 64>(qp, qa, Mg, theta_ref, st);
    case 1640: return launch_eval_quad_one<16, 40>(qp, qa, Mg, theta_ref, st);
    case 1656: return launch_eval_quad_one<16, 56>(qp, qa, Mg, theta_ref, st);
    case 1664: return launch_eval_quad_one<16, 64>(qp, qa, Mg, theta_ref, st);
    case 2840: return launch_eval_quad_one<28, 40>(qp, qa, Mg, theta_ref, st);
    case 2856: return launch_eval_quad_one<28, 56>(qp, qa, Mg, theta_ref, st);
    case 2864: return launch_eval_quad_one<28, 64>(qp, qa, Mg, theta_ref, st);
    default: return -1;
    }
}

}  // namespace tsf
