// tsf_inst_quad3.hip -- the quadratic-form fit kernel for aligned panels with P <= 64 (shared
// Z^T Z in LDS): the variant compiled for three waves per SIMD (L-BFGS history in LDS, <= 168
// VGPRs, 12 waves per CU).  Its own translation unit because it is built with
// -mllvm -disable-machine-licm (build.py): with machine LICM on, the compiler hoists the fp64
// literals of the exp / cubic-interpolation code into ~40 VGPRs for the whole kernel and then
// spills them to scratch, reloading them in the middle of the dependent chains.
#include "tsf_quad_launch.h"

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

}  // namespace tsf
