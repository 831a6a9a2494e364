// tsf_inst_quad4.hip -- the quadratic-form fit kernel for aligned panels with the shared Z^T Z held in
// REGISTERS (lane p: row p): 8 waves per CU at 256 VGPRs, history and residual staging in LDS.
// A wave's evaluation no longer waits on 54 LDS reads, which is what bounds the longest series of a
// launch (and with it the launch).  Built like tsf_inst_quad3.hip (no machine LICM).
#include "tsf_quad_launch.h"

namespace tsf {

int launch_quad_aligned_reg(int KP, const QuadPlan &qp, const QuadArgs &qa, double *Mg, hipStream_t st)
{
    switch (KP * 100 + qp.P4) {
    case 840: return launch_quad_mm<8, 1, QM_GLOBAL_REG, 40>(qp, qa, Mg, st);
    case 856: return launch_quad_mm<8, 1, QM_GLOBAL_REG, 56>(qp, qa, Mg, st);
    case 1640: return launch_quad_mm<16, 1, QM_GLOBAL_REG, 40>(qp, qa, Mg, st);
    case 1656: return launch_quad_mm<16, 1, QM_GLOBAL_REG, 56>(qp, qa, Mg, st);
    case 2840: return launch_quad_mm<28, 1, QM_GLOBAL_REG, 40>(qp, qa, Mg, st);
    case 2856: return launch_quad_mm<28, 1, QM_GLOBAL_REG, 56>(qp, qa, Mg, st);
    default: return -2;
    }
}

}  // namespace tsf
