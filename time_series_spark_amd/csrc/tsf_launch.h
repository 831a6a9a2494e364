// tsf_launch.h -- launcher entry points of the per-(GROWTH, MODE) kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
namespace tsf {
struct FitArgs;
struct QuadArgs;
struct MfmaTabs;
struct QuadPlan { int P4, PPL, NW, blocks, slots, n_cu; const int *opt; };      // opt: the context's route switches (TSF_OPT_*, -1 = default)
int quad_waves_per_block(int PPL);
// gram_build_kernel + fit_quad_kernel (tsf_inst_quad.hip)
int launch_quad(int KP, const QuadPlan &qp, const QuadArgs &qa, double *Mg, hipStream_t st);
int launch_quad_aligned1(int KP, const QuadPlan &qp, const QuadArgs &qa, double *Mg, hipStream_t st);
// gram_build_kernel + eval_quad_kernel: one quadratic-form evaluation per series around theta_ref (tsf_eval_quadratic; tsf_inst_quad3.hip)
int launch_eval_quad(int KP, const QuadPlan &qp, const QuadArgs &qa, double *Mg, const double *theta_ref, hipStream_t st);
// gram_build_kernel + map_quad_kernel: converge = MAP of linear / additive models on aligned panels, directly (tsf_map_quad.h);
// -1: this shape has no such kernel
int launch_map_quad(int KP, const QuadPlan &qp, const QuadArgs &qa, double *Mg, int PM, hipStream_t st);
int launch_quad_aligned_reg(int KP, const QuadPlan &qp, const QuadArgs &qa, double *Mg, hipStream_t st);   // -2: no such variant
// gram_build_kernel + newton_quad_kernel (Stan's Newton, quadratic-form evaluations; tsf_inst_quad.hip)
int launch_newton_quad(int KP, const QuadPlan &qp, const QuadArgs &qa, double *Mg, int PM, int n_cu, hipStream_t st);
// bytes of slot records the several-series-per-wave Newton kernel needs for this call (QuadArgs::nb_buf); 0: not that kernel
size_t newton_batch_scratch_bytes(int KP, int PM, int64_t N, int NTmax, int n_cu, const int *opt);
int launch_g0m0(int KP, const FitArgs &a, int eval_only, hipStream_t st);
int launch_g0m1(int KP, const FitArgs &a, int eval_only, hipStream_t st);
int launch_g0m2(int KP, const FitArgs &a, int eval_only, hipStream_t st);
int launch_g1m0(int KP, const FitArgs &a, int eval_only, hipStream_t st);
int launch_g1m1(int KP, const FitArgs &a, int eval_only, hipStream_t st);
int launch_g1m2(int KP, const FitArgs &a, int eval_only, hipStream_t st);
// map_kernel (tsf_spec.converge = MAP: the continuation of every fitted series to the maximum a posteriori estimate; tsf_map_kernels.h)
int launch_map_g0m0(int KP, const FitArgs &a, hipStream_t st);
int launch_map_g0m1(int KP, const FitArgs &a, hipStream_t st);
int launch_map_g0m2(int KP, const FitArgs &a, hipStream_t st);
int launch_map_g1m0(int KP, const FitArgs &a, hipStream_t st);
int launch_map_g1m1(int KP, const FitArgs &a, hipStream_t st);
int launch_map_g1m2(int KP, const FitArgs &a, hipStream_t st);
// fit_mfma_kernel (aligned panels, residual form, 16 series per workgroup on the matrix cores;
// tsf_inst_mfma.hip)
size_t mfma_lds_bytes(int KP);
int launch_mfma_layout(const FitArgs &a, int KP, const MfmaTabs &mt, double *XF, double *XB, double *XT,
                       double *tq, uint16_t *cq, int8_t *cpof, double *yq, int *overflow, hipStream_t st);
int launch_mfma(int KP, int growth, int mode, const FitArgs &a, const MfmaTabs &mt, int blocks, hipStream_t st);
// newton_kernel (Stan's Newton optimiser; P <= 64, one explicit-mode kernels only)
int launch_newton_g0m0(int KP, const FitArgs &a, int P, hipStream_t st);
int launch_newton_g0m1(int KP, const FitArgs &a, int P, hipStream_t st);
int launch_newton_g0m2(int KP, const FitArgs &a, int P, hipStream_t st);
int launch_newton_g1m0(int KP, const FitArgs &a, int P, hipStream_t st);
int launch_newton_g1m1(int KP, const FitArgs &a, int P, hipStream_t st);
int launch_newton_g1m2(int KP, const FitArgs &a, int P, hipStream_t st);
}  // namespace tsf
