// tsf_launch.h -- launcher entry points of the per-(GROWTH, MODE) kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
namespace tsf {
struct FitArgs;
struct QuadArgs;
struct QuadPlan { int P4, PPL, NW, blocks, slots; };
int quad_waves_per_block(int PPL);
// gram_build_kernel + fit_quad_kernel (tsf_inst_quad.hip)
int launch_quad(int KP, const QuadPlan &qp, const QuadArgs &qa, double *Mg, hipStream_t st);
int launch_g0m0(int KP, const FitArgs &a, int eval_only, hipStream_t st);
int launch_g0m1(int KP, const FitArgs &a, int eval_only, hipStream_t st);
int launch_g0m2(int KP, const FitArgs &a, int eval_only, hipStream_t st);
int launch_g1m0(int KP, const FitArgs &a, int eval_only, hipStream_t st);
int launch_g1m1(int KP, const FitArgs &a, int eval_only, hipStream_t st);
int launch_g1m2(int KP, const FitArgs &a, int eval_only, hipStream_t st);
// newton_kernel (Stan's Newton optimiser; P <= 64, one explicit-mode kernels only)
int launch_newton_g0m0(int KP, const FitArgs &a, int PM, hipStream_t st);
int launch_newton_g0m1(int KP, const FitArgs &a, int PM, hipStream_t st);
int launch_newton_g0m2(int KP, const FitArgs &a, int PM, hipStream_t st);
int launch_newton_g1m0(int KP, const FitArgs &a, int PM, hipStream_t st);
int launch_newton_g1m1(int KP, const FitArgs &a, int PM, hipStream_t st);
int launch_newton_g1m2(int KP, const FitArgs &a, int PM, hipStream_t st);
}  // namespace tsf
