// tsf_launch.h -- launcher entry points of the per-(GROWTH, MODE) kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
namespace tsf {
struct FitArgs;
int launch_g0m0(int KP, const FitArgs &a, int eval_only, hipStream_t st);
int launch_g0m1(int KP, const FitArgs &a, int eval_only, hipStream_t st);
int launch_g0m2(int KP, const FitArgs &a, int eval_only, hipStream_t st);
int launch_g1m0(int KP, const FitArgs &a, int eval_only, hipStream_t st);
int launch_g1m1(int KP, const FitArgs &a, int eval_only, hipStream_t st);
int launch_g1m2(int KP, const FitArgs &a, int eval_only, hipStream_t st);
}  // namespace tsf
