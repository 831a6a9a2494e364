// dev probe (round 5): dependent-chain latency, in shader cycles, of the building blocks of one evaluation as a LONE wave
// sees them (the cooperative kernel's waves mostly run alone on their SIMD): fp64 fma / add, IEEE division, square root,
// dm_exp_sel, a DPP move + add, v_readlane, the butterfly and the scans of tsf_common.h, an LDS round trip.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probes/latency_probe.hip -o tools/probes/bin/latency_probe
#include "../../time_series_spark_amd/csrc/tsf_common.h"
#include "../../time_series_spark_amd/csrc/tsf_detmath.h"
#include <cstdio>
#include <vector>
using namespace tsf;

constexpr int REPS = 512;
template <class F>
__device__ __forceinline__ void timed(int k, long long *out, double &v, F f)
{
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < REPS; ++i) { f(v); asm volatile("" : "+v"(v)); }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[k] = t1 - t0;
}

__global__ void k_lat(long long *out, double *sink, double seed)
{
    __shared__ double lds[256];
    const int l = threadIdx.x;
    double v = seed + 1e-9 * l;
    lds[l] = v; lds[64 + l] = 0.5;
    __syncthreads();
    int k = 0;
    timed(k++, out, v, [&](double &x) { x = __builtin_fma(x, 0.999999, 1e-7); });
    timed(k++, out, v, [&](double &x) { x = x + 1e-7; });
    timed(k++, out, v, [&](double &x) { x = 1.0 / (1.0 + x * 1e-3); });
    timed(k++, out, v, [&](double &x) { x = __builtin_sqrt(x + 1.0); });
    timed(k++, out, v, [&](double &x) { x = dm_exp_sel(-x * 1e-3); });
    timed(k++, out, v, [&](double &x) { x = x + dpp_mov<DPP_XOR1>(x) * 1e-9; });
    timed(k++, out, v, [&](double &x) { x = x + dpp_mov<DPP_ROW_SHL(4)>(x) * 1e-9; });
    timed(k++, out, v, [&](double &x) { x = x + dpp_mov<DPP_WAVE_SHR1>(x) * 1e-9; });
    timed(k++, out, v, [&](double &x) { x = x + dpp_mov<DPP_ROW_BCAST15>(x) * 1e-9; });
    timed(k++, out, v, [&](double &x) { x = x + readlane_f64(x, 17) * 1e-9; });
    timed(k++, out, v, [&](double &x) { x = x * 0.5 + bfly_sum(x) * 1e-9; });
    timed(k++, out, v, [&](double &x) { x = x * 0.5 + lane63(bfly_sum_l63(x)) * 1e-9; });
    timed(k++, out, v, [&](double &x) { x = x * 0.5 + suffix_scan(x) * 1e-9; });
    timed(k++, out, v, [&](double &x) { x = x * 0.5 + prefix_scan(x) * 1e-9; });
    timed(k++, out, v, [&](double &x) { double a = 0.999, b = x; affine_suffix_scan(a, b); x = x * 0.5 + b * 1e-9; });
    timed(k++, out, v, [&](double &x) { double a = x, b = x; swap32(a, b); x = (a + b) * 0.5; });
    timed(k++, out, v, [&](double &x) { double a = x, b = x; swap16(a, b); x = (a + b) * 0.5; });
    timed(k++, out, v, [&](double &x) { lds[128 + l] = x; __builtin_amdgcn_wave_barrier(); x = lds[128 + (l ^ 1)] + 1e-9; });
    timed(k++, out, v, [&](double &x) { x = x * lds[64 + ((int)(x) & 63)] + 1.0; });
    timed(k++, out, v, [&](double &x) { x = row_bfly_sum(x) * 0.0625; });
    sink[l] = v;
}

int main()
{
    long long *d; double *s;
    hipMalloc(&d, 64 * sizeof(long long)); hipMalloc(&s, 64 * sizeof(double));
    hipMemset(d, 0, 64 * sizeof(long long));
    k_lat<<<1, 64>>>(d, s, 1.25);
    hipDeviceSynchronize();
    k_lat<<<1, 64>>>(d, s, 1.25);
    hipDeviceSynchronize();
    std::vector<long long> h(64);
    hipMemcpy(h.data(), d, 64 * sizeof(long long), hipMemcpyDeviceToHost);
    const char *names[] = {"fma", "add", "1/(1+x) (IEEE division + fma)", "sqrt", "dm_exp_sel", "dpp xor1 + fma", "dpp row_shl:4 + fma",
                           "dpp wave_shr:1 + fma", "dpp row_bcast15 + fma", "readlane + fma", "bfly_sum (+2 ops)", "bfly_sum_l63 + lane63",
                           "suffix_scan", "prefix_scan", "affine_suffix_scan", "swap32 + add", "swap16 + add", "LDS write + read (other lane)",
                           "LDS read, data-dependent address", "row_bfly_sum"};
    for (int k = 0; k < 20; ++k) printf("%-36s %8.1f cycles per dependent step\n", names[k], (double)h[k] / REPS);
    return 0;
}
