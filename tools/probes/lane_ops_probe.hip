// dev probe: latency of the cross-lane moves the reductions are built from, on one wave of gfx950:
// a dependent chain of 64-bit "move + add" steps (the shape of one butterfly stage), cycles per step.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include tools/probes/lane_ops_probe.hip -o tools/probes/bin/lane_ops_probe
#include "../../time_series_spark_amd/csrc/tsf_common.h"
#include <cstdio>
#include <vector>
using namespace tsf;

template <int KIND>
__device__ __forceinline__ double step(double v)
{
    if (KIND == 0) return v + dpp_mov<DPP_XOR1>(v);                       // quad_perm DPP
    if (KIND == 1) return v + dpp_mov<DPP_MIRROR>(v);                     // row_mirror DPP
    if (KIND == 2) return v + dpp_mov<DPP_ROW_BCAST15>(v);                // row_bcast
    if (KIND == 3) { double a = v, b = v; swap16(a, b); return a + b; }   // v_permlane16_swap
    if (KIND == 4) { double a = v, b = v; swap32(a, b); return a + b; }   // v_permlane32_swap
    if (KIND == 5) return v + readlane_f64(v, 63);                        // v_readlane + add
    if (KIND == 6) return v + __shfl_xor(v, 16, 64);                      // ds_bpermute
    if (KIND == 7) return v + 1.0;                                        // plain dependent add
    if (KIND == 8) return __builtin_fma(v, 1.0000001, 0.5);               // plain dependent fma
    return v;
}

template <int KIND>
__global__ void chain(double *out, long long *cyc, int n)
{
    double v = (double)threadIdx.x * 1e-9;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) v = step<KIND>(v);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[KIND] = t1 - t0;
}

int main()
{
    double *out; long long *cyc;
    hipMalloc((void **)&out, 64 * 8); hipMalloc((void **)&cyc, 16 * 8);
    hipMemset(cyc, 0, 16 * 8);
    const int n = 256;
    hipLaunchKernelGGL(chain<0>, dim3(1), dim3(64), 0, 0, out, cyc, n);
    hipLaunchKernelGGL(chain<1>, dim3(1), dim3(64), 0, 0, out, cyc, n);
    hipLaunchKernelGGL(chain<2>, dim3(1), dim3(64), 0, 0, out, cyc, n);
    hipLaunchKernelGGL(chain<3>, dim3(1), dim3(64), 0, 0, out, cyc, n);
    hipLaunchKernelGGL(chain<4>, dim3(1), dim3(64), 0, 0, out, cyc, n);
    hipLaunchKernelGGL(chain<5>, dim3(1), dim3(64), 0, 0, out, cyc, n);
    hipLaunchKernelGGL(chain<6>, dim3(1), dim3(64), 0, 0, out, cyc, n);
    hipLaunchKernelGGL(chain<7>, dim3(1), dim3(64), 0, 0, out, cyc, n);
    hipLaunchKernelGGL(chain<8>, dim3(1), dim3(64), 0, 0, out, cyc, n);
    std::vector<long long> h(16);
    if (hipMemcpy(h.data(), cyc, 16 * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAILED\n"); return 1; }
    const char *names[] = {"quad_perm DPP + add", "row_mirror DPP + add", "row_bcast15 DPP + add", "permlane16_swap + add",
                           "permlane32_swap + add", "readlane(63) + add", "ds_bpermute (shfl_xor 16) + add", "add only", "fma only"};
    for (int k = 0; k < 9; ++k)
        printf("%-34s %8.1f cycles per dependent step (s_memtime ticks / %d steps)\n", names[k], (double)h[k] / (n * 16), n * 16);
    return 0;
}
