// dev probe (round 5): the reverse sweep of the logistic trend (logistic_reverse_pre / _post, tsf_fit_kernels.h) and the
// segment tables (logistic_tables_lanes) on a LONE wave, in shader cycles per call -- what the cooperative kernel's trend wave
// runs per evaluation, without its neighbours.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I time_series_spark_amd/csrc tools/probes/reverse_probe.hip -o tools/probes/bin/reverse_probe
#include "tsf_fit_kernels.h"
#include <cstdio>
#include <vector>
using namespace tsf;

struct FakeLds { double ks[NTAB + 1], mc[NTAB + 1], tp1[NTAB], tp2[NTAB], tot1[W + 1], tot2[W + 1]; };

__global__ void k_rev(long long *out, double *sink)
{
    __shared__ FakeLds lds;
    const int l = threadIdx.x;
    SeriesView sv;
    sv.S = 25;
    sv.tc_l = l < 25 ? 0.03 * (l + 1) : 0.0;
    sv.Lj_l = l < 25 ? (2 * l + 1) : 0; sv.Ljm1_l = (l >= 1 && l <= 25) ? (2 * (l - 1) + 1) : 0;
    if (l <= 25) { lds.ks[l] = 0.5 + 0.01 * l; lds.mc[l] = 0.1 + 0.001 * l; }
    if (l < 25) { lds.tp1[l] = 0.3 + l; lds.tp2[l] = 0.2 + l; }
    lds.tot1[l] = 64.0 - l; lds.tot2[l] = 32.0 - 0.5 * l;
    if (l == 0) { lds.tot1[W] = 0.0; lds.tot2[W] = 0.0; }
    __syncthreads();
    double acc = 0.0;
    constexpr int REPS = 256;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < REPS; ++i) {
        LogisticReversePre pr;
        logistic_reverse_pre(sv, lds, pr);
        acc += pr.q2;
        asm volatile("" : "+v"(acc));
        lds.ks[l <= 25 ? l : 0] = lds.ks[l <= 25 ? l : 0] + 1e-9 * acc * 0.0;
        wave_sync();
    }
    long long t1 = __builtin_readcyclecounter();
    if (l == 0) out[0] = (t1 - t0) / REPS;
    LogisticReversePre pr;
    logistic_reverse_pre(sv, lds, pr);
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < REPS; ++i) {
        double gk, gm, gd;
        logistic_reverse_post(sv, lds, pr, lds.tot1[0], lds.tot2[0], gk, gm, gd);
        acc += gk + gm + gd;
        asm volatile("" : "+v"(acc));
        if (l < 25) lds.tp1[l] = lds.tp1[l] + 1e-12 * acc * 0.0;      // (the next call depends on this one)
        wave_sync();
    }
    t1 = __builtin_readcyclecounter();
    if (l == 0) out[1] = (t1 - t0) / REPS;
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < REPS; ++i) {
        double ksn, mcn;
        logistic_tables_lanes(0.5 + 1e-9 * acc * 0.0, 0.1, 0.001 * l, sv.tc_l, 25, ksn, mcn);
        acc += ksn + mcn;
        asm volatile("" : "+v"(acc));
    }
    t1 = __builtin_readcyclecounter();
    if (l == 0) out[2] = (t1 - t0) / REPS;
    sink[l] = acc;
}

int main()
{
    long long *d; double *s;
    hipMalloc(&d, 8 * sizeof(long long)); hipMalloc(&s, 64 * sizeof(double));
    for (int rep = 0; rep < 2; ++rep) { k_rev<<<1, 64>>>(d, s); hipDeviceSynchronize(); }
    long long h[8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("logistic_reverse_pre  %lld cycles per call (lone wave, dependent calls)\nlogistic_reverse_post %lld\nlogistic_tables_lanes %lld\n", h[0], h[1], h[2]);
    return 0;
}
