// dev probe: the reduction helpers of tsf_common.h (bfly_sum: row-broadcast stages, bfly_sum4:
// transposed stages) against a plain __shfl_xor butterfly 1,2,4,8,16,32 -- bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include tools/probes/bfly_probe.hip -o tools/probes/bin/bfly_probe
#include "../../time_series_spark_amd/csrc/tsf_common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace tsf;

__device__ double ref_bfly(double v)
{
    for (int o = 1; o < 64; o <<= 1) v = v + __shfl_xor(v, o, 64);
    return v;
}

__global__ void probe(const double *in, double *out, int n_sets)
{
    const int lane = threadIdx.x;
    for (int s = blockIdx.x; s < n_sets; s += gridDim.x) {
        const double *p = in + (size_t)s * 256;
        double a = p[lane], b = p[64 + lane], c = p[128 + lane], d = p[192 + lane];
        double *o = out + (size_t)s * 12 * 64;
        o[0 * 64 + lane] = ref_bfly(a); o[1 * 64 + lane] = ref_bfly(b); o[2 * 64 + lane] = ref_bfly(c); o[3 * 64 + lane] = ref_bfly(d);
        o[4 * 64 + lane] = bfly_sum(a); o[5 * 64 + lane] = bfly_sum(b); o[6 * 64 + lane] = bfly_sum(c); o[7 * 64 + lane] = bfly_sum(d);
        double sa, sb, sc, sd;
        bfly_sum4(a, b, c, d, sa, sb, sc, sd);
        o[8 * 64 + lane] = sa; o[9 * 64 + lane] = sb; o[10 * 64 + lane] = sc; o[11 * 64 + lane] = sd;
    }
}

int main()
{
    const int n_sets = 4096;
    std::vector<double> h((size_t)n_sets * 256);
    srand(7);
    for (size_t i = 0; i < h.size(); ++i) {
        const double u = (double)rand() / RAND_MAX - 0.5;
        const int e = rand() % 40 - 20;
        h[i] = (rand() % 97 == 0) ? (rand() % 2 ? 0.0 : -0.0) : ldexp(u, e);
    }
    double *din, *dout;
    hipMalloc((void **)&din, h.size() * 8);
    hipMalloc((void **)&dout, (size_t)n_sets * 12 * 64 * 8);
    hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(256), dim3(64), 0, 0, din, dout, n_sets);
    std::vector<double> o((size_t)n_sets * 12 * 64);
    if (hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAILED: copy\n"); return 2; }
    long bad1 = 0, bad4 = 0, nonuni = 0;
    for (int s = 0; s < n_sets; ++s)
        for (int q = 0; q < 4; ++q)
            for (int l = 0; l < 64; ++l) {
                const double *b = o.data() + (size_t)s * 12 * 64;
                if (memcmp(&b[q * 64 + l], &b[q * 64], 8)) nonuni++;
                if (memcmp(&b[(4 + q) * 64 + l], &b[q * 64 + l], 8)) bad1++;
                if (memcmp(&b[(8 + q) * 64 + l], &b[q * 64 + l], 8)) bad4++;
            }
    printf("bfly probe: %d sets x 4 sums x 64 lanes: reference non-uniform %ld, bfly_sum mismatches %ld, bfly_sum4 mismatches %ld -> %s\n",
           n_sets, nonuni, bad1, bad4, (bad1 || bad4 || nonuni) ? "FAILED" : "ok");
    return (bad1 || bad4 || nonuni) ? 1 : 0;
}
