// mfma_f64_probe.hip -- what does v_mfma_f64_16x16x4_f64 compute, bit for bit, on gfx950?
//
// The canonical-arithmetic contract needs the exact association and
// rounding of every floating-point operation, so before the matrix cores are used for any sum
// the oracle has to be able to restate them.  This probe feeds random (cancellation-heavy)
// operands through one MFMA and compares the result with candidate CPU statements:
//   seq      : fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c))))       k ascending, C first
//   rev      : k descending, C first
//   seq_clast: ((a0b0 (+) a1b1 ...) chain from 0) + c last
//   pair     : (fma(a1,b1,a0*b0) + fma(a3,b3,a2*b2)) + c
//   exact    : one rounding of the exact sum (__float128 is wide enough for 4 products here)
// and verifies the operand / result lane layout with small integers (exact in any order).
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/probes/mfma_f64_probe.hip -o tools/probes/bin/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void mfma_kernel(const double *A, const double *B, const double *C, double *D, int n)
{
    // A: [n][16][4], B: [n][4][16], C, D: [n][16][16]
    const int l = threadIdx.x;
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        const double a = A[(size_t)t * 64 + (l % 16) * 4 + (l / 16)];
        const double b = B[(size_t)t * 64 + (l / 16) * 16 + (l % 16)];
        double4_t c;
        for (int r = 0; r < 4; ++r) c[r] = C[(size_t)t * 256 + (4 * r + (l / 16)) * 16 + (l % 16)];
        double4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) D[(size_t)t * 256 + (4 * r + (l / 16)) * 16 + (l % 16)] = d[r];
    }
}

// two chained MFMAs (the accumulator of the first is the C of the second)
__global__ void mfma_chain_kernel(const double *A, const double *B, const double *C, double *D, int n)
{
    const int l = threadIdx.x;
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        double4_t c;
        for (int r = 0; r < 4; ++r) c[r] = C[(size_t)t * 256 + (4 * r + (l / 16)) * 16 + (l % 16)];
        for (int rep = 0; rep < 2; ++rep) {
            const size_t tt = ((size_t)t + rep) % n;
            const double a = A[tt * 64 + (l % 16) * 4 + (l / 16)];
            const double b = B[tt * 64 + (l / 16) * 16 + (l % 16)];
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) D[(size_t)t * 256 + (4 * r + (l / 16)) * 16 + (l % 16)] = c[r];
    }
}

// throughput: NACC independent accumulators per wave, 4 waves per workgroup, every CU busy
template <int NACC>
__global__ void mfma_rate_kernel(double *out, int iters)
{
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static void rate(const char *what, int waves_per_block)
{
    double *out;
    const int blocks = 256 * 4, iters = 20000;
    hipMalloc(&out, (size_t)blocks * 64 * waves_per_block * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_rate_kernel<NACC>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, 100);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_rate_kernel<NACC>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)blocks * waves_per_block * iters * NACC;
    printf("%s: %.3f ms, %.1f TFLOP/s f64 (%d acc/wave, %d waves/block, %d blocks); cycles per MFMA per SIMD at 2.4 GHz: %.1f\n",
           what, ms, n_mfma * 2048.0 / (ms * 1e-3) / 1e12, NACC, waves_per_block, blocks,
           (ms * 1e-3) * 2.4e9 / (n_mfma / (256.0 * 4.0)));
    hipFree(out);
}

static double rnd(unsigned long long &s)
{
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    const double u = (double)(s >> 11) / 9007199254740992.0;      // [0,1)
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    const int e = (int)((s >> 33) % 41) - 20;
    return ldexp(2.0 * u - 1.0, e);
}

int main()
{
    const int n = 4096;
    std::vector<double> A((size_t)n * 64), B((size_t)n * 64), C((size_t)n * 256), D((size_t)n * 256), D2((size_t)n * 256);
    unsigned long long s = 751;
    for (auto &v : A) v = rnd(s);
    for (auto &v : B) v = rnd(s);
    for (auto &v : C) v = rnd(s);
    // layout check block: tile 0 uses small integers
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = (double)(1 + i + 17 * k);
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (double)(3 + 5 * j + 101 * k);
    for (int i = 0; i < 256; ++i) C[i] = (double)(i % 7);
    // cancellation block: tiles 1..1023 make c ~ -(a0 b0) so that rounding order shows
    for (int t = 1; t < 1024; ++t)
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            const double *a = &A[(size_t)t * 64 + i * 4];
            const double p = a[0] * B[(size_t)t * 64 + j] + a[2] * B[(size_t)t * 64 + 32 + j];
            C[(size_t)t * 256 + i * 16 + j] = -p * (1.0 + 1e-9 * (double)((i * 16 + j) % 5));
        }
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dD, D.size() * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_kernel, dim3(64), dim3(64), 0, 0, dA, dB, dC, dD, n);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(mfma_chain_kernel, dim3(64), dim3(64), 0, 0, dA, dB, dC, dD, n);
    if (hipDeviceSynchronize() != hipSuccess) { printf("chain kernel failed\n"); return 1; }
    hipMemcpy(D2.data(), dD, D2.size() * 8, hipMemcpyDeviceToHost);

    // layout check
    int lay_bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double e = C[i * 16 + j];
        for (int k = 0; k < 4; ++k) e += A[i * 4 + k] * B[k * 16 + j];
        if (e != D[i * 16 + j]) lay_bad++;
    }
    printf("layout check (a=A[l%%16][l/16], b=B[l/16][l%%16], c[r]=C[4*r+l/16][l%%16]): %s (%d mismatches)\n",
           lay_bad ? "WRONG" : "ok", lay_bad);

    const char *names[] = {"seq (k ascending, C first)", "rev (k descending, C first)", "seq chain from 0, + C last",
                           "pairwise + C", "exact sum, one rounding", "seq, products rounded (no fma)"};
    long match[6] = {0, 0, 0, 0, 0, 0}, total = 0, match_chain = 0;
    for (int t = 1; t < n; ++t)
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            const double *a = &A[(size_t)t * 64 + i * 4];
            double b[4];
            for (int k = 0; k < 4; ++k) b[k] = B[(size_t)t * 64 + k * 16 + j];
            const double c = C[(size_t)t * 256 + i * 16 + j];
            const double got = D[(size_t)t * 256 + i * 16 + j];
            double v[6];
            v[0] = fma(a[3], b[3], fma(a[2], b[2], fma(a[1], b[1], fma(a[0], b[0], c))));
            v[1] = fma(a[0], b[0], fma(a[1], b[1], fma(a[2], b[2], fma(a[3], b[3], c))));
            v[2] = fma(a[3], b[3], fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0]))) + c;
            v[3] = (fma(a[1], b[1], a[0] * b[0]) + fma(a[3], b[3], a[2] * b[2])) + c;
            {
                __float128 e = (__float128)c;
                for (int k = 0; k < 4; ++k) e += (__float128)a[k] * (__float128)b[k];
                v[4] = (double)e;
            }
            v[5] = (((c + a[0] * b[0]) + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3];
            for (int m = 0; m < 6; ++m) if (memcmp(&v[m], &got, 8) == 0) match[m]++;
            total++;
            // chained: second MFMA uses tile t+1's operands on top of the first result
            const size_t t2 = ((size_t)t + 1) % n;
            const double *a2 = &A[t2 * 64 + i * 4];
            double w = v[0];
            for (int k = 0; k < 4; ++k) w = fma(a2[k], B[t2 * 64 + k * 16 + j], w);
            if (memcmp(&w, &D2[(size_t)t * 256 + i * 16 + j], 8) == 0) match_chain++;
        }
    for (int m = 0; m < 6; ++m) printf("%-34s: %ld / %ld bit-identical\n", names[m], match[m], total);
    printf("two chained MFMAs == 8 sequential fma : %ld / %ld bit-identical\n", match_chain, total);
    rate<1>("dependent chain, 1 wave per block", 1);
    rate<4>("4 independent accumulators, 1 wave per block", 1);
    rate<4>("4 independent accumulators, 4 waves per block", 4);
    return 0;
}
