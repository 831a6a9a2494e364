// dev probe (round 5; round-4 review item 8): what does rocprofv3's FETCH_SIZE report for the access shapes of this
// library?  MI355X_MICROARCH.md calibrates the counter for 16 B / lane streaming reads only (reports exactly 1/2) and asks
// for a calibration on a known byte count for anything else.  Four kernels stream the same 4 GiB buffer ONCE (far
// beyond the 256 MiB Infinity Cache), each with one load shape of the fit kernels; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   (tools/dev/fetch_calib.sh)
// and FETCH_SIZE x 1024 / 4 GiB is the factor to divide by.
//   k_b8   : 8 B per lane, 512 B per wave and instruction  (the step-major f64 tables: yw, tw, Xw)
//   k_b16  : 16 B per lane, 1 KiB per wave and instruction (the base pairs Bw; the guide's calibrated shape)
//   k_b2   : 2 B per lane, 128 B per wave and instruction  (the segment words cw)
//   k_b8s  : 8 B per lane, rows of 28 x 512 B read as the table kernel reads a design row (stride 512 B between loads)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fetch_calib.hip -o tools/probes/bin/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr size_t BYTES = (size_t)4 << 30;

__global__ void k_b8(const double *p, size_t n, double *out)
{
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 1.2345e-300) out[0] = acc;
}
__global__ void k_b16(const double2 *p, size_t n, double *out)
{
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 1.2345e-300) out[0] = acc;
}
__global__ void k_b2(const uint16_t *p, size_t n, double *out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0xFFFFFFFFu) out[0] = (double)acc;
}
// one wave per "series": rows of 28 columns x 64 lanes, column j of row q at p[(q * 28 + j) * 64 + lane]
__global__ void k_b8s(const double *p, size_t n_rows, double *out)
{
    double acc = 0.0;
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t q = wave; q < n_rows; q += waves) {
        const double *r = p + q * 28 * 64 + lane;
#pragma unroll
        for (int j = 0; j < 28; ++j) acc += r[j * 64];
    }
    if (acc == 1.2345e-300) out[0] = acc;
}

int main()
{
    void *buf; double *out;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc((void **)&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, BYTES);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        k_b8<<<256 * 16, 256>>>((const double *)buf, BYTES / 8, out);
        k_b16<<<256 * 16, 256>>>((const double2 *)buf, BYTES / 16, out);
        k_b2<<<256 * 16, 256>>>((const uint16_t *)buf, BYTES / 2, out);
        k_b8s<<<256 * 16, 256>>>((const double *)buf, BYTES / (28 * 64 * 8), out);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED\n"); return 1; }
    printf("streamed %zu bytes per kernel (k_b8s: %zu)\n", BYTES, (BYTES / (28 * 64 * 8)) * 28 * 64 * 8);
    return 0;
}
