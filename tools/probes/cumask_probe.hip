// probe: which compute units a hipExtStreamCreateWithCUMask stream runs on (XCC id, SE/CU id per workgroup), and
// whether kernels on two masked streams run side by side (one spins until the other has started).
// hipcc --offload-arch=gfx950 -O2 tools/probes/cumask_probe.hip -o tools/probes/bin/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

__global__ void where_kernel(unsigned *out, int spin)
{
    unsigned xcc = 0, hwid = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}

__global__ void waiter_kernel(volatile int *flag, int *result, long long max_cycles)
{
    long long t0 = __builtin_readcyclecounter();
    int seen = 0;
    while (__builtin_readcyclecounter() - t0 < max_cycles) {
        if (__hip_atomic_load((int *)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { seen = 1; break; }
        __builtin_amdgcn_s_sleep(32);
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { result[0] = seen; result[1] = (int)((__builtin_readcyclecounter() - t0) / 1000); }
}
__global__ void setter_kernel(int *flag) { if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

static void report(const char *tag, const std::vector<unsigned> &h, int n)
{
    std::map<unsigned, std::set<unsigned>> per;
    for (int i = 0; i < n; ++i) {
        const unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        per[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    printf("%s:", tag);
    int total = 0;
    for (auto &kv : per) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  distinct (xcc, se, sh, cu) = %d\n", total);
}

int main()
{
    const int NB = 4096;
    unsigned *d;
    hipMalloc(&d, sizeof(unsigned) * 2 * NB);
    std::vector<unsigned> h(2 * NB);
    auto run = [&](const char *tag, hipStream_t st) {
        hipMemsetAsync(d, 0, sizeof(unsigned) * 2 * NB, st);
        hipLaunchKernelGGL(where_kernel, dim3(NB), dim3(512), 0, st, d, 20000);
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * NB, hipMemcpyDeviceToHost);
        report(tag, h, NB);
    };
    run("default stream", nullptr);
    const unsigned masks[][8] = {
        {0x0000ffffu, 0, 0, 0, 0, 0, 0, 0},
        {0xffffffffu, 0, 0, 0, 0, 0, 0, 0},
        {0x00000003u, 0x00000003u, 0x00000003u, 0x00000003u, 0x00000003u, 0x00000003u, 0x00000003u, 0x00000003u},
        {0x01010101u, 0x01010101u, 0, 0, 0, 0, 0, 0},
        {0xfffffffcu, 0xfffffffcu, 0xfffffffcu, 0xfffffffcu, 0xfffffffcu, 0xfffffffcu, 0xfffffffcu, 0xfffffffcu},
    };
    const char *names[] = {"mask bits 0-15", "mask bits 0-31", "mask bits 0,1 of every word", "mask every 8th bit of words 0,1", "mask all but bits 0,1 of every word"};
    for (int m = 0; m < 5; ++m) {
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, masks[m]);
        if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", names[m], hipGetErrorString(e)); continue; }
        run(names[m], st);
        hipStreamDestroy(st);
    }
    // side by side?
    int *flag, *res;
    hipMalloc(&flag, 4); hipMalloc(&res, 8);
    for (int variant = 0; variant < 2; ++variant) {
        hipStream_t a, b;
        if (variant == 0) { hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking); }
        else { hipExtStreamCreateWithCUMask(&a, 8, masks[2]); hipExtStreamCreateWithCUMask(&b, 8, masks[4]); }
        hipMemset(flag, 0, 4); hipMemset(res, 0, 8);
        hipLaunchKernelGGL(waiter_kernel, dim3(16), dim3(512), 0, a, (volatile int *)flag, res, 2400000000ll / 4);   // <= 0.25 s
        hipLaunchKernelGGL(where_kernel, dim3(NB * 4), dim3(64), 0, b, d, 20000);
        hipLaunchKernelGGL(setter_kernel, dim3(1), dim3(64), 0, b, flag);
        hipDeviceSynchronize();
        int hr[2];
        hipMemcpy(hr, res, 8, hipMemcpyDeviceToHost);
        printf("%s streams: waiter saw the flag %d after %d kcycles\n", variant ? "masked" : "plain", hr[0], hr[1]);
        hipStreamDestroy(a); hipStreamDestroy(b);
    }
    return 0;
}
