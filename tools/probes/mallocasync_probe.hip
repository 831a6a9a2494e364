// dev probe (round 5; round-4 review item 2): is stream-ordered scratch (hipMallocAsync / hipFreeAsync) of gigabyte size
// safe on this ROCm when it is used the way libtsf_amd used it in round 3 -- NO library code here.
//
// Round 3 saw intermittently wrong Newton fits after a LARGE call had taken ~6 GB of hipMallocAsync scratch for the
// slot records of newton_batch_kernel; the library moved to cached hipMalloc blocks and the cause was never found.
// This program repeats the allocation pattern with a kernel whose every byte is checkable:
//   per call: p = hipMallocAsync(size_i, stream); kernel W writes pattern(call, index) into p in "rounds" the way the
//   slot kernel does (a block owns a slice, writes a record, re-reads and updates it R times); kernel V verifies every
//   word; hipFreeAsync(p, stream) right behind the launches (as the library did); sizes cycle small -> 6 GB -> small;
//   in between: a synchronous hipMemcpy (what the host-pointer wrappers do), a hipMalloc + hipFree of an unrelated
//   buffer (ensure_ws growing: hipFree synchronises the device and may hand the pool's memory back), a long-lived
//   hipMalloc'd "workspace" whose contents are re-verified after every call (aliasing with pool memory), and a second
//   stream that allocates from the same pool while the first stream's kernels still run.
// Modes (argv[1], bit mask): 1 release threshold = max (the pool never returns memory); 2 no second stream;
// 4 no unrelated hipMalloc / hipFree between the calls; 8 no synchronous hipMemcpy behind the free (only the
// hipDeviceSynchronize at the end of the call); 16 CONTROL: the same kernels and checks on plain hipMalloc / hipFree
// blocks (hipFree after the device has been synchronised) -- what the library does.
// Prints the number of corrupted words per phase; exit code 0 = nothing wrong seen.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mallocasync_probe.hip -o tools/probes/bin/mallocasync_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

__device__ __host__ inline uint64_t pat(uint64_t call, uint64_t i) { uint64_t z = (call + 1) * 0x9E3779B97F4A7C15ull + i * 0xBF58476D1CE4E5B9ull; z ^= z >> 31; return z * 0x94D049BB133111EBull; }

// each block owns a contiguous slice; R rounds of read-modify-write the way a slot record is used
__global__ void k_write(uint64_t *p, size_t n, uint64_t call, int rounds)
{
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) p[i] = pat(call, i) - (uint64_t)rounds;
    for (int r = 0; r < rounds; ++r) {
        __syncthreads();
        // read what a NEIGHBOUR thread of the block wrote in the round before (global-memory visibility inside a block)
        for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            const size_t j = lo + (i - lo + 1) % (hi - lo);
            const uint64_t vj = p[j];
            if (vj != pat(call, j) - (uint64_t)(rounds - r)) atomicAdd((unsigned long long *)(p + n), 1ull);     // p[n]: error counter
        }
        __syncthreads();
        for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) p[i] = p[i] + 1;
    }
}
__global__ void k_verify(const uint64_t *p, size_t n, uint64_t call, unsigned long long *bad)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (p[i] != pat(call, i)) atomicAdd(bad, 1ull);
}
__global__ void k_fill(uint64_t *p, size_t n, uint64_t call)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = pat(call, i);
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    int rt = 0, drv = 0;
    hipRuntimeGetVersion(&rt); hipDriverGetVersion(&drv);
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, HIP runtime %d, driver %d, mode %d\n", prop.name, rt, drv, mode);
    hipStream_t s1, s2; CHECK(hipStreamCreate(&s1)); CHECK(hipStreamCreate(&s2));
    if (mode & 1) {
        hipMemPool_t pool; CHECK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t thr = UINT64_MAX; CHECK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
    }
    unsigned long long *bad; CHECK(hipMalloc((void **)&bad, 64)); CHECK(hipMemset(bad, 0, 64));
    // the long-lived workspace (cached hipMalloc block of a context)
    const size_t wsn = (size_t)96 << 20;           // 768 MB of words
    uint64_t *ws; CHECK(hipMalloc((void **)&ws, wsn * 8));
    k_fill<<<1024, 256, 0, s1>>>(ws, wsn, 777); CHECK(hipStreamSynchronize(s1));
    const double gb[] = {0.05, 0.3, 0.05, 6.0, 0.05, 0.3, 2.5, 0.05, 6.0, 0.02, 0.3, 0.05, 4.0, 0.3, 0.05, 0.05};
    std::vector<uint64_t> host(1 << 20);
    unsigned long long total_bad = 0;
    for (int streams = 0; streams < 2; ++streams) {        // 0: the null stream (the library's host-pointer entry points), 1: a created stream
        hipStream_t st = streams ? s1 : nullptr;
        for (int call = 0; call < 16; ++call) {
            const size_t n = (size_t)(gb[call] * 1e9 / 8);
            uint64_t *p = nullptr;
            if (mode & 16) CHECK(hipMalloc((void **)&p, (n + 8) * 8));
            else CHECK(hipMallocAsync((void **)&p, (n + 8) * 8, st));
            CHECK(hipMemsetAsync(p + n, 0, 64, st));
            k_write<<<1024, 64, 0, st>>>(p, n, (uint64_t)call + 100 * streams, 6);
            k_verify<<<2048, 256, 0, st>>>(p, n, (uint64_t)call + 100 * streams, bad);
            // a second stream takes pool memory while the kernels above are still running
            if (!(mode & 2)) {
                uint64_t *q = nullptr;
                const size_t qn = (size_t)32 << 20;
                CHECK(hipMallocAsync((void **)&q, qn * 8, s2));
                k_fill<<<512, 256, 0, s2>>>(q, qn, 5000 + call);
                k_verify<<<512, 256, 0, s2>>>(q, qn, 5000 + call, bad + 1);
                CHECK(hipFreeAsync(q, s2));
            }
            unsigned long long inner = 0;
            CHECK(hipMemcpyAsync(&inner, p + n, 8, hipMemcpyDeviceToHost, st));
            if (!(mode & 16)) CHECK(hipFreeAsync(p, st));  // right behind the launches, as the library did
            // what the host-pointer wrappers do next: a synchronous copy on the null stream
            if (!(mode & 8)) CHECK(hipMemcpy(host.data(), ws, host.size() * 8, hipMemcpyDeviceToHost));
            if (call % 3 == 1 && !(mode & 4)) {                           // ensure_ws growing: an unrelated hipMalloc + hipFree (device sync)
                void *u = nullptr; CHECK(hipMalloc(&u, (size_t)(1 + call % 4) << 28)); CHECK(hipMemset(u, 0xA5, 1 << 20)); CHECK(hipFree(u));
            }
            CHECK(hipStreamSynchronize(s2)); CHECK(hipDeviceSynchronize());
            if (mode & 16) CHECK(hipFree(p));
            unsigned long long hb[8];
            CHECK(hipMemcpy(hb, bad, 64, hipMemcpyDeviceToHost));
            // the workspace must be untouched
            k_verify<<<2048, 256, 0, s1>>>(ws, wsn, 777, bad + 2); CHECK(hipStreamSynchronize(s1));
            unsigned long long wb = 0; CHECK(hipMemcpy(&wb, bad + 2, 8, hipMemcpyDeviceToHost));
            printf("stream %s call %2d size %5.2f GB: corrupted words after the rounds %llu, inside the rounds %llu, second stream %llu, workspace %llu\n",
                   streams ? "s1  " : "null", call, gb[call], hb[0], inner, hb[1], wb);
            total_bad += hb[0] + inner + hb[1] + wb;
            CHECK(hipMemset(bad, 0, 64));
        }
    }
    printf("%s (%llu corrupted words in all)\n", total_bad == 0 ? "MALLOCASYNC_PROBE_CLEAN" : "MALLOCASYNC_PROBE_CORRUPTION", total_bad);
    return total_bad == 0 ? 0 : 1;
}
