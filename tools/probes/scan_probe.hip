// dev probe (round 5): the lane scans of tsf_common.h -- prefix_scan, affine_prefix_scan, affine_suffix_scan and the
// wave-wide DPP shifts they lean on -- against a plain host statement of the same trees (what oracle/prophet_canon.c
// does), bit for bit, on random inputs.  Also prints the direction of wave_shl:1 / wave_shr:1.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include tools/probes/scan_probe.hip -o tools/probes/bin/scan_probe
#include "../../time_series_spark_amd/csrc/tsf_common.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
using namespace tsf;

__global__ void k_scans(const double *in_a, const double *in_b, double *out)
{
    const int l = threadIdx.x;
    const double a = in_a[l], b = in_b[l];
    out[0 * 64 + l] = prefix_scan(b);
    double pa = a, pb = b; affine_prefix_scan(pa, pb); out[1 * 64 + l] = pa; out[2 * 64 + l] = pb;
    double sa = a, sb = b; affine_suffix_scan(sa, sb); out[3 * 64 + l] = sa; out[4 * 64 + l] = sb;
    out[5 * 64 + l] = dpp_mov<DPP_WAVE_SHL1>((double)l + 100.0);
    out[6 * 64 + l] = dpp_mov<DPP_WAVE_SHR1>((double)l + 100.0);
    out[7 * 64 + l] = suffix_scan(b);
}

static void h_prefix(double *v)
{
    double n[64];
    for (int off = 1; off < 16; off <<= 1) { for (int L = 0; L < 64; ++L) n[L] = v[L] + (((L & 15) >= off) ? v[L - off] : 0.0); memcpy(v, n, sizeof(n)); }
    const double t0 = v[15], t1 = v[31], t2 = v[47], s1 = t0 + t1, s2 = s1 + t2;
    for (int L = 0; L < 64; ++L) { const int r = L >> 4; v[L] = v[L] + (r == 0 ? 0.0 : (r == 1 ? t0 : (r == 2 ? s1 : s2))); }
}
static void h_suffix(double *v)
{
    double n[64];
    for (int off = 1; off < 16; off <<= 1) { for (int L = 0; L < 64; ++L) n[L] = v[L] + (((L & 15) + off < 16) ? v[L + off] : 0.0); memcpy(v, n, sizeof(n)); }
    const double t1 = v[16], t2 = v[32], t3 = v[48], s2 = t2 + t3, s1 = t1 + s2;
    for (int L = 0; L < 64; ++L) { const int r = L >> 4; v[L] = v[L] + (r == 0 ? s1 : (r == 1 ? s2 : (r == 2 ? t3 : 0.0))); }
}
static void h_affine(double *a, double *b, bool prefix)
{
    double na[64], nb[64];
    for (int off = 1; off < 16; off <<= 1) {
        for (int L = 0; L < 64; ++L) {
            const bool in = prefix ? ((L & 15) >= off) : ((L & 15) + off < 16);
            const int o = prefix ? L - off : L + off;
            const double ea = in ? a[o] : 1.0, eb = in ? b[o] : 0.0;
            na[L] = a[L] * ea; nb[L] = fma(a[L], eb, b[L]);
        }
        memcpy(a, na, sizeof(na)); memcpy(b, nb, sizeof(nb));
    }
    double ca[4], cb[4];
    if (prefix) {
        const double a0 = a[15], b0 = b[15], a1 = a[31], b1 = b[31], a2 = a[47], b2 = b[47];
        ca[0] = 1.0; cb[0] = 0.0; ca[1] = a0; cb[1] = b0; ca[2] = a1 * a0; cb[2] = fma(a1, b0, b1); ca[3] = a2 * ca[2]; cb[3] = fma(a2, cb[2], b2);
    } else {
        const double a1 = a[16], b1 = b[16], a2 = a[32], b2 = b[32], a3 = a[48], b3 = b[48];
        ca[3] = 1.0; cb[3] = 0.0; ca[2] = a3; cb[2] = b3; ca[1] = a2 * a3; cb[1] = fma(a2, b3, b2); ca[0] = a1 * ca[1]; cb[0] = fma(a1, cb[1], b1);
    }
    for (int L = 0; L < 64; ++L) { const int r = L >> 4; na[L] = a[L] * ca[r]; nb[L] = fma(a[L], cb[r], b[L]); }
    memcpy(a, na, sizeof(na)); memcpy(b, nb, sizeof(nb));
}

int main()
{
    double *da, *db, *dout;
    hipMalloc((void **)&da, 64 * 8); hipMalloc((void **)&db, 64 * 8); hipMalloc((void **)&dout, 8 * 64 * 8);
    std::mt19937_64 rng(5);
    std::uniform_real_distribution<double> ua(0.9, 1.1), ub(-1.0, 1.0);
    long bad = 0;
    std::vector<double> out(8 * 64);
    for (int trial = 0; trial < 200; ++trial) {
        double a[64], b[64];
        for (int l = 0; l < 64; ++l) { a[l] = ua(rng); b[l] = ub(rng); }
        if (trial & 1) for (int l = 28; l < 64; ++l) { a[l] = 1.0; b[l] = 0.0; }      // identity padding, as the kernels use it
        hipMemcpy(da, a, sizeof(a), hipMemcpyHostToDevice); hipMemcpy(db, b, sizeof(b), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_scans, dim3(1), dim3(64), 0, 0, da, db, dout);
        if (hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAILED (hip)\n"); return 1; }
        double p[64], s[64], pa[64], pb[64], sa[64], sb[64];
        memcpy(p, b, sizeof(b)); h_prefix(p); memcpy(s, b, sizeof(b)); h_suffix(s);
        memcpy(pa, a, sizeof(a)); memcpy(pb, b, sizeof(b)); h_affine(pa, pb, true);
        memcpy(sa, a, sizeof(a)); memcpy(sb, b, sizeof(b)); h_affine(sa, sb, false);
        for (int l = 0; l < 64; ++l) {
            bad += memcmp(&out[0 * 64 + l], &p[l], 8) != 0; bad += memcmp(&out[7 * 64 + l], &s[l], 8) != 0;
            bad += memcmp(&out[1 * 64 + l], &pa[l], 8) != 0; bad += memcmp(&out[2 * 64 + l], &pb[l], 8) != 0;
            bad += memcmp(&out[3 * 64 + l], &sa[l], 8) != 0; bad += memcmp(&out[4 * 64 + l], &sb[l], 8) != 0;
        }
    }
    printf("wave_shl:1  lane 0 reads %.0f (lane 1 = 101), lane 63 reads %.0f (nothing = 0)\n", out[5 * 64 + 0], out[5 * 64 + 63]);
    printf("wave_shr:1  lane 0 reads %.0f (nothing = 0), lane 1 reads %.0f (lane 0 = 100), lane 16 reads %.0f (lane 15 = 115)\n", out[6 * 64 + 0], out[6 * 64 + 1], out[6 * 64 + 16]);
    const bool dir_ok = out[5 * 64 + 0] == 101.0 && out[5 * 64 + 63] == 0.0 && out[6 * 64 + 0] == 0.0 && out[6 * 64 + 1] == 100.0 && out[6 * 64 + 16] == 115.0 && out[5 * 64 + 15] == 116.0;
    printf("scans: %ld mismatching values over 200 trials; shifts %s\n", bad, dir_ok ? "as assumed" : "NOT as assumed");
    printf("%s\n", (bad == 0 && dir_ok) ? "SCAN_PROBE_OK" : "SCAN_PROBE_FAILED");
    return (bad == 0 && dir_ok) ? 0 : 1;
}
