// tsf_mfma_tabs.h -- constants and table descriptor of the matrix-core residual kernel
// (tsf_mfma_kernels.h), shared with the host side (tsf_api.hip).
#pragma once
#include <stdint.h>

namespace tsf {

typedef double d4_t __attribute__((ext_vector_type(4)));

constexpr int MT_NS = 16;       // series slots per workgroup
constexpr int MT_NW = 8;        // waves per workgroup (512 threads: 256 VGPRs per lane, no spills)
constexpr int MT_SPW = MT_NS / MT_NW;   // slots owned by one wave
constexpr int MT_LEAVES = 16;   // leaves of the cross-wave column tree: chunk classes L mod 16
constexpr int MT_BSTR = 73;     // row stride (doubles) of the point / gradient matrices in LDS
constexpr int MT_SP = 28;       // changepoints this kernel handles (fbprophet's default: 25)
constexpr int MT_SOLO_MAX = 3;  // at most this many requests in a round: the waves share each evaluation instead
constexpr int MT_MAXCP = 7;     // changepoint rows one chunk may hold (2 + 2*7 = 16 trend columns)

// Re-laid tables of ONE aligned grid (mfma_layout_kernel).  L: chunk 0..63; g: group of 16 row
// slots of a chunk, NG = ceil(NT/16); slot i of group g is row q = 16 NG - 1 - (16 g + i) of the
// chunk (descending: slot order = chain order), a zero row where q >= rows of the chunk.
struct MfmaTabs {
    const double *XF;           // [64][NG][KF][64]     forward A operands: slot lane%16, column 4kk + lane/16
    const double *XB;           // [64][NG][4][NCB][64] backward A operands: column 16cb + lane%16, slot 4rr + lane/16
    const double *XT;           // [64][NG][4][64]      trend A operands: trend column lane%16, slot 4rr + lane/16
    const double *tq;           // [64][NG][4][4]       scaled time of slot 4rr + k at [k][rr]
    const uint16_t *cq;         // [64][NG][4][4]       trend segment of that row, 0xFFFF = no row
    const int8_t *cpof;         // [64][8]              changepoint whose partial sums are trend column pair i of chunk L
    const double *yq;           // [N][64][NG][4][4]    scaled y, same order as tq
    double *hist;               // [slots][2][MAXH][64] L-BFGS history S, Y of every resident slot
    int *counter;               // work queue head
    const int *overflow;        // != 0: some chunk holds more than MT_MAXCP changepoint rows
    int NG, KF, NCB, rr0;       // rr0: first 4-slot block of group 0 that holds rows
    int run_if_overflow;        // this launch is the fallback (fit_kernel) / the tile kernel
    long long *dbg;             // -DTSF_MFMA_TIMING builds only
};

}  // namespace tsf
