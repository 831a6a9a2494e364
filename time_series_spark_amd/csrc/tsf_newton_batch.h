// tsf_newton_batch.h -- Stan's Newton optimiser for aligned linear/additive panels, several series per
// wave, so that the one part of an iteration that has NO lane parallelism runs with lane = series.
//
// Where the time of newton_quad_kernel goes (profiles/r03_newton, 20 000 x 90, P = 34): 63 % of a fit
// is the implicit-QL rotation chain of the eigen-solver -- about a thousand rotations per
// decomposition, each a dependent scalar sequence (square root, reciprocal, a dozen multiply-adds)
// that every lane of the wave computes identically.  The vector unit is saturated by eleven waves
// doing that 64-fold redundant arithmetic.
//
// Here a wave owns NS slots (series in flight).  One round =
//   for every slot, lane = parameter (what newton_one_quad does, cut in two at the eigen-solver):
//       finish the slot's iteration   V <- Q (reloaded) x recorded rotations, projection, step,
//                                     halving trials, convergence test; a finished series leaves,
//                                     the next one of the queue takes the slot
//       start the next iteration      residual pass + re-centring, finite-difference Hessian,
//                                     A + A^T, Householder tridiagonalisation, Q -> slot record
//   then ONE pass of the QL chain for all slots at once, lane = slot: d, e of every slot side by side
//   in LDS, each lane runs its own sweeps (same operations, same order as ql_chain) and writes its
//   rotations (c, s, column) to the slot's list and its eigenvalues to the slot's record.
// The chain's instructions are shared by NS series instead of one; the rotations are applied to V
// afterwards with lane = row, a pipelined loop without the chain's latency.  Per series the operations
// and their order are those of newton_one_quad / oracle cn_newton: bit-identical.
//
// Slot records live in global memory (L2-resident working set): scalars, theta, gradient, reference
// point and c of the quadratic form, d, e, eigenvalues, Q (PM x PM) and the rotation list.  A list that
// overflows its capacity (LCAP rotations; 2 P^2 + 256 is ~2.5 x the usual count) sends that slot through
// the in-wave chain (ql_chain) instead -- same bits, old speed.
#pragma once
#include "tsf_newton_quad.h"

namespace tsf {

#define NB_MAX_SLOTS 16
#define NB_LONG_IT 256          // iterations after which a series counts as long (the mean is ~130)
enum { NB_EMPTY = 0, NB_NEED_A = 1, NB_WAIT_CHAIN = 2, NB_HAVE_EIG = 3 };
// slot record, in doubles
enum { NB_LP = 0, NB_LASTLP, NB_F0, NB_S0, NB_IT, NB_MI, NB_NEVAL, NB_CNT, NB_SERIES, NB_OVER, NB_NSW,
       NB_TH = 16, NB_G = NB_TH + W, NB_REF = NB_G + W, NB_CVEC = NB_REF + W, NB_D = NB_CVEC + W,
       NB_E = NB_D + W, NB_LAM = NB_E + W, NB_V = NB_LAM + W };

struct NewtonBatchArgs {
    double *rec;            // [blocks * NS][rec_stride]
    int *rot_idx;           // [blocks * NS][LCAP]: sweep headers (first column, rotations recorded), two ints per sweep
    long long rec_stride;   // doubles per slot record: NB_V + PM * PM + 2 * LCAP
    int NS, LCAP;
    int flags;              // dev switches: 1 no hold, 2 Z^T Z of the halving trials from global memory
};

__host__ __device__ constexpr long long nb_rec_doubles(int PM, int LCAP)
{
    return (((long long)NB_V + (long long)PM * PM + 1) & ~1LL) + 2LL * LCAP;        // (the list starts 16-byte aligned)
}
__host__ __device__ constexpr int nb_lcap(int P) { return 2 * P * P + 256; }

template <int KP>
constexpr size_t newton_batch_shared_bytes(int PM, int NTmax)
{
    size_t b = newton_quad_shared_bytes<KP>(PM, NTmax);
    const size_t chain = sizeof(double) * 2 * (size_t)PM * NB_MAX_SLOTS;     // d, e of every slot, [index][slot]
    return b > chain ? b : chain;
}
template <int KP>
constexpr size_t newton_batch_lds_bytes(int PM, int NTmax)
{
    return newton_batch_shared_bytes<KP>(PM, NTmax) + ((sizeof(NewtonQuadLds<KP>) + 15) & ~(size_t)15) + 64;
}

#ifdef TSF_QUAD_TIMING
#define NBT_LAP(k) do { const long long t_ = __builtin_readcyclecounter(); nbt[k] += t_ - nbt0; nbt0 = t_; } while (0)
#else
#define NBT_LAP(k) do { } while (0)
#endif

// The QL chain of ql_chain with lane = slot: d and e (e shifted down by one: e[j] couples j and j + 1,
// e[n-1] = 0) of this lane's slot are dl[j * NB_MAX_SLOTS + lane], el[...].  Rotations go to cs / ri.
// The (l, guard) loop nest of ql_chain is flattened into one loop over sweeps so that lanes whose
// eigenvalues need different numbers of sweeps do not wait for each other at every l; per lane the
// sequence of operations is the same.
__device__ __forceinline__ void ql_chain_slots(int n, double *dl, double *el, double *cs, int *ri, int LCAP,
                                               int &cnt_out, int &nsw_out, bool &over_out)
{
    const int lane = lane_id();
    double *d = dl + lane, *e = el + lane;
    constexpr int ST = NB_MAX_SLOTS;
    int cnt = 0, nsw = 0;
    bool over = false;
    int l = 0, guard = 0;
    while (l < n) {
        // m: first index in [l, n-2] whose off-diagonal element is negligible, else n-1; m == l: this
        // eigenvalue is done
        int m = l;
        for (;;) {
            for (m = l; m < n - 1; ++m) {
                const double dd = __builtin_fabs(d[m * ST]) + __builtin_fabs(d[(m + 1) * ST]);
                if (__builtin_fabs(e[m * ST]) + dd == dd) break;
            }
            if (m != l) break;
            ++l; guard = 0;
            if (l >= n) break;
        }
        if (l >= n) break;
        const double dl_ = d[l * ST], el_ = e[l * ST];
        double g = (d[(l + 1) * ST] - dl_) / (2.0 * el_);
        double r = ql_pythag(g, 1.0);
        g = d[m * ST] - dl_ + el_ / (g + (g >= 0.0 ? __builtin_fabs(r) : -__builtin_fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i = m - 1;
        bool underflow = false;
        // d[i + 1] of a rotation is the d[i] of the one before it (a rotation writes d[i + 1] only): carried;
        // e[i - 1], d[i - 1] of the next rotation are read while this one computes
        double di1 = d[m * ST];
        double ei = e[i * ST], di = d[i * ST];
        const int cnt0 = cnt;
        for (; i >= l; --i) {
            double ein = 0.0, din = 0.0;
            if (i > l) { ein = e[(i - 1) * ST]; din = d[(i - 1) * ST]; }
            double f = s * ei;
            const double b = c * ei;
            r = __builtin_sqrt(__builtin_fma(f, f, g * g));
            e[(i + 1) * ST] = r;
            if (r == 0.0) {
                d[(i + 1) * ST] = di1 - p;
                e[m * ST] = 0.0;
                underflow = true;
                break;
            }
            { const double rinv = 1.0 / r; s = f * rinv; c = g * rinv; }
            g = di1 - p;
            r = (di - g) * s + 2.0 * c * b;
            p = s * r;
            d[(i + 1) * ST] = g + p;
            g = c * r - b;
            if (cnt < LCAP) { cs[2 * cnt] = c; cs[2 * cnt + 1] = s; }
            else over = true;
            ++cnt;
            di1 = di; di = din; ei = ein;
        }
        if (!underflow) {
            d[l * ST] = d[l * ST] - p;
            e[l * ST] = g;
            e[m * ST] = 0.0;
        }
        // the sweep's header: its first column (rotation k of the sweep mixes columns m-1-k and m-k) and
        // how many rotations it recorded (an underflow ends a sweep early)
        if (2 * nsw + 1 < LCAP) { ri[2 * nsw] = m - 1; ri[2 * nsw + 1] = cnt - cnt0; }
        else over = true;
        ++nsw;
        if (++guard == 60) { ++l; guard = 0; }
    }
    cnt_out = cnt;
    nsw_out = nsw;
    over_out = over;
}

// recorded rotations applied to the columns of V (lane = row): what ql_chain does inside its sweeps.
// Within a sweep the columns descend one by one: the shared column is carried in a register and the
// next four are read together ahead of the dependent chain.  Rotations (c, s pairs) and sweep headers are
// fetched 64 at a time, one per lane (coalesced), staged in LDS (st_cs: 128 doubles, st_h: 128 ints) and
// read back as broadcasts; the next 64 are in flight while these are applied.
__device__ __forceinline__ void apply_sweeps(int n, int PM, double *Vm, const double *cs, const int *hdr, int cnt, int nsw,
                                             double *st_cs, int *st_h)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef int i2 __attribute__((ext_vector_type(2)));
    const int lane = lane_id();
    const bool act = lane < n;
    double *vrow = Vm + (act ? lane : 0) * PM;
    const d2 *cs2 = reinterpret_cast<const d2 *>(cs);
    const i2 *h2 = reinterpret_cast<const i2 *>(hdr);
    d2 pc = {0.0, 0.0};
    i2 ph = {0, 0};
    if (lane < cnt) pc = cs2[lane];
    if (lane < nsw) ph = h2[lane];
    wave_sync();
    reinterpret_cast<d2 *>(st_cs)[lane] = pc;
    reinterpret_cast<i2 *>(st_h)[lane] = ph;
    wave_sync();
    int rb = 0, hb = 0;
    if (W + lane < cnt) pc = cs2[W + lane];
    if (W + lane < nsw) ph = h2[W + lane];
    int r = 0, prevcol = -1;
    double vcar = 0.0;
    for (int sw = 0; sw < nsw; ++sw) {
        if (sw >= hb + W) {
            wave_sync();
            reinterpret_cast<i2 *>(st_h)[lane] = ph;
            wave_sync();
            hb += W;
            if (hb + W + lane < nsw) ph = h2[hb + W + lane];
        }
        const int i0 = __builtin_amdgcn_readfirstlane(st_h[2 * (sw - hb)]);
        const int len = __builtin_amdgcn_readfirstlane(st_h[2 * (sw - hb) + 1]);
        if (act) { if (prevcol >= 0) vrow[prevcol] = vcar; vcar = vrow[i0 + 1]; }
        int k = 0;
        while (k < len) {
            if (r >= rb + W) {
                wave_sync();
                reinterpret_cast<d2 *>(st_cs)[lane] = pc;
                wave_sync();
                rb += W;
                if (rb + W + lane < cnt) pc = cs2[rb + W + lane];
            }
            int chunk = len - k;
            if (chunk > rb + W - r) chunk = rb + W - r;
            if (act) {
                const d2 *pcs = reinterpret_cast<const d2 *>(st_cs) + (r - rb);
                double *col = vrow + (i0 - k);          // col[-u]: column of rotation k + u
                int u = 0;
                for (; u + 4 <= chunk; u += 4) {
                    double v4[4];
                    d2 q4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v4[j] = col[-(u + j)]; q4[j] = pcs[u + j]; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        col[-(u + j) + 1] = __builtin_fma(q4[j].y, v4[j], q4[j].x * vcar);
                        vcar = __builtin_fma(q4[j].x, v4[j], -(q4[j].y * vcar));
                    }
                }
                for (; u < chunk; ++u) {
                    const double v0 = col[-u];
                    const d2 q = pcs[u];
                    col[-u + 1] = __builtin_fma(q.y, v0, q.x * vcar);
                    vcar = __builtin_fma(q.x, v0, -(q.y * vcar));
                }
            }
            k += chunk; r += chunk;
        }
        prevcol = i0 - len + 1;
    }
    if (act && prevcol >= 0) vrow[prevcol] = vcar;
    wave_sync();
}

template <int KP>
__global__ __launch_bounds__(64, (KP <= 16 ? TSF_NEWTON_QUAD_WPS : 1)) void newton_batch_kernel(QuadArgs qa, int PM, NewtonBatchArgs nb)
{
    constexpr int PPL = 1;
    extern __shared__ __align__(16) unsigned char smem[];
    const FitArgs &a = qa.f;
    // LDS: [shared region][NewtonQuadLds][slot stages].  The region comes FIRST: the halving trials read a
    // compact copy of Z^T Z from it with 64 lanes per row of PM entries, and what the lanes beyond a row's
    // end read (the next row; after the last row the finite ref / cvec that follow) only ever meets D = 0
    unsigned char *shared = smem;
    const size_t shared_bytes = newton_batch_shared_bytes<KP>(PM, a.NTmax);
    NewtonQuadLds<KP> &lds = *reinterpret_cast<NewtonQuadLds<KP> *>(smem + shared_bytes);
    int *stage_l = reinterpret_cast<int *>(smem + shared_bytes + ((sizeof(NewtonQuadLds<KP>) + 15) & ~(size_t)15));   // [NB_MAX_SLOTS]
    QuadLds<KP, 1> &wl = *reinterpret_cast<QuadLds<KP, 1> *>(shared);      // residual-pass scratch ...
    double *rb = reinterpret_cast<double *>(shared + sizeof(QuadLds<KP, 1>));
    double *Am = reinterpret_cast<double *>(shared);                          // ... the matrix ...
    double *Vm = Am;
    double *chd = reinterpret_cast<double *>(shared);                         // ... and the chain's d, e: same bytes
    double *che = chd + (size_t)PM * NB_MAX_SLOTS;
    const int lane = lane_id();
    const DevSpec *sp = a.sp;
    const int NS = nb.NS;
    const double *Mp = qa.Mg;
    const double epsilon = 1e-3, half_epsilon = 0.5 * epsilon;
    if (lane < NB_MAX_SLOTS) stage_l[lane] = NB_EMPTY;
    wave_sync();
#ifdef TSF_QUAD_TIMING
    long long nbt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long nbt0 = __builtin_readcyclecounter();
    const long long nbt_start = nbt0;
#endif
    // A wave that holds a series far beyond the usual iteration count stops taking new series until its
    // slots have drained: with NS series in flight a slot advances one iteration per NS slot-phases, and
    // the longest series of a call (20 x the mean) would otherwise crawl for as long as the queue lasts
    // and then keep the whole launch waiting.  Other waves take the queue meanwhile.
    bool queue_empty = false, hold = false;
    for (;;) {
        int n_wait = 0, n_busy = 0;
        bool any_long = false;
        for (int s = 0; s < NS; ++s) {
            double *rec = nb.rec + ((size_t)blockIdx.x * NS + s) * nb.rec_stride;
            int *ridx = nb.rot_idx + ((size_t)blockIdx.x * NS + s) * nb.LCAP;
            double *rcs = rec + (((size_t)NB_V + (size_t)PM * PM + 1) & ~(size_t)1);
            int stage = stage_l[s];
            // the slot's series, if it has one
            int64_t n = 0;
            SeriesView sv;
            LaneConst<PPL> lk;
            double th[PPL], g[PPL];
            th[0] = 0.0; g[0] = 0.0;
            double lp = 0.0, lastlp = 0.0, f0 = 0.0, s0 = 0.0;
            int it = 0, mI = 0;
            bool fresh = false;
            int ret = 0;
            bool done = false;
            // the halving trials of an iteration and what follows them (convergence test, next stage), given the step: shared
            // by the two ways an iteration gets its step -- the eigen route (NB_HAVE_EIG, a round later) and the Cholesky
            // shortcut (at once, in the round the Hessian was formed)
            auto halve_and_finish = [&](const double stepv) {
                const int P = sv.P;
                done = false;           // (a slot may finish one series and carry another through an iteration in the same round)
                // The ~30 halving trials of the iteration lie on ONE line through the point the quadratic form was just
                // re-centred at: SSE = s0 + 2 size (c . step) + size^2 (step . Z^T Z step) (cn_newton, round 6) -- ONE mat-vec
                // per iteration, straight from L2 (until round 5 every trial was a mat-vec on a compact LDS copy of Z^T Z made
                // for the purpose: 19 % of a fit), then three scalars and cn_assemble_q's prior sums per trial.
                wave_sync();
                double sw, cs;
                {
                    const double Dl = (lane == 2 || lane >= P) ? 0.0 : stepv;
                    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                    const int P4l = (P + 3) & ~3;       // (rows >= P of Z^T Z are zero and meet D = 0)
                    for (int qq = 0; qq < P4l; qq += 4) {
                        a0 = __builtin_fma(Mp[(size_t)(qq + 0) * W + lane], readlane_f64(Dl, qq + 0), a0);
                        a1 = __builtin_fma(Mp[(size_t)(qq + 1) * W + lane], readlane_f64(Dl, qq + 1), a1);
                        a2 = __builtin_fma(Mp[(size_t)(qq + 2) * W + lane], readlane_f64(Dl, qq + 2), a2);
                        a3 = __builtin_fma(Mp[(size_t)(qq + 3) * W + lane], readlane_f64(Dl, qq + 3), a3);
                    }
                    const double wv = (a0 + a1) + (a2 + a3);
                    sw = bfly_sum(Dl * wv + 0.0);
                    cs = bfly_sum(lds.cvec[lane] * Dl + 0.0);
                }
                // Stan's `while (f1 < f0)` step-halving loop
                double size = 2.0, f1 = -1e100;
                bool moved = true;
                double x[PPL], gx[PPL], fx;
                x[0] = th[0];
                while (f1 < f0) {
                    size *= 0.5;
                    if (size < 1e-50) { moved = false; break; }
                    x[0] = th[0] - size * stepv;
                    sv.n_eval++;
                    const double q2l = (size * size) * sw;
                    const double cdl = -(size * cs);
                    const double ssel = __builtin_fma(-2.0, cdl, s0) + q2l;
                    const double zero[PPL] = {0.0};
                    const bool bad = assemble_q<PPL>(sv, lk, x, ssel, zero, fx, gx);
                    f1 = bad ? -1e100 : -fx;
                }
                NBT_LAP(4);
                ++it;
                if (moved) { th[0] = x[0]; lp = f1; }
                else lp = f0;
                if (mI > 0 && __builtin_fabs(lp - lastlp) < 1e-8) { ret = TSF_ST_NEWTON_CONVERGED; done = true; }
                else if (++mI >= a.opt.max_iter) { ret = TSF_ST_MAXIT; done = true; }
                if (done) {
                    store_theta<PPL>(a, sv, n, th, a.theta);
                    if (lane == 0) { a.status[n] = ret; a.n_iter[n] = it; a.n_eval[n] = sv.n_eval; a.fval[n] = -lp; }
                    stage = NB_EMPTY;
                } else {
                    stage = NB_NEED_A;
                }
            };
            if (stage == NB_HAVE_EIG) {
                // ---------------- finish the iteration: eigenvectors, projection, step, halving ----------------
                n = (int64_t)rec[NB_SERIES];
                make_view_q<KP, PPL>(a, n, sv);
                lane_consts<PPL>(sp, sv, lds.lanec, lk);
                const int P = sv.P;
                lp = rec[NB_LP]; lastlp = rec[NB_LASTLP]; f0 = rec[NB_F0]; s0 = rec[NB_S0];
                it = (int)rec[NB_IT]; mI = (int)rec[NB_MI]; sv.n_eval = (int)rec[NB_NEVAL];
                const int cnt = (int)rec[NB_CNT];
                const bool over = rec[NB_OVER] != 0.0;
                th[0] = rec[NB_TH + lane]; g[0] = rec[NB_G + lane];
                lds.ref[lane] = rec[NB_REF + lane];
                lds.cvec[lane] = rec[NB_CVEC + lane];
                {   // Q back into LDS: eight loads in flight at a time (the records do not fit the L2)
                    const int nV = PM * PM;
                    int i = lane;
                    for (; i + 7 * W < nV; i += 8 * W) {
                        double v8[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v8[u] = rec[NB_V + i + u * W];
#pragma unroll
                        for (int u = 0; u < 8; ++u) Vm[i + u * W] = v8[u];
                    }
                    for (; i < nV; i += W) Vm[i] = rec[NB_V + i];
                }
                wave_sync();
                NBT_LAP(7);
                double lam;
                if (over) {         // the rotation list did not hold this decomposition: the in-wave chain
                    lds.ql.d[lane] = rec[NB_D + lane];
                    lds.ql.e[lane] = rec[NB_E + lane];
                    wave_sync();
                    lam = ql_chain(P, PM, Vm, lds.ql);
                } else {
                    apply_sweeps(P, PM, Vm, rcs, ridx, cnt, (int)rec[NB_NSW], lds.ql.d, reinterpret_cast<int *>(lds.ql.hh));
                    lam = rec[NB_LAM + lane];
                    wave_sync();
                }
                NBT_LAP(2);
                double pa = 0.0;
                for (int i = 0; i < P; ++i) {
                    const double gi = -readlane_f64(g[0], i);
                    const double vij = (lane < P) ? Vm[i * PM + lane] : 0.0;
                    pa = __builtin_fma(vij, gi, pa);
                }
                const double proj = (lane < P) ? -pa / __builtin_fabs(lam) : 0.0;
                double sa = 0.0;
                for (int j = 0; j < P; ++j) {
                    const double pj = readlane_f64(proj, j);
                    const double vij = (lane < P) ? Vm[lane * PM + j] : 0.0;
                    sa = __builtin_fma(vij, pj, sa);
                }
                const double stepv = (lane < P) ? sa : 0.0;
                NBT_LAP(3);
                halve_and_finish(stepv);
            }
            for (;;) {
            // ---------------- an empty slot takes the next series of the queue ----------------
            while (stage == NB_EMPTY && !queue_empty && !hold) {
                int n32 = atomicAdd(qa.counter, lane == 0 ? 1 : 0);     // every lane takes part (see fit_quad_kernel)
                n32 = __builtin_amdgcn_readfirstlane(n32);
                if (n32 >= a.N) { queue_empty = true; break; }
                n = n32;
                make_view_q<KP, PPL>(a, n, sv);
                const SeriesTab st = a.stab[n];
                if (lane == 0) {
                    a.y_scale[n] = st.y_scale;
                    if (!a.aligned || n == 0) a.grid_out[a.aligned ? 0 : n] = a.gtab[grid_index(a, n)].info;
                }
                th[0] = (lane == 0) ? st.k0 : (lane == 1 ? st.m0 : 0.0);
                if (st.status0 != 0) {
                    if (st.status0 == TSF_ST_CONSTANT && lane == 2) th[0] = -20.72326583694641;
                    store_theta<PPL>(a, sv, n, th, a.theta);
                    if (lane == 0) { a.status[n] = st.status0; a.n_iter[n] = 0; a.n_eval[n] = 0; a.fval[n] = 0.0; }
                    continue;
                }
                lane_consts<PPL>(sp, sv, lds.lanec, lk);
                lp = 0.0; lastlp = 0.0; it = 0; mI = 0; sv.n_eval = 0;
                fresh = true;
                stage = NB_NEED_A;
            }
            if (stage != NB_NEED_A) break;
            {
                // ---------------- start an iteration: everything up to the tridiagonal matrix ----------------
                const int P = sv.P;
                double x[PPL], gx[PPL], fx, sse_e, ztr_e[PPL];
                x[0] = th[0];
                wl.th[W + lane] = 0.0;      // theta of a pass is zero beyond P (the region held the matrix)
                sv.n_eval++;
                const bool bad = resid_eval_q<KP, PPL>(sv, wl, lk, rb, x, fx, gx, sse_e, ztr_e);
                NBT_LAP(5);
                bool failed = false;
                if (fresh) {
                    // newton_one_quad evaluates the initial point twice (log_prob, then the first
                    // grad_hess_log_prob): same point, same bits, counted twice
                    if (bad) { ret = TSF_ST_INIT_NONFINITE; lp = -fx; failed = true; }
                    else { lp = -fx; sv.n_eval++; }
                } else if (bad) {
                    ret = TSF_ST_NEWTON_FAIL; failed = true;
                }
                fresh = false;          // (the Cholesky shortcut may start the next iteration of this series in the same round)
                if (!failed) {
                    lds.ref[lane] = (lane == 2) ? 0.0 : x[0];
                    lds.cvec[lane] = (lane == 2) ? 0.0 : ztr_e[0];
                    s0 = sse_e;
                    wave_sync();
                    lastlp = lp;
                    f0 = -fx;
                    g[0] = gx[0];
                    // finite-difference Hessian: newton_one_quad's collapsed form, verbatim
                    const double thl = th[0], refl = lds.ref[lane], cvl = lds.cvec[lane];
                    const double lcl = lk.lc[lane], scl = lk.sc[lane];
                    const double Td = (double)sv.T;
                    const double ls0 = readlane_f64(thl, 2);
                    const double sigma0 = dm_exp(ls0);
                    const double s2_0 = sigma0 * sigma0;
                    const double inv_s2_0 = 1.0 / s2_0;
                    bool fd_bad = false;
                    double mnext = Mp[lane];
                    for (int d = 0; d < P; ++d) {
                        const double mcol = (d == 2) ? 0.0 : mnext;
                        if (d + 1 < P) mnext = Mp[(size_t)(d + 1) * W + lane];
                        const double cvd = readlane_f64(cvl, d);
                        double accd = 0.0;
#pragma unroll
                        for (int pi = 0; pi < 4; ++pi) {
                            const double pert = (pi == 0) ? -2 * epsilon : (pi == 1 ? -1 * epsilon : (pi == 2 ? epsilon : 2 * epsilon));
                            const double coef = (pi == 0) ? 1.0 / 12.0 : (pi == 1 ? -2.0 / 3.0 : (pi == 2 ? 2.0 / 3.0 : -1.0 / 12.0));
                            const double thp = (lane == d) ? thl + pert : thl;
                            const double Dd = (d == 2) ? 0.0 : readlane_f64(thp - refl, d);
                            const double v = mcol * Dd + 0.0;
                            const double q2d = Dd * readlane_f64(v, d) + 0.0;
                            const double cd = cvd * Dd + 0.0;
                            const double sse = __builtin_fma(-2.0, cd, s0) + q2d;
                            const double ztr = cvl - v;
                            double s2 = s2_0, inv_s2 = inv_s2_0;
                            if (d == 2) {
                                const double sigma = dm_exp(ls0 + pert);
                                s2 = sigma * sigma;
                                inv_s2 = 1.0 / s2;
                            }
                            const double nis = -inv_s2;
                            const double sgn = (double)((thp > 0.0) - (thp < 0.0));
                            double gv = __builtin_fma(thp, lcl, nis * ztr) + sgn * scl;
                            if (lane == 2) gv = (Td - sse * inv_s2) + 4.0 * s2;
                            if (lane >= P) gv = 0.0;
                            fd_bad = fd_bad || !finite_f64(gv);
                            accd = __builtin_fma(half_epsilon * coef, -gv, accd);
                        }
                        if (lane < P) Am[d * PM + lane] = accd;
                    }
                    sv.n_eval += 4 * P;
                    if (__any(fd_bad)) { ret = TSF_ST_NEWTON_FAIL; failed = true; }
                }
                if (failed) {
                    store_theta<PPL>(a, sv, n, th, a.theta);
                    if (lane == 0) { a.status[n] = ret; a.n_iter[n] = it; a.n_eval[n] = sv.n_eval; a.fval[n] = -lp; }
                    stage = NB_EMPTY;       // the slot takes the next series right away
                    continue;
                } else {
                    wave_sync();
                    NBT_LAP(6);
                    for (int r = 0; r < P; ++r) {       // H = A + A^T
                        double u = 0.0, v = 0.0;
                        const bool mine = lane < P && r <= lane;
                        if (mine) { u = Am[r * PM + lane]; v = Am[lane * PM + r]; }
                        wave_sync();
                        if (mine) { const double h = u + v; Am[r * PM + lane] = h; Am[lane * PM + r] = h; }
                        wave_sync();
                    }
                    {   // H negative definite (a third of the iterations): the step by Cholesky, the iteration finished here
                        // and now -- no tridiagonalisation, no record, no round of the chain (chol_neg_solve, round 6)
                        double st_c;
                        if (chol_neg_solve(P, PM, Am, g[0], st_c)) {
                            NBT_LAP(0);
                            halve_and_finish(st_c);
                            continue;
                        }
                    }
                    ql_tridiag_q(P, PM, Am, Vm, lds.ql);
                    NBT_LAP(0);
                    // the slot's record: Q, d, e, the vectors and scalars of the iteration
                    for (int i = lane; i < PM * PM; i += W) rec[NB_V + i] = Vm[i];
                    rec[NB_D + lane] = lds.ql.d[lane];
                    rec[NB_E + lane] = lds.ql.e[lane];
                    rec[NB_TH + lane] = th[0];
                    rec[NB_G + lane] = g[0];
                    rec[NB_REF + lane] = lds.ref[lane];
                    rec[NB_CVEC + lane] = lds.cvec[lane];
                    if (lane == 0) {
                        rec[NB_LP] = lp; rec[NB_LASTLP] = lastlp; rec[NB_F0] = f0; rec[NB_S0] = s0;
                        rec[NB_IT] = (double)it; rec[NB_MI] = (double)mI; rec[NB_NEVAL] = (double)sv.n_eval;
                        rec[NB_SERIES] = (double)n;
                    }
                    stage = NB_WAIT_CHAIN;
                    ++n_wait;
                    wave_sync();
                    NBT_LAP(7);
                    break;
                }
            }
            }
            if (lane == 0) stage_l[s] = stage;
            if (stage != NB_EMPTY) { ++n_busy; if (it >= NB_LONG_IT) any_long = true; }
            wave_sync();
        }
        if (any_long && !(nb.flags & 1)) hold = true;
        if (n_busy == 0) {
            if (queue_empty || !hold) break;    // (not holding: the slots found the queue empty)
            hold = false;                       // drained: take series again
            continue;
        }
        if (n_wait == 0) break;                 // (unreachable: a busy slot waits for its chain)
        // ---------------- the QL chains of every waiting slot, lane = slot ----------------
        __threadfence();        // the records written above are read below by other lanes
        const int P = 3 + a.gtab[0].S_fit + sp->K;      // one grid: every series of the panel has the same P
        {
            const bool mine = lane < NS && stage_l[lane < NB_MAX_SLOTS ? lane : 0] == NB_WAIT_CHAIN;
            double *rec = nb.rec + ((size_t)blockIdx.x * NS + (mine ? lane : 0)) * nb.rec_stride;
            if (mine) {
                for (int j = 0; j < P; ++j) {
                    chd[j * NB_MAX_SLOTS + lane] = rec[NB_D + j];
                    che[j * NB_MAX_SLOTS + lane] = (j + 1 < P) ? rec[NB_E + j + 1] : 0.0;
                }
                int cnt = 0, nsw = 0;
                bool over = false;
                ql_chain_slots(P, chd, che, rec + (((size_t)NB_V + (size_t)PM * PM + 1) & ~(size_t)1),
                               nb.rot_idx + ((size_t)blockIdx.x * NS + lane) * nb.LCAP, nb.LCAP, cnt, nsw, over);
                rec[NB_NSW] = (double)(over ? 0 : nsw);
                for (int j = 0; j < P; ++j) rec[NB_LAM + j] = chd[j * NB_MAX_SLOTS + lane];
                for (int j = P; j < W; ++j) rec[NB_LAM + j] = 0.0;
                rec[NB_CNT] = (double)(over ? 0 : cnt);
                rec[NB_OVER] = over ? 1.0 : 0.0;
                stage_l[lane] = NB_HAVE_EIG;
            }
        }
        __threadfence();
        wave_sync();
        NBT_LAP(1);
    }
#ifdef TSF_QUAD_TIMING
    if (qa.dbg && lane == 0) {
        for (int k = 0; k < 8; ++k) qa.dbg[(size_t)blockIdx.x * 9 + k] = nbt[k];
        qa.dbg[(size_t)blockIdx.x * 9 + 8] = __builtin_readcyclecounter() - nbt_start;
    }
#endif
}

}  // namespace tsf
