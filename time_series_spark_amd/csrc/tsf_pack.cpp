// Host-side panel packing behind tsf_pack_rows / tsf_pack_fetch / tsf_pack_free (include/tsf.h).
//
// What it replaces in the reference: the row movement of
//   df.groupby('series_id', 'dim_id').apply(udf)      /root/reference/src/jobs/prophet_modeler.py:139-141
// (Spark shuffle + one Arrow -> pandas frame per group) and the per-group host steps fbprophet
// performs before Stan sees the data: history = df[df['y'].notnull()], sort by ds
// (UPSTREAM-RECALL fbprophet 0.5 Prophet.fit / setup_dataframe).  Here the whole
// long table is regrouped once into contiguous per-series runs -- the SoA layout
// tsf_fit_ragged / tsf_fit_aligned take -- plus the three per-series statistics the Python layer
// needs for fbprophet's 'auto' seasonality rules and the reference's cap = max(y) * multiplier
// (prophet_modeler.py:59-60).  No device work; plain C++ threads.
//
// Order contract (what tests/ compare against numpy's lexsort): series ascending by
// (series_id, dim_id); within a series ascending ds, ties in input order (stable); rows whose
// y is NaN dropped.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <new>
#include <numeric>
#include <thread>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "../../include/tsf.h"
#include "tsf_pool.h"

namespace {

struct Key {
    int64_t sid, did;
    bool operator==(const Key &o) const { return sid == o.sid && did == o.did; }
    bool operator<(const Key &o) const { return sid < o.sid || (sid == o.sid && did < o.did); }
};

inline uint64_t mix(uint64_t a, uint64_t b) {
    uint64_t h = a * 0x9E3779B97F4A7C15ull ^ (b + 0xD1B54A32D192ED03ull + (a << 6) + (a >> 2));
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return h;
}

// open addressing (series key -> dense id in first-appearance order)
struct KeyMap {
    std::vector<int64_t> slot;      // id or -1
    std::vector<Key> keys;          // by id
    uint64_t mask = 0;

    void init(size_t cap_pow2) {
        slot.assign(cap_pow2, -1);
        mask = cap_pow2 - 1;
    }
    void grow() {
        std::vector<int64_t> ns(slot.size() * 2, -1);
        uint64_t nm = ns.size() - 1;
        for (size_t id = 0; id < keys.size(); ++id) {
            uint64_t h = mix((uint64_t)keys[id].sid, (uint64_t)keys[id].did) & nm;
            while (ns[h] >= 0) h = (h + 1) & nm;
            ns[h] = (int64_t)id;
        }
        slot.swap(ns);
        mask = nm;
    }
    int64_t get(const Key &k) {
        uint64_t h = mix((uint64_t)k.sid, (uint64_t)k.did) & mask;
        for (;;) {
            int64_t id = slot[h];
            if (id < 0) break;
            if (keys[(size_t)id] == k) return id;
            h = (h + 1) & mask;
        }
        if ((keys.size() + 1) * 2 > slot.size()) {
            grow();
            h = mix((uint64_t)k.sid, (uint64_t)k.did) & mask;
            while (slot[h] >= 0) h = (h + 1) & mask;
        }
        slot[h] = (int64_t)keys.size();
        keys.push_back(k);
        return (int64_t)keys.size() - 1;
    }
};

// f(begin, end, piece_index) over static contiguous pieces, on the library's pool (tsf_pool.h).  An exception inside a
// piece (allocation failure) comes back to the caller as std::bad_alloc instead of terminating the process.
using tsfpool::parallel_for;

// one packed row; stable scatter + stable sort keep ties in input order, so the input row
// number itself is not needed
struct Pair {
    int64_t ds;
    double y;
};

}  // namespace

struct tsf_pack {
    int64_t n_in = 0, n_rows = 0, n_series = 0;
    int32_t identity = 0;
    int n_threads = 1;
    const int64_t *ds = nullptr;       // caller's arrays; must stay alive until fetch
    const void *y = nullptr;           // of y_dtype (TSF_Y_*)
    int32_t y_dtype = TSF_Y_F64;
    std::vector<Key> keys;             // [n_series], ascending
    std::vector<int64_t> offsets;      // [n_series + 1]
    std::vector<Pair> pairs;           // [n_rows] (ds, y) of each packed row (empty if identity)
    // what tsf_pack_fetch's pass over the packed rows sees on the way (tsf_pack_flags): every series on the first
    // series' timestamp vector; an infinite y; every y an integer that fits int32 (the reference's schema)
    int32_t aligned = 0, has_inf = 0, integral = 0, has_nat = 0, fetched = 0;
};

namespace {

// KT: the key columns' type (int32 as the reference's schema has them, prophet_modeler.py:12-17, or int64); YT: y's
// (int32 -- the reference's quantity --, float or double).  A table that already is in packed order is used in place,
// in its own types (round 6: the DataFrame boundary converted three 7.3 M-row columns to int64 / float64 before the packer
// saw them, most of model_panel's time).
template <class KT, class YT>
int pack_rows_impl(int64_t n, const KT *series_id, const KT *dim_id, const int64_t *ds,
                   const YT *y, int32_t y_dtype, int32_t n_threads, tsf_pack **out, int64_t *n_rows,
                   int64_t *n_series, int32_t *identity) {
    *out = nullptr;
    tsf_pack *p = nullptr;
    try {
        p = new tsf_pack();
        int hw = (int)std::thread::hardware_concurrency();
        if (hw < 1) hw = 1;
        p->n_threads = n_threads > 0 ? n_threads : std::min(hw, 32);
        p->n_in = n;
        p->ds = ds;
        p->y = y;
        p->y_dtype = y_dtype;
        const int nt = p->n_threads;
        const bool timing = std::getenv("TSF_PACK_TIMING") != nullptr;
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!timing) return;
            auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "tsf_pack_rows %-10s %8.3f ms\n", what,
                         std::chrono::duration<double, std::milli>(now - t_last).count());
            t_last = now;
        };

        // ---- pass 0: is the table already packed? (grouped, keys ascending, ds ascending, no NaN)
        std::atomic<int> ordered(1);
        parallel_for(n, nt, [&](int64_t a, int64_t b, int) {
            bool ok = true;
            for (int64_t i = a; i < b && ok; ++i) {
                if (std::isnan((double)y[i])) ok = false;
                if (i == 0) continue;
                Key k0{(int64_t)series_id[i - 1], (int64_t)dim_id[i - 1]}, k1{(int64_t)series_id[i], (int64_t)dim_id[i]};
                if (k1 < k0) ok = false;
                else if (k1 == k0 && ds[i] < ds[i - 1]) ok = false;
            }
            if (!ok) ordered.store(0);
        });
        lap("check");
        if (ordered.load()) {
            p->identity = 1;
            p->n_rows = n;
            // run starts, found by the pool (round 6: this loop ran on ONE thread over both key columns -- 117 MB for
            // 10 000 x 730 rows, 15 of the packer's 21 ms on the GPU box), appended in chunk order
            std::vector<std::vector<int64_t>> starts((size_t)nt);
            parallel_for(n, nt, [&](int64_t a, int64_t b, int t) {
                std::vector<int64_t> &s = starts[(size_t)t];
                for (int64_t i = a; i < b; ++i)
                    if (i == 0 || series_id[i] != series_id[i - 1] || dim_id[i] != dim_id[i - 1]) s.push_back(i);
            });
            size_t total_runs = 0;
            for (auto &s : starts) total_runs += s.size();
            p->keys.reserve(total_runs);
            p->offsets.reserve(total_runs + 1);
            for (auto &s : starts)
                for (int64_t i : s) {
                    p->keys.push_back(Key{(int64_t)series_id[i], (int64_t)dim_id[i]});
                    p->offsets.push_back(i);
                }
            p->offsets.push_back(n);
            p->n_series = (int64_t)p->keys.size();
        } else {
            // ---- pass 1 (parallel, one contiguous chunk of rows per thread): chunk-local dense
            // series ids in first-appearance order (runs of one key reuse the last id), NaN rows
            // marked -1, rows per local id counted
            const int64_t per = (n + nt - 1) / nt;
            const int nchunk = (int)((n + per - 1) / per);
            struct Chunk {
                KeyMap map;
                std::vector<int64_t> count;      // by local id
                std::vector<int64_t> start;      // by local id: next write position
                std::vector<int32_t> lid32;      // per row of the chunk (per < 2^31)
                std::vector<int64_t> lid64;
            };
            std::vector<Chunk> ch((size_t)nchunk);
            const bool small = per < (int64_t)0x7fffffff;
            parallel_for(n, nt, [&](int64_t a, int64_t b, int t) {
                Chunk &c = ch[(size_t)t];
                c.map.init(1 << 10);
                if (small) c.lid32.resize((size_t)(b - a));
                else c.lid64.resize((size_t)(b - a));
                Key last{0, 0};
                int64_t last_id = -1;
                for (int64_t i = a; i < b; ++i) {
                    Key k{(int64_t)series_id[i], (int64_t)dim_id[i]};
                    if (last_id < 0 || !(k == last)) {
                        last_id = c.map.get(k);
                        last = k;
                        if ((size_t)last_id >= c.count.size()) c.count.resize((size_t)last_id + 1, 0);
                    }
                    int64_t v = -1;
                    if (!std::isnan((double)y[i])) {
                        v = last_id;
                        ++c.count[(size_t)last_id];
                    }
                    if (small) c.lid32[(size_t)(i - a)] = (int32_t)v;
                    else c.lid64[(size_t)(i - a)] = v;
                }
            });
            lap("local ids");
            // ---- merge: global ids, totals, ascending-key order, run offsets
            KeyMap map;
            map.init(1 << 12);
            std::vector<int64_t> total;
            std::vector<std::vector<int64_t>> trans((size_t)nchunk);
            for (int t = 0; t < nchunk; ++t) {
                Chunk &c = ch[(size_t)t];
                trans[(size_t)t].resize(c.map.keys.size());
                for (size_t l = 0; l < c.map.keys.size(); ++l) {
                    int64_t g = map.get(c.map.keys[l]);
                    if ((size_t)g >= total.size()) total.resize((size_t)g + 1, 0);
                    total[(size_t)g] += c.count[l];
                    trans[(size_t)t][l] = g;
                }
            }
            const size_t G0 = map.keys.size();
            std::vector<int64_t> by_key(G0);
            std::iota(by_key.begin(), by_key.end(), (int64_t)0);
            std::sort(by_key.begin(), by_key.end(),
                      [&](int64_t a, int64_t b) { return map.keys[(size_t)a] < map.keys[(size_t)b]; });
            std::vector<int64_t> cursor(G0, -1);
            int64_t pos = 0;
            for (int64_t g : by_key) {
                if (total[(size_t)g] == 0) continue;     // every row NaN: the series disappears
                p->keys.push_back(map.keys[(size_t)g]);
                p->offsets.push_back(pos);
                cursor[(size_t)g] = pos;
                pos += total[(size_t)g];
            }
            p->offsets.push_back(pos);
            p->n_rows = pos;
            p->n_series = (int64_t)p->keys.size();
            // where each chunk writes inside each run: chunks in row order => stable
            for (int t = 0; t < nchunk; ++t) {
                Chunk &c = ch[(size_t)t];
                c.start.resize(c.count.size());
                for (size_t l = 0; l < c.count.size(); ++l) {
                    int64_t g = trans[(size_t)t][l];
                    c.start[l] = cursor[(size_t)g];
                    cursor[(size_t)g] += c.count[l];
                }
            }
            lap("merge");
            // ---- pass 2 (parallel): stable scatter of (ds, y) into the runs
            p->pairs.resize((size_t)pos);
            parallel_for(n, nt, [&](int64_t a, int64_t b, int t) {
                Chunk &c = ch[(size_t)t];
                for (int64_t i = a; i < b; ++i) {
                    int64_t l = small ? (int64_t)c.lid32[(size_t)(i - a)] : c.lid64[(size_t)(i - a)];
                    if (l >= 0) p->pairs[(size_t)c.start[(size_t)l]++] = Pair{ds[i], (double)y[i]};
                }
            });
            std::vector<Chunk>().swap(ch);
            lap("scatter");
            // ---- pass 3 (parallel over series): stable sort by ds where a run is not ascending
            const int64_t NS = p->n_series;
            std::atomic<int64_t> next(0);
            auto worker = [&]() {
                for (;;) {
                    int64_t s0 = next.fetch_add(64);
                    if (s0 >= NS) break;
                    int64_t s1 = std::min<int64_t>(NS, s0 + 64);
                    for (int64_t s = s0; s < s1; ++s) {
                        Pair *a = p->pairs.data() + p->offsets[(size_t)s];
                        Pair *b = p->pairs.data() + p->offsets[(size_t)s + 1];
                        bool asc = true;
                        for (Pair *q = a + 1; q < b; ++q)
                            if (q->ds < (q - 1)->ds) { asc = false; break; }
                        if (!asc)
                            std::stable_sort(a, b, [](const Pair &u, const Pair &v) { return u.ds < v.ds; });
                    }
                }
            };
            if (nt <= 1 || NS < 128) worker();
            else tsfpool::Pool::get().run(nt, [&](int) { worker(); });
            lap("sort");
        }
    } catch (const std::bad_alloc &) {
        delete p;
        return -2;
    } catch (...) {
        delete p;
        return -3;
    }
    *out = p;
    if (n_rows) *n_rows = p->n_rows;
    if (n_series) *n_series = p->n_series;
    if (identity) *identity = p->identity;
    return 0;
}

}  // namespace

extern "C" {

int tsf_pack_rows_typed(int64_t n, const void *series_id, const void *dim_id, int32_t key_bytes, const int64_t *ds,
                        const void *y, int32_t y_dtype, int32_t n_threads, tsf_pack **out, int64_t *n_rows,
                        int64_t *n_series, int32_t *identity) {
    if (!out || n < 0 || (n > 0 && (!series_id || !dim_id || !ds || !y))) return -1;
    if ((key_bytes != 4 && key_bytes != 8) || y_dtype < TSF_Y_F64 || y_dtype > TSF_Y_I32) return -1;
#define TSF_PACK_GO(KT, YT) return pack_rows_impl<KT, YT>(n, (const KT *)series_id, (const KT *)dim_id, ds, (const YT *)y, y_dtype, \
                                                          n_threads, out, n_rows, n_series, identity)
    if (key_bytes == 4) {
        if (y_dtype == TSF_Y_F64) TSF_PACK_GO(int32_t, double);
        if (y_dtype == TSF_Y_F32) TSF_PACK_GO(int32_t, float);
        TSF_PACK_GO(int32_t, int32_t);
    }
    if (y_dtype == TSF_Y_F64) TSF_PACK_GO(int64_t, double);
    if (y_dtype == TSF_Y_F32) TSF_PACK_GO(int64_t, float);
    TSF_PACK_GO(int64_t, int32_t);
#undef TSF_PACK_GO
}

int tsf_pack_rows(int64_t n, const int64_t *series_id, const int64_t *dim_id, const int64_t *ds,
                  const double *y, int32_t n_threads, tsf_pack **out, int64_t *n_rows,
                  int64_t *n_series, int32_t *identity) {
    return tsf_pack_rows_typed(n, series_id, dim_id, 8, ds, y, TSF_Y_F64, n_threads, out, n_rows, n_series, identity);
}

int tsf_pack_fetch(tsf_pack *p, int64_t *key_series_id, int64_t *key_dim_id, int64_t *offsets,
                   int64_t *ds_out, double *y_out, int64_t *span, int64_t *min_dt, double *y_max) {
    if (!p) return -1;
    const int64_t NS = p->n_series;
    const int64_t *ds = p->ds;
    const void *yv_ = p->y;
    const int32_t ydt = p->y_dtype;
    auto y_at = [yv_, ydt](int64_t r) -> double {
        return ydt == TSF_Y_F64 ? ((const double *)yv_)[r] : (ydt == TSF_Y_F32 ? (double)((const float *)yv_)[r] : (double)((const int32_t *)yv_)[r]);
    };
    const bool ident = p->identity != 0;
    const Pair *pairs = ident ? nullptr : p->pairs.data();
    if (offsets) std::memcpy(offsets, p->offsets.data(), sizeof(int64_t) * (size_t)(NS + 1));
    for (int64_t s = 0; s < NS; ++s) {
        if (key_series_id) key_series_id[s] = p->keys[(size_t)s].sid;
        if (key_dim_id) key_dim_id[s] = p->keys[(size_t)s].did;
    }
    // gather + statistics, series-parallel; the same pass notes what tsf_pack_flags reports
    std::atomic<int64_t> next(0);
    std::atomic<int> not_aligned(0), any_inf(0), not_integral(0), any_nat(0);
    const int64_t len0 = NS > 0 ? p->offsets[1] - p->offsets[0] : 0;
    const int64_t a0 = NS > 0 ? p->offsets[0] : 0;
    auto worker = [&]() {
        for (;;) {
            int64_t s0 = next.fetch_add(64);
            if (s0 >= NS) break;
            int64_t s1 = std::min<int64_t>(NS, s0 + 64);
            for (int64_t s = s0; s < s1; ++s) {
                int64_t a = p->offsets[(size_t)s], b = p->offsets[(size_t)s + 1];
                int64_t first = 0, prev = 0, md = -1;
                double ym = -std::numeric_limits<double>::infinity();
                bool same = (b - a == len0), fin = true, whole = true;
                for (int64_t r = a; r < b; ++r) {
                    int64_t d = ident ? ds[r] : pairs[r].ds;
                    double v = ident ? y_at(r) : pairs[r].y;
                    if (ds_out && !(ident && ds_out == ds)) ds_out[r] = d;
                    if (y_out && !(ident && (const void *)y_out == yv_)) y_out[r] = v;
                    if (same && d != (ident ? ds[a0 + (r - a)] : pairs[a0 + (r - a)].ds)) same = false;
                    if (std::isinf(v)) fin = false;
                    if (d == std::numeric_limits<int64_t>::min()) any_nat.store(1, std::memory_order_relaxed);     // pandas NaT
                    if (!(v >= -2147483648.0 && v <= 2147483647.0 && v == (double)(int32_t)v)) whole = false;
                    if (r == a) first = d;
                    else {
                        int64_t dt = d - prev;
                        if (dt > 0 && (md < 0 || dt < md)) md = dt;
                    }
                    prev = d;
                    if (v > ym) ym = v;
                }
                if (!same) not_aligned.store(1, std::memory_order_relaxed);
                if (!fin) any_inf.store(1, std::memory_order_relaxed);
                if (!whole) not_integral.store(1, std::memory_order_relaxed);
                if (span) span[s] = prev - first;
                if (min_dt) min_dt[s] = md;
                if (y_max) y_max[s] = ym;
            }
        }
    };
    try {
        if (p->n_threads <= 1 || NS < 128) worker();
        else tsfpool::Pool::get().run(p->n_threads, [&](int) { worker(); });
    } catch (...) {
        return -2;
    }
    p->aligned = (NS > 0 && len0 > 0 && !not_aligned.load()) ? 1 : 0;
    p->has_inf = any_inf.load();
    p->has_nat = any_nat.load();
    p->integral = (NS > 0 && !not_integral.load()) ? 1 : 0;
    p->fetched = 1;
    return 0;
}

int tsf_pack_flags(const tsf_pack *p, int32_t *aligned, int32_t *has_inf, int32_t *integral, int32_t *has_nat) {
    if (!p || !p->fetched) return -1;
    if (aligned) *aligned = p->aligned;
    if (has_inf) *has_inf = p->has_inf;
    if (integral) *integral = p->integral;
    if (has_nat) *has_nat = p->has_nat;
    return 0;
}

void tsf_pack_free(tsf_pack *p) { delete p; }

// ---- model blobs ------------------------------------------------------------------------------
// The `model` column of the fit UDF's output (/root/reference/src/jobs/prophet_modeler.py:72-75: pickle.dumps(model) per
// series).  The replacement blob (time_series_spark_amd/panel.py, version 2) is a prefix shared by the batch -- magic,
// version, the constructor arguments as JSON -- and one fixed little-endian record per series:
//   f64 y_scale | i64 start_ns, t_scale_ns, last_ds_ns | i32 T, S, i1, NT, status, n_iter, n_theta, n_tchange |
//   f64 theta[n_theta] | f64 t_change[n_tchange]
// Written here for the whole batch into ONE buffer, blob n at n * stride: the buffer is the data buffer of an Arrow
// binary column as it stands (offsets n * stride), so the model parquet is written without a Python object per series.
int tsf_model_blobs(int64_t N, const void *prefix, int32_t prefix_len, int32_t n_theta, const double *theta,
                    const double *y_scale, const tsf_grid_info *grid, int32_t n_grids, const int64_t *last_ds,
                    const int32_t *status, const int32_t *n_iter, int32_t n_tchange, void *out, int32_t n_threads) {
    if (N < 0 || prefix_len < 0 || n_theta < 0 || n_tchange < 0 || n_tchange > TSF_MAX_S + 4 ||
        (N > 0 && (!theta || !y_scale || !grid || !last_ds || !status || !n_iter || !out || (prefix_len > 0 && !prefix))) ||
        !(n_grids == 1 || (int64_t)n_grids == N))
        return -1;
    const size_t stride = (size_t)prefix_len + 64 + 8 * ((size_t)n_theta + (size_t)n_tchange);
    int hw = (int)std::thread::hardware_concurrency();
    if (hw < 1) hw = 1;
    int nt = n_threads > 0 ? n_threads : std::min(hw, 16);
    if (N < 2048) nt = 1;
    try {
        parallel_for(N, nt, [&](int64_t a, int64_t b, int) {
            for (int64_t n = a; n < b; ++n) {
                char *q = (char *)out + (size_t)n * stride;
                if (prefix_len) std::memcpy(q, prefix, (size_t)prefix_len);
                q += prefix_len;
                const tsf_grid_info &g = grid[n_grids == 1 ? 0 : n];
                const int64_t i8[3] = {g.start_ns, g.t_scale_ns, last_ds[n]};
                const int32_t i4[8] = {g.T, g.S, g.i1, g.NT, status[n], n_iter[n], n_theta, n_tchange};
                std::memcpy(q, &y_scale[n], 8);
                std::memcpy(q + 8, i8, 24);
                std::memcpy(q + 32, i4, 32);
                std::memcpy(q + 64, theta + (size_t)n * n_theta, 8 * (size_t)n_theta);
                std::memcpy(q + 64 + 8 * (size_t)n_theta, g.t_change, 8 * (size_t)n_tchange);
            }
        });
    } catch (...) {
        return -2;
    }
    return 0;
}

}  // extern "C"
