// Native reader for the model-input files behind tsf_csv_read / tsf_csv_fetch / tsf_csv_free
// (include/tsf.h).
//
// What it replaces in the reference: ProphetModeler.read_input_dataframe
//   spark.read.csv(path, header=False, schema=MODEL_INPUT_SCHEMA)
//   /root/reference/src/jobs/prophet_modeler.py:102-116 (schema :12-17)
// over a directory of header-less CSV files, Hive-partitioned by series_id
// (tests/fixtures/model-input/series_id=751/sample-model-input.csv: `91,2001-01-05 11:15:00,36445`).
// The JVM reader hands Spark rows; this one parses straight into the four columns the packer
// (tsf_pack_rows) takes: series_id, dim_id, ds [ns since the epoch], y [f64, NaN = null].
// The files are read into memory by a pool of threads, then parsed by the pool in segments
// (a small file is one segment, a big one is cut at line ends every 4 MB); rows come out in
// file order, files in the order given.  gzip / zlib compressed parts (.gz, .deflate) are inflated in
// memory (zlib), as Spark's Hadoop codecs do transparently.
#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/tsf.h"
#include "tsf_pool.h"

namespace {

// One segment of input (a small file, or a piece of a big one cut at line ends) and where its rows go: straight
// into the table's one block of columns at `first` (round 4: a vector per column and segment, then a copy into the
// caller's arrays, touched every row's 32 bytes three times and every page of three allocations once -- the page
// faults, not the parsing, were most of the 0.13 s this reader took on 10 000 part files).
struct SegOut {
    int64_t *sid = nullptr, *did = nullptr, *ds = nullptr;
    double *y = nullptr;
    int64_t first = 0;      // first row of the segment in the block
    int64_t cap = 0;        // lines of the segment: an upper bound of its rows (blank / dropped lines write none)
    int64_t n = 0;          // rows written
    int err = 0;            // TSF_CSV_* code
    int64_t err_line = 0;   // 1-based
    int64_t malformed = 0;  // permissive mode: records that did not match the schema (dropped, counted)
};

// file contents: one uninitialised allocation (a std::vector would zero 170 MB first)
// Large host blocks of the reader -- the 8 MB chunks its threads read file bytes into, the table of parsed columns --
// are kept by the process between uses (round 6), up to TSF_HOST_CACHE_MB megabytes (default 768; 0: never): a block that
// comes back from here is mapped and its pages are resident, where a fresh one costs a page fault and the zeroing of every
// 2 MB page it touches and, later, their unmapping -- measured on the GPU box at a quarter of the read stage for
// 10 000 x 730 rows (profiles/r06_host/).  The jobs' pipelines read chunk k + 2 into what chunk k was read into, and a
// process that runs the job again (a service, a reused Python worker) starts with warm blocks.  Exact sizes only.
struct HostCache {
    struct Entry { void *p; size_t bytes; };
    static std::mutex &mu() { static std::mutex m; return m; }
    static std::vector<Entry> &list() { static std::vector<Entry> *l = new std::vector<Entry>(); return *l; }
    static size_t &held() { static size_t h = 0; return h; }
    static size_t limit() {
        static const size_t lim = [] {
            const char *e = std::getenv("TSF_HOST_CACHE_MB");
            const long v = e ? std::atol(e) : 768;
            return (size_t)(v < 0 ? 0 : v) << 20;
        }();
        return lim;
    }
    static void *take(size_t bytes) {
        std::lock_guard<std::mutex> lk(mu());
        auto &l = list();
        for (size_t i = l.size(); i-- > 0;)
            if (l[i].bytes == bytes) {
                void *p = l[i].p;
                held() -= bytes;
                l[i] = l.back();
                l.pop_back();
                return p;
            }
        return nullptr;
    }
    static bool keep(void *p, size_t bytes) {
        if (!p || bytes == 0) return false;
        std::lock_guard<std::mutex> lk(mu());
        if (held() + bytes > limit()) return false;
        list().push_back(Entry{p, bytes});
        held() += bytes;
        return true;
    }
};

// Bump allocator of one reader thread: the bytes of the files it reads go side by side into 8 MB chunks on
// transparent huge pages.  (One malloc per file -- 10 000 x 146 KB, each its own mmap above glibc's threshold -- was
// 36 000 page faults and 20 000 map / unmap calls contending for the process's mmap lock: 15 ms in a bare process,
// 39 ms inside the job, whose address space also holds the HIP runtime's mappings.)
struct Arena {
    static constexpr size_t CHUNK = (size_t)8 << 20, HUGE = (size_t)2 << 20;
    std::vector<void *> chunks;
    char *cur = nullptr;
    size_t left = 0;
    Arena() = default;
    Arena(const Arena &) = delete;
    Arena &operator=(const Arena &) = delete;
    ~Arena() { for (void *c : chunks) std::free(c); }
    // nullptr: too large for a chunk (the caller takes a block of its own), or out of memory
    char *get(size_t bytes) {
        bytes = (bytes + 63) & ~(size_t)63;
        if (bytes > CHUNK / 4) return nullptr;
        if (bytes > left) {
            void *c = HostCache::take(CHUNK);
            if (!c) {
                if (posix_memalign(&c, HUGE, CHUNK) != 0) return nullptr;
                static const bool thp = !(std::getenv("TSF_CSV_THP") && std::atoi(std::getenv("TSF_CSV_THP")) == 0);   // (dev: probe)
                if (thp) (void)::madvise(c, CHUNK, MADV_HUGEPAGE);
            }
            chunks.push_back(c);
            cur = (char *)c; left = CHUNK;
        }
        char *r = cur;
        cur += bytes; left -= bytes;
        return r;
    }
};

// Large blocks go back to the system on a thread of their own: unmapping 146 MB of file bytes took 20-50 ms of the
// read stage and the 234 MB table 28 ms of whoever dropped it (measured on the GPU box, profiles/r04_host4/) -- time the
// job spends better packing and fitting.  The thread is detached and touches nothing but its own list of blocks (no
// object to outlive, nothing to join at exit or to inherit across a fork).  TSF_CSV_BG_FREE=0: free in place.
struct Reaper {
    static bool enabled() {
        static const bool on = !(std::getenv("TSF_CSV_BG_FREE") && std::atoi(std::getenv("TSF_CSV_BG_FREE")) == 0);
        return on;
    }
    // arena chunks: to the process's cache while it has room, the rest back to the system
    static void give_chunks(std::vector<void *> &&chunks) noexcept {
        std::vector<void *> rest;
        for (void *c : chunks)
            if (!HostCache::keep(c, Arena::CHUNK)) rest.push_back(c);
        chunks.clear();
        if (enabled()) give(std::move(rest));
        else for (void *c : rest) std::free(c);
    }
    static void give(std::vector<void *> &&blocks) noexcept {
        if (blocks.empty()) return;
        std::shared_ptr<std::vector<void *>> held;
        try {
            held = std::make_shared<std::vector<void *>>(std::move(blocks));
        } catch (...) {                     // (blocks is untouched when make_shared throws)
            for (void *c : blocks) std::free(c);
            return;
        }
        try {
            std::thread([held]() { for (void *c : *held) std::free(c); }).detach();
        } catch (...) {                     // no thread to be had: in place
            for (void *c : *held) std::free(c);
        }
    }
};

struct FileBuf {
    char *p = nullptr;
    size_t n = 0;
    bool borrowed = false;          // p belongs to an Arena
    FileBuf() = default;
    FileBuf(const FileBuf &) = delete;
    FileBuf &operator=(const FileBuf &) = delete;
    FileBuf(FileBuf &&o) noexcept : p(o.p), n(o.n), borrowed(o.borrowed) { o.p = nullptr; o.n = 0; o.borrowed = false; }
    FileBuf &operator=(FileBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; borrowed = o.borrowed; o.p = nullptr; o.n = 0; o.borrowed = false; }
        return *this;
    }
    ~FileBuf() { if (!borrowed) std::free(p); }
    bool alloc(size_t bytes, Arena *a = nullptr) {
        release();
        if (a && bytes > 0 && (p = a->get(bytes)) != nullptr) { borrowed = true; n = bytes; return true; }
        p = (char *)std::malloc(bytes ? bytes : 1); n = p ? bytes : 0;
        return p != nullptr;
    }
    void release() { if (!borrowed) std::free(p); p = nullptr; n = 0; borrowed = false; }
};

// days from 1970-01-01 of a proleptic Gregorian date (valid for all int years)
inline int64_t days_from_civil(int64_t y, unsigned m, unsigned d) {
    y -= m <= 2;
    const int64_t era = (y >= 0 ? y : y - 399) / 400;
    const unsigned yoe = (unsigned)(y - era * 400);
    const unsigned doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return era * 146097 + (int64_t)doe - 719468;
}

inline bool digits(const char *p, int n, int64_t *out) {
    int64_t v = 0;
    for (int i = 0; i < n; ++i) {
        unsigned c = (unsigned)(p[i] - '0');
        if (c > 9) return false;
        v = v * 10 + c;
    }
    *out = v;
    return true;
}

// [a, b) trimmed of blanks and one pair of double quotes
inline void trim(const char *&a, const char *&b) {
    while (a < b && (*a == ' ' || *a == '\t')) ++a;
    while (b > a && (b[-1] == ' ' || b[-1] == '\t' || b[-1] == '\r')) --b;
    if (b - a >= 2 && *a == '"' && b[-1] == '"') {
        ++a;
        --b;
    }
}

// yyyy-MM-dd[( |T)HH:mm[:ss[.fffffffff]]][Z]  -> ns since the epoch (naive / UTC)
bool parse_timestamp(const char *a, const char *b, int64_t *out) {
    trim(a, b);
    if (b - a < 10) return false;
    int64_t Y, M, D, h = 0, mi = 0, s = 0, frac = 0;
    if (!digits(a, 4, &Y) || a[4] != '-' || !digits(a + 5, 2, &M) || a[7] != '-' || !digits(a + 8, 2, &D))
        return false;
    if (M < 1 || M > 12 || D < 1 || D > 31) return false;
    const char *p = a + 10;
    if (p < b) {
        if (*p != ' ' && *p != 'T') return false;
        ++p;
        if (b - p < 5 || !digits(p, 2, &h) || p[2] != ':' || !digits(p + 3, 2, &mi)) return false;
        p += 5;
        if (p < b && *p == ':') {
            if (b - p < 3 || !digits(p + 1, 2, &s)) return false;
            p += 3;
            if (p < b && *p == '.') {
                ++p;
                int nd = 0;
                while (p < b && (unsigned)(*p - '0') <= 9) {
                    if (nd < 9) {
                        frac = frac * 10 + (*p - '0');
                        ++nd;
                    }
                    ++p;
                }
                if (nd == 0) return false;
                for (; nd < 9; ++nd) frac *= 10;
            }
        }
        if (p < b && *p == 'Z') ++p;
        if (p != b) return false;
        if (h > 23 || mi > 59 || s > 59) return false;
    }
    static const int mdays[12] = {31, 29, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    if (D > mdays[M - 1]) return false;
    if (M == 2 && D == 29 && !((Y % 4 == 0 && Y % 100 != 0) || Y % 400 == 0)) return false;
    int64_t days = days_from_civil(Y, (unsigned)M, (unsigned)D);
    *out = ((days * 24 + h) * 60 + mi) * 60 * 1000000000ll + s * 1000000000ll + frac;
    return true;
}

bool parse_int(const char *a, const char *b, int64_t *out) {
    trim(a, b);
    if (a >= b) return false;
    bool neg = false;
    if (*a == '-' || *a == '+') {
        neg = *a == '-';
        ++a;
    }
    if (a >= b || b - a > 18) return false;
    int64_t v = 0;
    for (; a < b; ++a) {
        unsigned c = (unsigned)(*a - '0');
        if (c > 9) return false;
        v = v * 10 + c;
    }
    *out = neg ? -v : v;
    return true;
}

// quantity: integer in the reference schema; decimals are accepted too.  Empty = null.
bool parse_quantity(const char *a, const char *b, double *out) {
    trim(a, b);
    if (a >= b) {
        *out = std::numeric_limits<double>::quiet_NaN();
        return true;
    }
    int64_t iv;
    if (parse_int(a, b, &iv)) {
        *out = (double)iv;
        return true;
    }
    std::string tmp(a, b);
    char *end = nullptr;
    double v = std::strtod(tmp.c_str(), &end);
    if (end == tmp.c_str() || *end != '\0' || std::isinf(v)) return false;
    *out = v;
    return true;
}

// layout: one letter per column of the file: s series_id, d dim_id, t start_time, q quantity,
// x ignored.  Parses the lines of [p, end) (whole lines); line numbers in errors are relative to p.
void parse_range(const char *p, const char *end, const char *layout, int ncol, int64_t sid_const,
                 bool permissive, SegOut &out) {
    int64_t line = 0, w = 0;
    while (p < end) {
        const char *eol = (const char *)std::memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        ++line;
        const char *le = eol;
        while (le > p && (le[-1] == '\r' || le[-1] == ' ')) --le;
        if (le == p) {              // blank line
            p = eol + 1;
            continue;
        }
        int64_t sid = sid_const, did = 0, ds = 0;
        double q = 0;
        const char *fa = p;
        int col = 0;
        bool ok = true;
        for (; col < ncol && ok; ++col) {
            const char *fb = (col == ncol - 1) ? le : (const char *)std::memchr(fa, ',', (size_t)(le - fa));
            if (!fb) {
                ok = false;
                break;
            }
            switch (layout[col]) {
                case 's': ok = parse_int(fa, fb, &sid); break;
                case 'd': ok = parse_int(fa, fb, &did); break;
                case 't': ok = parse_timestamp(fa, fb, &ds); break;
                case 'q': ok = parse_quantity(fa, fb, &q); break;
                default: break;
            }
            fa = fb + 1;
        }
        if (!ok || col != ncol) {
            if (!permissive) {
                out.err = TSF_CSV_E_PARSE;
                out.err_line = line;
                return;
            }
            // Spark 2.4 PERMISSIVE: a record that does not convert becomes a row of NULLS -- dim_id
            // included.  There is no null key in these int64 columns, and a made-up one (0) would either
            // join a real series' rows or form a one-row group whose fit raises; the record is DROPPED
            // here and counted (tsf_csv_malformed), which is also what its only observable effect on a fit
            // is: fbprophet drops rows with a null y.
            out.malformed++;
            p = eol + 1;
            continue;
        }
        if (w >= out.cap) {         // (cannot happen: cap counts every line)
            out.err = TSF_CSV_E_PARSE;
            out.err_line = line;
            return;
        }
        out.sid[w] = sid;
        out.did[w] = did;
        out.ds[w] = ds;
        out.y[w] = q;
        ++w;
        p = eol + 1;
    }
    out.n = w;
}

// lines of [p, end): newline characters, plus one for an unterminated last line
int64_t count_lines(const char *p, const char *end) {
    int64_t n = 0;
    const char *q = p;
    while (q < end) {
        const char *nl = (const char *)std::memchr(q, '\n', (size_t)(end - q));
        if (!nl) break;
        ++n;
        q = nl + 1;
    }
    if (q < end) ++n;
    return n;
}

// gzip members (.gz: what Hadoop's GzipCodec writes and spark.read.csv decompresses transparently,
// prophet_modeler.py:109-114) and zlib streams (.deflate: DefaultCodec), recognised by their header bytes;
// several concatenated members are one file.  false = corrupt or truncated stream.
bool inflate_all(const FileBuf &in, std::vector<char> &out) {
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15 + 32) != Z_OK) return false;       // + 32: gzip or zlib header, detected
    out.clear();
    out.resize(in.n * 6 + 4096);
    // (avail_in is 32 bits wide: a compressed part of 4 GB or more is fed in pieces of at most 1 GB -- round-4 advice:
    // the cast used to truncate it and the file was reported corrupt)
    size_t fed = std::min<size_t>(in.n, (size_t)1 << 30);
    zs.next_in = (Bytef *)in.p;
    zs.avail_in = (uInt)fed;
    size_t have = 0;
    bool ok = true;
    for (;;) {
        if (zs.avail_in == 0 && fed < in.n) {
            const size_t k = std::min<size_t>(in.n - fed, (size_t)1 << 30);
            zs.next_in = (Bytef *)in.p + fed;
            zs.avail_in = (uInt)k;
            fed += k;
        }
        if (have == out.size()) out.resize(out.size() * 2);
        zs.next_out = (Bytef *)out.data() + have;
        zs.avail_out = (uInt)std::min<size_t>(out.size() - have, (size_t)1 << 30);
        const uInt before = zs.avail_out;
        const int rc = inflate(&zs, Z_NO_FLUSH);
        have += before - zs.avail_out;
        if (rc == Z_STREAM_END) {
            if (zs.avail_in == 0 && fed == in.n) break;
            if (inflateReset(&zs) != Z_OK) { ok = false; break; }   // next member
            continue;
        }
        if (rc != Z_OK) { ok = false; break; }
        if (zs.avail_in == 0 && fed == in.n && zs.avail_out != 0) { ok = false; break; }      // truncated
    }
    inflateEnd(&zs);
    out.resize(have);
    return ok;
}

// Spark picks the codec by the file's SUFFIX (.gz -> GzipCodec, .deflate -> DefaultCodec); so does this reader, and
// the gzip magic is accepted under any name (no CSV record starts with 0x1f 0x8b).  A zlib header, two printable-range
// bytes such as "x^", is NOT sniffed from the content any more: a plain file may start with it (round-4 advice).
bool has_suffix(const char *path, const char *suf) {
    const size_t n = std::strlen(path), k = std::strlen(suf);
    return n >= k && std::strcmp(path + n - k, suf) == 0;
}
bool is_deflated(const FileBuf &b, const char *path) {
    if (b.n < 2) return false;
    const unsigned char b0 = (unsigned char)b.p[0], b1 = (unsigned char)b.p[1];
    if (b0 == 0x1f && b1 == 0x8b) return true;                                  // gzip
    if (!(path && has_suffix(path, ".deflate"))) return false;
    return (b0 & 0x0f) == 8 && (b0 >> 4) <= 7 && ((b0 << 8) | b1) % 31 == 0 && b0 == 0x78;    // zlib, 32 K window
}

// the whole file with one open / fstat / read / close (10 000 part files of a Hive-partitioned input are
// 40 000 system calls instead of the 70 000 of fopen / fseek / ftell / fread / fclose)
bool read_whole(const char *path, FileBuf &buf, bool *corrupt, Arena *arena = nullptr) {
    *corrupt = false;
    const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat sb;
    if (::fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { ::close(fd); return false; }
    if (!buf.alloc((size_t)sb.st_size, arena)) { ::close(fd); throw std::bad_alloc(); }
    size_t got = 0;
    while (got < buf.n) {
        const ssize_t k = ::read(fd, buf.p + got, buf.n - got);
        if (k <= 0) break;
        got += (size_t)k;
    }
    ::close(fd);
    if (got != buf.n) return false;
    if (is_deflated(buf, path)) {
        std::vector<char> raw;
        if (!inflate_all(buf, raw)) { *corrupt = true; return false; }
        if (!buf.alloc(raw.size())) throw std::bad_alloc();
        std::memcpy(buf.p, raw.data(), raw.size());
    }
    return true;
}

// a file larger than this is cut at line ends into pieces parsed by different threads
constexpr size_t SEGMENT_BYTES = (size_t)4 << 20;

struct Segment {
    int32_t file;
    size_t begin, end;      // byte range of whole lines inside the file's buffer
};

}  // namespace

struct tsf_csv {
    std::vector<SegOut> segs;       // one entry per SEGMENT, in file order then byte order
    int64_t *sid = nullptr, *did = nullptr, *ds = nullptr;     // the table: one block, [cap] rows per column
    double *y = nullptr;
    void *block = nullptr;
    size_t block_bytes = 0;         // > 0: a 2 MB-aligned block of that size (HostCache takes it back)
    int64_t n_rows = 0, malformed = 0;
    int n_threads = 1;
    ~tsf_csv() { std::free(block); }
};

namespace {

// item(i, w): item i on worker w (0 <= w < n_threads)
template <class F>
void run_workers_w(int n_threads, int64_t n_items, std::atomic<int> &oom, F item) {
    std::atomic<int64_t> next(0);
    auto worker = [&](int w) {
        for (;;) {
            int64_t i = next.fetch_add(1);
            if (i >= n_items) break;
            try {
                item(i, w);
            } catch (...) {
                oom.store(1);
            }
        }
    };
    if (n_threads <= 1 || n_items < 2) {
        worker(0);
        return;
    }
    // (the library's pool, tsf_pool.h: a phase used to start and join its own threads -- ten times per job)
    const int k = (int64_t)n_threads < n_items ? n_threads : (int)n_items;
    try {
        tsfpool::Pool::get().run(k, worker);
    } catch (...) {
        oom.store(1);
    }
}

template <class F>
void run_workers(int n_threads, int64_t n_items, std::atomic<int> &oom, F item) {
    run_workers_w(n_threads, n_items, oom, [&](int64_t i, int) { item(i); });
}

}  // namespace

// tsf_csv_read and tsf_csv_read_loaded: `loaded` (or null) holds the bytes of the files already -- one entry per file,
// `loaded_ok[i]` = 1 read, 2 corrupt compressed stream, 0 could not be read (tsf_csv_discover with preload)
static int read_impl(int32_t n_files, const char *const *paths, const int64_t *series_id,
              const char *layout, int32_t n_threads, tsf_csv **out, int64_t *n_rows,
              int32_t *err_file, int64_t *err_line, FileBuf *loaded, const char *loaded_ok) {
    if (!out || n_files < 0 || (n_files > 0 && !paths && !loaded) || !layout) return -1;
    *out = nullptr;
    int ncol = (int)std::strlen(layout);
    // a trailing '?' = permissive mode (spark.read.csv's default mode=PERMISSIVE)
    const bool permissive = ncol > 0 && layout[ncol - 1] == '?';
    if (permissive) --ncol;
    bool has_s = false, has_d = false, has_t = false, has_q = false;
    for (int i = 0; i < ncol; ++i) {
        char c = layout[i];
        if (c == 's') has_s = true;
        else if (c == 'd') has_d = true;
        else if (c == 't') has_t = true;
        else if (c == 'q') has_q = true;
        else if (c != 'x') return -1;
    }
    if (!has_d || !has_t || !has_q || ncol < 3 || ncol > 16) return -1;
    if (!has_s && !series_id && n_files > 0) return -1;
    tsf_csv *t = nullptr;
    try {
        t = new tsf_csv();
        int hw = (int)std::thread::hardware_concurrency();
        if (hw < 1) hw = 1;
        t->n_threads = n_threads > 0 ? n_threads : (hw < 32 ? hw : 32);
        std::atomic<int> oom(0);
        // TSF_CSV_TIMING=1 (dev): wall clock of the phases on stderr
        static const bool timing = std::getenv("TSF_CSV_TIMING") != nullptr;
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_0 = timing ? now() : 0.0;
        double t_read = 0, t_count = 0, t_alloc = 0, t_parse = 0;
        // ---- phase 1: the files into memory (parallel); every thread also counts the lines of its files
        std::vector<Arena> arenas((size_t)t->n_threads);       // declared before the buffers that borrow from them
        std::vector<FileBuf> bufs_store((size_t)n_files);
        std::vector<FileBuf> &bufs = bufs_store;
        std::vector<char> opened((size_t)n_files, 0);
        if (loaded) {
            for (int32_t i = 0; i < n_files; ++i) { bufs[(size_t)i] = std::move(loaded[i]); opened[(size_t)i] = loaded_ok[i]; }
        } else {
            run_workers_w(t->n_threads, n_files, oom, [&](int64_t i, int w) {
                bool corrupt = false;
                opened[(size_t)i] = read_whole(paths[i], bufs[(size_t)i], &corrupt, &arenas[(size_t)w]) ? 1 : (corrupt ? 2 : 0);
            });
        }
        if (oom.load()) {
            delete t;
            return -2;
        }
        for (int32_t i = 0; i < n_files; ++i)
            if (opened[(size_t)i] != 1) {
                if (err_file) *err_file = i;
                if (err_line) *err_line = opened[(size_t)i] == 2 ? -1 : 0;     // -1: a corrupt / truncated compressed stream
                delete t;
                return TSF_CSV_E_OPEN;
            }
        if (timing) t_read = now();
        // ---- segments: small files whole, big ones cut after a line end every SEGMENT_BYTES
        std::vector<Segment> segs;
        for (int32_t i = 0; i < n_files; ++i) {
            const FileBuf &b = bufs[(size_t)i];
            size_t at = 0;
            while (b.n - at > SEGMENT_BYTES + SEGMENT_BYTES / 2) {
                const char *nl = (const char *)std::memchr(b.p + at + SEGMENT_BYTES, '\n', b.n - at - SEGMENT_BYTES);
                if (!nl) break;
                size_t stop = (size_t)(nl - b.p) + 1;
                segs.push_back(Segment{i, at, stop});
                at = stop;
            }
            segs.push_back(Segment{i, at, b.n});
        }
        t->segs.resize(segs.size());
        // ---- lines per segment (an upper bound of its rows), then ONE block for the four columns
        run_workers(t->n_threads, (int64_t)segs.size(), oom, [&](int64_t k) {
            const Segment &sg = segs[(size_t)k];
            const char *base = bufs[(size_t)sg.file].p;
            t->segs[(size_t)k].cap = count_lines(base + sg.begin, base + sg.end);
        });
        if (timing) t_count = now();
        int64_t cap = 0;
        for (SegOut &so : t->segs) { so.first = cap; cap += so.cap; }
        {
            // one block for the table, on transparent huge pages where the system grants them: 234 MB of 4 KB pages
            // are 57 000 page faults while the rows are written and 50 ms of unmapping when the table is freed
            const size_t bytes = (size_t)(cap > 0 ? cap : 1) * 32, huge = (size_t)2 << 20;
            const size_t rounded = (bytes + huge - 1) / huge * huge;
            if (bytes >= 4 * huge && (t->block = HostCache::take(rounded)) != nullptr)
                t->block_bytes = rounded;
            else if (bytes >= 4 * huge && posix_memalign(&t->block, huge, rounded) == 0) {
                (void)::madvise(t->block, rounded, MADV_HUGEPAGE);
                t->block_bytes = rounded;
            } else
                t->block = std::malloc(bytes);
        }
        if (!t->block) { delete t; return -2; }
        t->sid = (int64_t *)t->block; t->did = t->sid + cap; t->ds = t->did + cap; t->y = (double *)(t->ds + cap);
        for (SegOut &so : t->segs) { so.sid = t->sid + so.first; so.did = t->did + so.first; so.ds = t->ds + so.first; so.y = t->y + so.first; }
        if (timing) t_alloc = now();
        // ---- phase 2: parse in place (parallel over segments)
        run_workers(t->n_threads, (int64_t)segs.size(), oom, [&](int64_t k) {
            const Segment &sg = segs[(size_t)k];
            const char *base = bufs[(size_t)sg.file].p;
            parse_range(base + sg.begin, base + sg.end, layout, ncol,
                        series_id ? series_id[sg.file] : 0, permissive, t->segs[(size_t)k]);
        });
        if (oom.load()) {
            delete t;
            return -2;
        }
        if (timing) {
            t_parse = now();
            std::fprintf(stderr, "[csv-timing] %d files, %d threads: read %.1f ms, segments + line count %.1f, block %.1f, parse %.1f\n",
                         (int)n_files, t->n_threads, t_read - t_0, t_count - t_read, t_alloc - t_count, t_parse - t_alloc);
        }
        int64_t pos = 0;
        bool holes = false;
        for (size_t k = 0; k < segs.size(); ++k) {
            const SegOut &so = t->segs[k];
            if (so.err) {
                // line number inside the file = lines of the earlier segments + line in this one
                const Segment &sg = segs[k];
                const char *base = bufs[(size_t)sg.file].p;
                int64_t before = 0;
                for (const char *q = base; q < base + sg.begin; ++q) before += (*q == '\n');
                if (err_file) *err_file = sg.file;
                if (err_line) *err_line = before + so.err_line;
                int e = so.err;
                delete t;
                return e;
            }
            holes = holes || so.first != pos;
            pos += so.n;
            t->malformed += so.malformed;
        }
        t->n_rows = pos;
        {
            // the file buffers go back in parallel too: 10 000 frees are 20 ms on one thread, 4 ms on the pool
            const double t_a = timing ? now() : 0.0;
            run_workers(t->n_threads, n_files, oom, [&](int64_t i) { bufs[(size_t)i].release(); });
            {
                std::vector<void *> all;
                for (Arena &a : arenas) { all.insert(all.end(), a.chunks.begin(), a.chunks.end()); a.chunks.clear(); }
                Reaper::give_chunks(std::move(all));
            }
            if (timing) std::fprintf(stderr, "[csv-timing] releasing the file buffers %.1f ms\n", now() - t_a);
        }
        if (holes) {
            // blank or dropped lines left gaps between the segments' rows: close them (in order, so that a
            // segment only ever moves down onto rows already moved), then the four columns themselves --
            // did / ds / y start at multiples of cap, the caller's view is [n_rows] each
            pos = 0;
            for (SegOut &so : t->segs) {
                if (so.first != pos && so.n > 0) {
                    std::memmove(t->sid + pos, t->sid + so.first, (size_t)so.n * 8);
                    std::memmove(t->did + pos, t->did + so.first, (size_t)so.n * 8);
                    std::memmove(t->ds + pos, t->ds + so.first, (size_t)so.n * 8);
                    std::memmove(t->y + pos, t->y + so.first, (size_t)so.n * 8);
                }
                so.first = pos;
                pos += so.n;
            }
        }
    } catch (const std::bad_alloc &) {
        delete t;
        return -2;
    } catch (...) {
        delete t;
        return -3;
    }
    *out = t;
    if (n_rows) *n_rows = t->n_rows;
    return 0;
}

extern "C" {

int tsf_csv_read(int32_t n_files, const char *const *paths, const int64_t *series_id,
                 const char *layout, int32_t n_threads, tsf_csv **out, int64_t *n_rows,
                 int32_t *err_file, int64_t *err_line) {
    if (n_files > 0 && !paths) return -1;
    return read_impl(n_files, paths, series_id, layout, n_threads, out, n_rows, err_file, err_line, nullptr, nullptr);
}

int64_t tsf_csv_malformed(const tsf_csv *t) { return t ? t->malformed : -1; }

int tsf_csv_columns(tsf_csv *t, const int64_t **series_id, const int64_t **dim_id, const int64_t **ds, const double **y) {
    if (!t) return -1;
    if (series_id) *series_id = t->sid;
    if (dim_id) *dim_id = t->did;
    if (ds) *ds = t->ds;
    if (y) *y = t->y;
    return 0;
}

int tsf_csv_fetch(tsf_csv *t, int64_t *series_id, int64_t *dim_id, int64_t *ds, double *y) {
    if (!t) return -1;
    // (chunks of 1 M rows per column copied by the pool)
    const int64_t CH = (int64_t)1 << 20;
    const int64_t nch = (t->n_rows + CH - 1) / CH;
    std::atomic<int> oom(0);
    run_workers(t->n_threads, nch * 4, oom, [&](int64_t j) {
        const int64_t c = j / nch, at = (j % nch) * CH;
        const size_t n = (size_t)((t->n_rows - at < CH) ? t->n_rows - at : CH) * 8;
        if (c == 0 && series_id) std::memcpy(series_id + at, t->sid + at, n);
        if (c == 1 && dim_id) std::memcpy(dim_id + at, t->did + at, n);
        if (c == 2 && ds) std::memcpy(ds + at, t->ds + at, n);
        if (c == 3 && y) std::memcpy(y + at, t->y + at, n);
    });
    return 0;
}

void tsf_csv_free(tsf_csv *t) {
    if (t && t->block && t->block_bytes && HostCache::keep(t->block, t->block_bytes)) t->block = nullptr;
    if (t && t->block && Reaper::enabled()) {
        std::vector<void *> b{t->block};
        t->block = nullptr;
        Reaper::give(std::move(b));
    }
    delete t;
}
// ---- input discovery ---------------------------------------------------------------------------
// What spark.read.csv(path) reads under a directory (prophet_modeler.py:109-114) and how it finds the partition
// column: every regular file below `root` whose name -- and the name of every directory on the way -- does not
// start with '_' or '.', a `series_id=<int>` directory supplying series_id for everything below it.  The Python
// walk (jobs/prophet_modeler.find_model_input: os.scandir, one directory at a time) took 0.07 s of the 0.16 s
// read stage on 10 000 partition directories; here the directories are read by the pool and Python never sees a
// path unless something is wrong with one.
struct tsf_csv_dir {
    struct Item { std::string path; int64_t sid; bool has; FileBuf buf; char ok = 0; };
    std::vector<Item> items;
    std::vector<Arena> arenas;      // preload: the bytes of the files, read by the thread that listed their directory
    bool preloaded = false;
    ~tsf_csv_dir() {
        std::vector<void *> all;
        for (Arena &a : arenas) { all.insert(all.end(), a.chunks.begin(), a.chunks.end()); a.chunks.clear(); }
        Reaper::give_chunks(std::move(all));
    }
    std::vector<const char *> paths;
    std::vector<int64_t> sids;
    int32_t n_part = 0;
    int err = 0;
    std::string err_path;
    int nested = 0;                 // a `series_id=` directory below another one with a different value was seen
};

namespace {

bool hidden_name(const char *n) { return n[0] == '_' || n[0] == '.'; }

bool has_suffix(const std::string &s, const char *suf) {
    const size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// one directory: files appended to `out`, sub-directories to `dirs`
void scan_dir(const std::string &d, int64_t sid, bool has, std::vector<tsf_csv_dir::Item> &out,
              std::vector<tsf_csv_dir::Item> &dirs, int &err, std::string &err_path, Arena *arena, int *nested = nullptr) {
    DIR *h = ::opendir(d.c_str());
    if (!h) { err = TSF_CSV_E_OPEN; err_path = d; return; }
    while (struct dirent *e = ::readdir(h)) {
        const char *nm = e->d_name;
        if (hidden_name(nm)) continue;          // (covers "." and "..")
        std::string p = d + "/" + nm;
        unsigned char ty = e->d_type;
        if (ty == DT_UNKNOWN || ty == DT_LNK) {
            struct stat sb;
            if (::stat(p.c_str(), &sb) != 0) continue;          // dangling link: Spark lists nothing for it
            ty = S_ISDIR(sb.st_mode) ? DT_DIR : (S_ISREG(sb.st_mode) ? DT_REG : DT_UNKNOWN);
        }
        if (ty == DT_DIR) {
            int64_t v = sid;
            bool hv = has;
            if (std::strncmp(nm, "series_id=", 10) == 0) {
                char *end = nullptr;
                errno = 0;
                const long long q = std::strtoll(nm + 10, &end, 10);
                if (end == nm + 10 || *end != '\0' || errno != 0) { err = TSF_CSV_E_PARSE; err_path = p; continue; }
                if (has && q != sid && nested) *nested = 1;
                v = q; hv = true;
            }
            dirs.push_back(tsf_csv_dir::Item{std::move(p), v, hv, FileBuf(), 0});
        } else if (ty == DT_REG) {
            bool codec = false;
            for (const char *suf : {".bz2", ".snappy", ".lz4", ".zst", ".xz"})
                if (has_suffix(p, suf)) { err = TSF_CSV_E_CODEC; err_path = p; codec = true; }
            tsf_csv_dir::Item it{std::move(p), sid, has, FileBuf(), 0};
            if (arena && !codec) {      // preload: the listing thread reads the file while its directory entry is warm
                bool corrupt = false;
                it.ok = read_whole(it.path.c_str(), it.buf, &corrupt, arena) ? 1 : (corrupt ? 2 : 0);
            }
            out.push_back(std::move(it));
        }
    }
    ::closedir(h);
}

}  // namespace

// start (or null): the walk begins at these entries -- directories with the partition value they carry, files as they are
// -- instead of at `root` (tsf_csv_root_load: a range of the root's children)
static int discover_impl(const char *root, int32_t n_threads, tsf_csv_dir **out, int32_t *n_files, int32_t *n_partitioned,
                         bool preload, const std::vector<tsf_csv_dir::Item> *start = nullptr,
                         const std::vector<char> *start_is_dir = nullptr) {
    if ((!root && !start) || !out) return -1;
    *out = nullptr;
    tsf_csv_dir *d = nullptr;
    static const bool timing = std::getenv("TSF_CSV_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_0 = timing ? now() : 0.0;
    double t_walk = 0;
    try {
        d = new tsf_csv_dir();
        struct stat sb;
        d->preloaded = preload;
        if (!start && ::stat(root, &sb) == 0 && S_ISREG(sb.st_mode)) {
            tsf_csv_dir::Item it{root, 0, false, FileBuf(), 0};
            if (preload) {
                bool corrupt = false;
                it.ok = read_whole(root, it.buf, &corrupt, nullptr) ? 1 : (corrupt ? 2 : 0);
            }
            d->items.push_back(std::move(it));
        } else if (start || (::stat(root, &sb) == 0 && S_ISDIR(sb.st_mode))) {
            int hw = (int)std::thread::hardware_concurrency();
            if (hw < 1) hw = 1;
            const int nt = n_threads > 0 ? n_threads : (hw < 32 ? hw : 32);
            // breadth first: the directories of one level are read by the pool, each thread into its own lists
            std::vector<tsf_csv_dir::Item> level;
            if (preload) d->arenas = std::vector<Arena>((size_t)nt);
            if (start) {
                std::vector<size_t> plain;
                for (size_t i = 0; i < start->size(); ++i) {
                    const tsf_csv_dir::Item &s = (*start)[i];
                    if ((*start_is_dir)[i]) level.push_back(tsf_csv_dir::Item{s.path, s.sid, s.has, FileBuf(), 0});
                    else plain.push_back(i);
                }
                for (size_t i : plain) {          // files among the root's children (few, if any)
                    const tsf_csv_dir::Item &s = (*start)[i];
                    tsf_csv_dir::Item it{s.path, s.sid, s.has, FileBuf(), 0};
                    bool codec = false;
                    for (const char *suf : {".bz2", ".snappy", ".lz4", ".zst", ".xz"})
                        if (has_suffix(it.path, suf)) { d->err = TSF_CSV_E_CODEC; d->err_path = it.path; codec = true; }
                    if (preload && !codec) {
                        bool corrupt = false;
                        it.ok = read_whole(it.path.c_str(), it.buf, &corrupt, &d->arenas[0]) ? 1 : (corrupt ? 2 : 0);
                    }
                    d->items.push_back(std::move(it));
                }
            } else {
                std::string r(root);
                while (r.size() > 1 && r.back() == '/') r.pop_back();
                level.push_back(tsf_csv_dir::Item{r, 0, false, FileBuf(), 0});
            }
            while (!level.empty()) {
                const int64_t n = (int64_t)level.size();
                const int k = (int)std::min<int64_t>(nt, n);
                std::vector<std::vector<tsf_csv_dir::Item>> files((size_t)k), dirs((size_t)k);
                std::vector<int> errs((size_t)k, 0), nest((size_t)k, 0);
                std::vector<std::string> eps((size_t)k);
                std::atomic<int64_t> next(0);
                auto work = [&](int w) {
                    for (;;) {
                        const int64_t i = next.fetch_add(1);
                        if (i >= n) break;
                        scan_dir(level[(size_t)i].path, level[(size_t)i].sid, level[(size_t)i].has, files[(size_t)w], dirs[(size_t)w],
                                 errs[(size_t)w], eps[(size_t)w], preload ? &d->arenas[(size_t)w] : nullptr, &nest[(size_t)w]);
                    }
                };
                if (k <= 1) work(0);
                else tsfpool::Pool::get().run(k, work);
                level.clear();
                for (int w = 0; w < k; ++w) {
                    if (nest[(size_t)w]) d->nested = 1;
                    if (errs[(size_t)w] && !d->err) { d->err = errs[(size_t)w]; d->err_path = eps[(size_t)w]; }
                    for (auto &it : files[(size_t)w]) d->items.push_back(std::move(it));
                    for (auto &it : dirs[(size_t)w]) level.push_back(std::move(it));
                }
            }
        }
        if (timing) t_walk = now();
        // (partitioned files first, by partition value, then by path: what the packer wants to see -- a table
        // that already is grouped)
        std::sort(d->items.begin(), d->items.end(), [](const tsf_csv_dir::Item &a, const tsf_csv_dir::Item &b) {
            if (a.has != b.has) return a.has;
            if (a.has && a.sid != b.sid) return a.sid < b.sid;
            return a.path < b.path;
        });
        d->paths.reserve(d->items.size());
        d->sids.reserve(d->items.size());
        for (const auto &it : d->items) {
            d->paths.push_back(it.path.c_str());
            d->sids.push_back(it.sid);
            d->n_part += it.has ? 1 : 0;
        }
    } catch (const std::bad_alloc &) {
        delete d;
        return -2;
    } catch (...) {
        delete d;
        return -3;
    }
    if (timing) std::fprintf(stderr, "[csv-timing] discover: walk %.1f ms, sort + lists %.1f ms, %d files\n", t_walk - t_0, now() - t_walk, (int)d->items.size());
    *out = d;
    if (n_files) *n_files = (int32_t)d->items.size();
    if (n_partitioned) *n_partitioned = d->n_part;
    return d->err;
}

int tsf_csv_discover(const char *root, int32_t n_threads, tsf_csv_dir **out, int32_t *n_files, int32_t *n_partitioned) {
    return discover_impl(root, n_threads, out, n_files, n_partitioned, false);
}

int tsf_csv_discover_load(const char *root, int32_t n_threads, tsf_csv_dir **out, int32_t *n_files, int32_t *n_partitioned) {
    return discover_impl(root, n_threads, out, n_files, n_partitioned, true);
}

int tsf_csv_read_loaded(tsf_csv_dir *d, int32_t first, int32_t count, const char *layout, int32_t n_threads,
                        tsf_csv **out, int64_t *n_rows, int32_t *err_file, int64_t *err_line) {
    if (!d || !d->preloaded || first < 0 || count < 0 || (size_t)first + (size_t)count > d->items.size()) return -1;
    try {
        std::vector<FileBuf> bufs((size_t)count);
        std::vector<char> ok((size_t)count);
        for (int32_t i = 0; i < count; ++i) {
            bufs[(size_t)i] = std::move(d->items[(size_t)(first + i)].buf);
            ok[(size_t)i] = d->items[(size_t)(first + i)].ok;
            d->items[(size_t)(first + i)].ok = 0;          // a range is handed over once
        }
        const int rc = read_impl(count, d->paths.data() + first, d->sids.data() + first, layout, n_threads, out, n_rows,
                                 err_file, err_line, bufs.data(), ok.data());
        return rc;                                         // (err_file is relative to `first`)
    } catch (const std::bad_alloc &) {
        return -2;
    }
}

// ---- the input directory in chunks (round 6) -----------------------------------------------------
// tsf_csv_root_open lists the CHILDREN of the input directory once; tsf_csv_root_load walks and loads the subtrees of a
// range of them -- so that a job can read, fit and persist partition directories [0, c), [c, 2c), ... as a pipeline
// (jobs/prophet_modeler.ProphetModeler.model: files of chunk k + 1 are read while chunk k is on the GPU and chunk k - 1
// goes to parquet).  *hive_only = 1 when every child is a `series_id=<int>` directory and no value occurs twice: chunks
// of children then hold disjoint sets of series, and fitting chunk by chunk fits every series once, on all its rows --
// the layout the reference reads (tests/fixtures/model-input/series_id=751/...).  Anything else: read the tree whole.
struct tsf_csv_root {
    std::vector<tsf_csv_dir::Item> kids;      // sorted: partition directories by value, then the rest by name
    std::vector<char> is_dir;
    int hive_only = 0;
};

int tsf_csv_root_open(const char *root, tsf_csv_root **out, int32_t *n_children, int32_t *hive_only) {
    if (!root || !out) return -1;
    *out = nullptr;
    try {
        std::unique_ptr<tsf_csv_root> r(new tsf_csv_root());
        struct stat sb;
        if (::stat(root, &sb) != 0) return TSF_CSV_E_OPEN;
        if (S_ISDIR(sb.st_mode)) {
            std::string base(root);
            while (base.size() > 1 && base.back() == '/') base.pop_back();
            std::vector<tsf_csv_dir::Item> files, dirs;
            int err = 0;
            std::string ep;
            scan_dir(base, 0, false, files, dirs, err, ep, nullptr);
            if (err == TSF_CSV_E_OPEN) return TSF_CSV_E_OPEN;
            bool hive = (err == 0) && files.empty() && !dirs.empty();
            for (const auto &d : dirs) hive = hive && d.has;
            std::sort(dirs.begin(), dirs.end(), [](const tsf_csv_dir::Item &a, const tsf_csv_dir::Item &b) {
                if (a.has != b.has) return a.has;
                if (a.has && a.sid != b.sid) return a.sid < b.sid;
                return a.path < b.path;
            });
            for (size_t i = 1; i < dirs.size() && hive; ++i) hive = dirs[i].sid != dirs[i - 1].sid;
            std::sort(files.begin(), files.end(), [](const tsf_csv_dir::Item &a, const tsf_csv_dir::Item &b) { return a.path < b.path; });
            for (auto &d : dirs) { r->kids.push_back(std::move(d)); r->is_dir.push_back(1); }
            for (auto &f : files) { r->kids.push_back(std::move(f)); r->is_dir.push_back(0); }
            r->hive_only = hive ? 1 : 0;
        }
        if (n_children) *n_children = (int32_t)r->kids.size();
        if (hive_only) *hive_only = r->hive_only;
        *out = r.release();
        return 0;
    } catch (const std::bad_alloc &) {
        return -2;
    } catch (...) {
        return -3;
    }
}

int tsf_csv_root_load(tsf_csv_root *r, int32_t first, int32_t count, int32_t n_threads, tsf_csv_dir **out,
                      int32_t *n_files, int32_t *n_partitioned, int32_t *nested) {
    if (!r || !out || first < 0 || count < 0 || (size_t)first + (size_t)count > r->kids.size()) return -1;
    try {
        std::vector<tsf_csv_dir::Item> start;
        std::vector<char> is_dir;
        for (int32_t i = first; i < first + count; ++i) {
            const tsf_csv_dir::Item &k = r->kids[(size_t)i];
            start.push_back(tsf_csv_dir::Item{k.path, k.sid, k.has, FileBuf(), 0});
            is_dir.push_back(r->is_dir[(size_t)i]);
        }
        const int rc = discover_impl(nullptr, n_threads, out, n_files, n_partitioned, true, &start, &is_dir);
        if (nested) *nested = (*out) ? (*out)->nested : 0;
        return rc;
    } catch (const std::bad_alloc &) {
        return -2;
    }
}

void tsf_csv_root_free(tsf_csv_root *r) { delete r; }

const char *const *tsf_csv_dir_paths(const tsf_csv_dir *d) { return d ? d->paths.data() : nullptr; }
const int64_t *tsf_csv_dir_series_id(const tsf_csv_dir *d) { return d ? d->sids.data() : nullptr; }
const char *tsf_csv_dir_error_path(const tsf_csv_dir *d) { return d ? d->err_path.c_str() : ""; }
void tsf_csv_dir_free(tsf_csv_dir *d) { delete d; }



// ---- forecast sink ----------------------------------------------------------------------------
// ProphetScorer.convert_forecasts + write_forecasts (/root/reference/src/jobs/prophet_scorer.py:
// 130-150): one CSV with header
//   created_timestamp,series_id,dim_id,forecast_date,forecast_timestamp,forecast_quantity
// forecast_date = ds.date() as %Y-%m-%d (:107-108), forecast_timestamp in Spark 2.4's default
// CSV timestampFormat yyyy-MM-dd'T'HH:mm:ss.SSSXXX with the wall time taken as UTC.
}  // extern "C"

namespace {

inline void civil_from_days(int64_t z, int64_t *y, unsigned *m, unsigned *d) {
    z += 719468;
    const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
    const unsigned doe = (unsigned)(z - era * 146097);
    const unsigned yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
    const int64_t yy = (int64_t)yoe + era * 400;
    const unsigned doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
    const unsigned mp = (5 * doy + 2) / 153;
    *d = doy - (153 * mp + 2) / 5 + 1;
    *m = mp < 10 ? mp + 3 : mp - 9;
    *y = yy + (*m <= 2);
}

inline char *put2(char *p, unsigned v) {
    p[0] = (char)('0' + v / 10);
    p[1] = (char)('0' + v % 10);
    return p + 2;
}

inline char *put_int(char *p, int64_t v) {
    char tmp[24];
    int n = 0;
    uint64_t u = v < 0 ? (uint64_t)(-(v + 1)) + 1 : (uint64_t)v;
    if (v < 0) *p++ = '-';
    do {
        tmp[n++] = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    while (n) *p++ = tmp[--n];
    return p;
}

// yyyy-MM-dd (returns end); years outside 0..9999 are not expected from datetime64[ns]
inline char *put_date(char *p, int64_t days) {
    int64_t y;
    unsigned m, d;
    civil_from_days(days, &y, &m, &d);
    p = put2(p, (unsigned)(y / 100));
    p = put2(p, (unsigned)(y % 100));
    *p++ = '-';
    p = put2(p, m);
    *p++ = '-';
    return put2(p, d);
}

// the sink, for id / quantity columns of either width (the scorer's frame holds int32: prophet_scorer.py:73, :99-104)
template <class I>
int write_forecasts(const char *path, const char *created_timestamp, int64_t n,
                    const I *series_id, const I *dim_id, const int64_t *ds,
                    const I *quantity, int32_t n_threads) {
    if (!path || !created_timestamp || n < 0 || (n > 0 && (!series_id || !dim_id || !ds || !quantity)))
        return -1;
    const size_t clen = std::strlen(created_timestamp);
    if (clen > 64) return -1;
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0) return TSF_CSV_E_OPEN;
    static const char header[] =
        "created_timestamp,series_id,dim_id,forecast_date,forecast_timestamp,forecast_quantity\n";
    const size_t hlen = sizeof(header) - 1;
    int hw = (int)std::thread::hardware_concurrency();
    if (hw < 1) hw = 1;
    const int nt = n_threads > 0 ? n_threads : (hw < 32 ? hw : 32);
    const size_t row_max = clen + 1 + 21 + 21 + 11 + 25 + 21 + 1;
    // Blocks of rows are formatted by the pool into buffers of their own (uninitialised: round 3 zero-filled
    // 16 buffers of 8.5 MB before the first row), their lengths prefix-summed, and every block then written at its
    // offset by the thread that holds it (pwrite) instead of one turn of 14 threads and a serial fwrite of 60 MB.
    // 16 384 rows per block for a file of a million rows; round 6: the scorer writes one part per chunk of models
    // (~370 000 rows), whose 22 blocks left a third of the pool idle -- blocks shrink to 4 096 rows so that a part has at
    // least ~3 per thread.
    const int64_t block = n >= ((int64_t)1 << 22) ? (1 << 14) : (1 << 12);
    const int64_t nb = (n + block - 1) / block;
    bool ok = true;
    try {
        std::vector<FileBuf> bufs((size_t)nb);
        std::vector<size_t> used((size_t)nb, 0);
        std::atomic<int> oom(0);
        run_workers(nt, nb, oom, [&](int64_t k) {
            const int64_t a = k * block, b = a + block < n ? a + block : n;
            if (!bufs[(size_t)k].alloc((size_t)(b - a) * row_max)) throw std::bad_alloc();
            char *p = bufs[(size_t)k].p;
            for (int64_t r = a; r < b; ++r) {
                std::memcpy(p, created_timestamp, clen);
                p += clen;
                *p++ = ',';
                p = put_int(p, series_id[r]);
                *p++ = ',';
                p = put_int(p, dim_id[r]);
                *p++ = ',';
                const int64_t v = ds[r];
                int64_t days = v / 86400000000000ll, rem = v % 86400000000000ll;
                if (rem < 0) {
                    rem += 86400000000000ll;
                    --days;
                }
                p = put_date(p, days);
                *p++ = ',';
                p = put_date(p, days);
                *p++ = 'T';
                const int64_t secs = rem / 1000000000ll;
                const unsigned ms = (unsigned)((rem % 1000000000ll) / 1000000ll);
                p = put2(p, (unsigned)(secs / 3600));
                *p++ = ':';
                p = put2(p, (unsigned)(secs / 60 % 60));
                *p++ = ':';
                p = put2(p, (unsigned)(secs % 60));
                *p++ = '.';
                *p++ = (char)('0' + ms / 100);
                p = put2(p, ms % 100);
                *p++ = 'Z';
                *p++ = ',';
                p = put_int(p, quantity[r]);
                *p++ = '\n';
            }
            used[(size_t)k] = (size_t)(p - bufs[(size_t)k].p);
        });
        if (oom.load()) { ::close(fd); return -2; }
        std::vector<size_t> at((size_t)nb + 1, hlen);
        for (int64_t k = 0; k < nb; ++k) at[(size_t)k + 1] = at[(size_t)k] + used[(size_t)k];
        auto put = [&](const char *q, size_t len, size_t off) {
            while (len > 0) {
                const ssize_t w = ::pwrite(fd, q, len, (off_t)off);
                if (w <= 0) return false;
                q += w; len -= (size_t)w; off += (size_t)w;
            }
            return true;
        };
        std::atomic<int> bad(0);
        if (!put(header, hlen, 0)) bad.store(1);
        run_workers(nt, nb, oom, [&](int64_t k) {
            if (!put(bufs[(size_t)k].p, used[(size_t)k], at[(size_t)k])) bad.store(1);
            bufs[(size_t)k].release();
        });
        ok = bad.load() == 0 && oom.load() == 0;
    } catch (...) {
        ::close(fd);
        return -2;
    }
    if (::close(fd) != 0) ok = false;
    return ok ? 0 : TSF_CSV_E_OPEN;
}

}  // namespace

extern "C" {

int tsf_csv_write_forecasts(const char *path, const char *created_timestamp, int64_t n,
                            const int64_t *series_id, const int64_t *dim_id, const int64_t *ds,
                            const int64_t *quantity, int32_t n_threads) {
    return write_forecasts<int64_t>(path, created_timestamp, n, series_id, dim_id, ds, quantity, n_threads);
}

int tsf_csv_write_forecasts_i32(const char *path, const char *created_timestamp, int64_t n,
                                const int32_t *series_id, const int32_t *dim_id, const int64_t *ds,
                                const int32_t *quantity, int32_t n_threads) {
    return write_forecasts<int32_t>(path, created_timestamp, n, series_id, dim_id, ds, quantity, n_threads);
}

}  // extern "C"
