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
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tsf.h"

namespace {

struct FileCols {
    std::vector<int64_t> sid, did, ds;
    std::vector<double> y;
    int err = 0;            // TSF_CSV_* code
    int64_t err_line = 0;   // 1-based
    int64_t malformed = 0;  // permissive mode: records that did not match the schema (dropped, counted)
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
                 bool permissive, FileCols &out) {
    size_t guess = (size_t)(end - p) / 24 + 1;
    out.did.reserve(guess);
    out.ds.reserve(guess);
    out.y.reserve(guess);
    out.sid.reserve(guess);
    int64_t line = 0;
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
        out.sid.push_back(sid);
        out.did.push_back(did);
        out.ds.push_back(ds);
        out.y.push_back(q);
        p = eol + 1;
    }
}

// gzip members (.gz: what Hadoop's GzipCodec writes and spark.read.csv decompresses transparently,
// prophet_modeler.py:109-114) and zlib streams (.deflate: DefaultCodec), recognised by their header bytes;
// several concatenated members are one file.  false = corrupt or truncated stream.
bool inflate_all(const std::vector<char> &in, std::vector<char> &out) {
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15 + 32) != Z_OK) return false;       // + 32: gzip or zlib header, detected
    out.clear();
    out.resize(in.size() * 6 + 4096);
    zs.next_in = (Bytef *)in.data();
    zs.avail_in = (uInt)in.size();
    size_t have = 0;
    bool ok = true;
    for (;;) {
        if (have == out.size()) out.resize(out.size() * 2);
        zs.next_out = (Bytef *)out.data() + have;
        zs.avail_out = (uInt)std::min<size_t>(out.size() - have, (size_t)1 << 30);
        const uInt before = zs.avail_out;
        const int rc = inflate(&zs, Z_NO_FLUSH);
        have += before - zs.avail_out;
        if (rc == Z_STREAM_END) {
            if (zs.avail_in == 0) break;
            if (inflateReset(&zs) != Z_OK) { ok = false; break; }   // next member
            continue;
        }
        if (rc != Z_OK) { ok = false; break; }
        if (zs.avail_in == 0 && zs.avail_out != 0) { ok = false; break; }      // truncated
    }
    inflateEnd(&zs);
    out.resize(have);
    return ok;
}

bool is_deflated(const std::vector<char> &b) {
    if (b.size() < 2) return false;
    const unsigned char b0 = (unsigned char)b[0], b1 = (unsigned char)b[1];
    if (b0 == 0x1f && b1 == 0x8b) return true;                                  // gzip
    return (b0 & 0x0f) == 8 && (b0 >> 4) <= 7 && ((b0 << 8) | b1) % 31 == 0 && b0 == 0x78;    // zlib, 32 K window
}

// the whole file with one open / fstat / read / close (10 000 part files of a Hive-partitioned input are
// 40 000 system calls instead of the 70 000 of fopen / fseek / ftell / fread / fclose)
bool read_whole(const char *path, std::vector<char> &buf, bool *corrupt) {
    *corrupt = false;
    const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat sb;
    if (::fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { ::close(fd); return false; }
    buf.resize((size_t)sb.st_size);
    size_t got = 0;
    while (got < buf.size()) {
        const ssize_t k = ::read(fd, buf.data() + got, buf.size() - got);
        if (k <= 0) break;
        got += (size_t)k;
    }
    ::close(fd);
    if (got != buf.size()) return false;
    if (is_deflated(buf)) {
        std::vector<char> raw;
        if (!inflate_all(buf, raw)) { *corrupt = true; return false; }
        buf.swap(raw);
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
    std::vector<FileCols> files;    // one entry per SEGMENT, in file order then byte order
    std::vector<int64_t> first;     // row offset of each segment
    int64_t n_rows = 0;
    int n_threads = 1;
};

namespace {

template <class F>
void run_workers(int n_threads, int64_t n_items, std::atomic<int> &oom, F item) {
    std::atomic<int64_t> next(0);
    auto worker = [&]() {
        for (;;) {
            int64_t i = next.fetch_add(1);
            if (i >= n_items) break;
            try {
                item(i);
            } catch (...) {
                oom.store(1);
            }
        }
    };
    if (n_threads <= 1 || n_items < 2) {
        worker();
        return;
    }
    std::vector<std::thread> th;
    int k = (int64_t)n_threads < n_items ? n_threads : (int)n_items;
    for (int i = 0; i < k; ++i) th.emplace_back(worker);
    for (auto &x : th) x.join();
}

}  // namespace

extern "C" {

int tsf_csv_read(int32_t n_files, const char *const *paths, const int64_t *series_id,
                 const char *layout, int32_t n_threads, tsf_csv **out, int64_t *n_rows,
                 int32_t *err_file, int64_t *err_line) {
    if (!out || n_files < 0 || (n_files > 0 && !paths) || !layout) return -1;
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
        // ---- phase 1: the files into memory (parallel)
        std::vector<std::vector<char>> bufs((size_t)n_files);
        std::vector<char> opened((size_t)n_files, 0);
        run_workers(t->n_threads, n_files, oom, [&](int64_t i) {
            bool corrupt = false;
            opened[(size_t)i] = read_whole(paths[i], bufs[(size_t)i], &corrupt) ? 1 : (corrupt ? 2 : 0);
        });
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
        // ---- segments: small files whole, big ones cut after a line end every SEGMENT_BYTES
        std::vector<Segment> segs;
        for (int32_t i = 0; i < n_files; ++i) {
            const std::vector<char> &b = bufs[(size_t)i];
            size_t at = 0;
            while (b.size() - at > SEGMENT_BYTES + SEGMENT_BYTES / 2) {
                const char *nl = (const char *)std::memchr(b.data() + at + SEGMENT_BYTES, '\n',
                                                           b.size() - at - SEGMENT_BYTES);
                if (!nl) break;
                size_t stop = (size_t)(nl - b.data()) + 1;
                segs.push_back(Segment{i, at, stop});
                at = stop;
            }
            segs.push_back(Segment{i, at, b.size()});
        }
        t->files.resize(segs.size());
        // ---- phase 2: parse (parallel over segments)
        run_workers(t->n_threads, (int64_t)segs.size(), oom, [&](int64_t k) {
            const Segment &sg = segs[(size_t)k];
            const char *base = bufs[(size_t)sg.file].data();
            parse_range(base + sg.begin, base + sg.end, layout, ncol,
                        series_id ? series_id[sg.file] : 0, permissive, t->files[(size_t)k]);
        });
        if (oom.load()) {
            delete t;
            return -2;
        }
        t->first.resize(segs.size() + 1);
        int64_t pos = 0;
        for (size_t k = 0; k < segs.size(); ++k) {
            const FileCols &fc = t->files[k];
            if (fc.err) {
                // line number inside the file = lines of the earlier segments + line in this one
                const Segment &sg = segs[k];
                const char *base = bufs[(size_t)sg.file].data();
                int64_t before = 0;
                for (const char *q = base; q < base + sg.begin; ++q) before += (*q == '\n');
                if (err_file) *err_file = sg.file;
                if (err_line) *err_line = before + fc.err_line;
                int e = fc.err;
                delete t;
                return e;
            }
            t->first[k] = pos;
            pos += (int64_t)fc.ds.size();
        }
        t->first[segs.size()] = pos;
        t->n_rows = pos;
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

int64_t tsf_csv_malformed(const tsf_csv *t) {
    if (!t) return -1;
    int64_t n = 0;
    for (const FileCols &fc : t->files) n += fc.malformed;
    return n;
}

int tsf_csv_fetch(tsf_csv *t, int64_t *series_id, int64_t *dim_id, int64_t *ds, double *y) {
    if (!t) return -1;
    const int32_t nf = (int32_t)t->files.size();
    std::atomic<int32_t> next(0);
    auto worker = [&]() {
        for (;;) {
            int32_t i = next.fetch_add(1);
            if (i >= nf) break;
            const FileCols &fc = t->files[(size_t)i];
            size_t n = fc.ds.size();
            int64_t at = t->first[(size_t)i];
            if (!n) continue;
            if (series_id) std::memcpy(series_id + at, fc.sid.data(), n * sizeof(int64_t));
            if (dim_id) std::memcpy(dim_id + at, fc.did.data(), n * sizeof(int64_t));
            if (ds) std::memcpy(ds + at, fc.ds.data(), n * sizeof(int64_t));
            if (y) std::memcpy(y + at, fc.y.data(), n * sizeof(double));
        }
    };
    if (t->n_threads <= 1 || nf < 2) {
        worker();
    } else {
        std::vector<std::thread> th;
        int k = t->n_threads < nf ? t->n_threads : nf;
        for (int i = 0; i < k; ++i) th.emplace_back(worker);
        for (auto &x : th) x.join();
    }
    return 0;
}

void tsf_csv_free(tsf_csv *t) { delete t; }

// ---- forecast sink ----------------------------------------------------------------------------
// ProphetScorer.convert_forecasts + write_forecasts (/root/reference/src/jobs/prophet_scorer.py:
// 130-150): one CSV with header
//   created_timestamp,series_id,dim_id,forecast_date,forecast_timestamp,forecast_quantity
// forecast_date = ds.date() as %Y-%m-%d (:107-108), forecast_timestamp in Spark 2.4's default
// CSV timestampFormat yyyy-MM-dd'T'HH:mm:ss.SSSXXX with the wall time taken as UTC.
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

}  // namespace

int tsf_csv_write_forecasts(const char *path, const char *created_timestamp, int64_t n,
                            const int64_t *series_id, const int64_t *dim_id, const int64_t *ds,
                            const int64_t *quantity, int32_t n_threads) {
    if (!path || !created_timestamp || n < 0 || (n > 0 && (!series_id || !dim_id || !ds || !quantity)))
        return -1;
    const size_t clen = std::strlen(created_timestamp);
    if (clen > 64) return -1;
    FILE *f = std::fopen(path, "wb");
    if (!f) return TSF_CSV_E_OPEN;
    static const char header[] =
        "created_timestamp,series_id,dim_id,forecast_date,forecast_timestamp,forecast_quantity\n";
    bool ok = std::fwrite(header, 1, sizeof(header) - 1, f) == sizeof(header) - 1;
    int hw = (int)std::thread::hardware_concurrency();
    if (hw < 1) hw = 1;
    int nt = n_threads > 0 ? n_threads : (hw < 16 ? hw : 16);
    const size_t row_max = clen + 1 + 21 + 21 + 11 + 25 + 21 + 1;
    const int64_t block = 1 << 16;                 // rows formatted per thread per turn
    try {
        std::vector<std::vector<char>> bufs((size_t)nt);
        std::vector<size_t> used((size_t)nt);
        for (auto &b : bufs) b.resize((size_t)block * row_max);
        for (int64_t base = 0; base < n && ok; base += block * nt) {
            auto work = [&](int t) {
                int64_t a = base + (int64_t)t * block, b = a + block < n ? a + block : n;
                char *p = bufs[(size_t)t].data();
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
                used[(size_t)t] = a < b ? (size_t)(p - bufs[(size_t)t].data()) : 0;
            };
            int live = 0;
            for (int t = 0; t < nt; ++t)
                if (base + (int64_t)t * block < n) live = t + 1;
            if (live <= 1) {
                work(0);
            } else {
                std::vector<std::thread> th;
                for (int t = 0; t < live; ++t) th.emplace_back(work, t);
                for (auto &x : th) x.join();
            }
            for (int t = 0; t < live && ok; ++t)
                ok = std::fwrite(bufs[(size_t)t].data(), 1, used[(size_t)t], f) == used[(size_t)t];
        }
    } catch (...) {
        std::fclose(f);
        return -2;
    }
    if (std::fclose(f) != 0) ok = false;
    return ok ? 0 : TSF_CSV_E_OPEN;
}

}  // extern "C"
