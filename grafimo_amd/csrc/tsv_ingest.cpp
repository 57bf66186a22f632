// tsv_ingest.cpp -- host-side parser of vg's k-mer TSV rows into columnar arrays.
//
// Replaces the text handling of score_seqs (score_sequences.py:273-293, :305-307, paths
// relative to /root/reference/src/grafimo/): one row per haplotype k-mer,
//   REGION \t KMER \t CHR:START(+|-) \t CHR:STOP(+|-) \t COUNT \t ref|non.ref \t NODEPATH
// split on whitespace; strand = last char of column 3; start/stop = the integer after the
// first ':' of columns 3/4 minus the strand char; '-' rows dropped before counting when
// skip_reverse; "ref" rows whose |stop-start| != W become "non.ref".
// Files are mmap'ed and parsed by a small pool of host threads (one file at a time each).

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "gfm_tsv_internal.hpp"
#include "gfm_workers.hpp"

#define GFM_API extern "C" __attribute__((visibility("default")))

using gfm_tsv_detail::FileCols;

namespace {

struct ErrSlot {
    ErrSlot &operator=(const std::string &m) { gfm_set_error_(m.c_str()); return *this; }
    ErrSlot &operator=(const char *m) { gfm_set_error_(m); return *this; }
} t_err;

inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

// CHR:NUM(+|-) -> NUM and strand; the reference takes split(":")[1] and drops its last char
bool parse_pos(const char *b, const char *e, int64_t *val, char *strand)
{
    if (e - b < 3) return false;
    const char *colon = static_cast<const char *>(memchr(b, ':', (size_t)(e - b)));
    if (!colon) return false;
    const char *p = colon + 1;
    const char *q = static_cast<const char *>(memchr(p, ':', (size_t)(e - p)));
    const char *fe = q ? q : e;      // field after the first ':' (up to a second ':', if any)
    if (fe - p < 2) return false;
    *strand = e[-1];                 // data[2][-1]: last char of the whole column
    const char *ne = fe - 1;         // [:-1]
    bool neg = false;
    if (p < ne && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
    if (p >= ne) return false;
    int64_t v = 0;
    for (; p < ne; ++p) {
        if (*p < '0' || *p > '9') return false;
        v = v * 10 + (*p - '0');
    }
    *val = neg ? -v : v;
    return true;
}

bool parse_int(const char *b, const char *e, int64_t *val)
{
    if (b >= e) return false;
    bool neg = false;
    if (*b == '-' || *b == '+') { neg = *b == '-'; ++b; }
    if (b >= e) return false;
    int64_t v = 0;
    for (; b < e; ++b) {
        if (*b < '0' || *b > '9') return false;
        v = v * 10 + (*b - '0');
    }
    *val = neg ? -v : v;
    return true;
}

}  // namespace

void gfm_tsv_detail::parse_file(const char *path, int W, bool skip_rev, FileCols &out)
{
    int fd = open(path, O_RDONLY);
    if (fd < 0) { out.error = std::string("Unable to open ") + path; return; }
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); out.error = std::string("Unable to stat ") + path; return; }
    const size_t len = (size_t)sb.st_size;
    if (len == 0) { close(fd); return; }
    // Small files are read into a per-thread buffer: with hundreds of parse threads, mmap/munmap of
    // thousands of region files serialise on the process's address-space lock.  Big files are mapped.
    constexpr size_t kReadLimit = (size_t)32 << 20;
    static thread_local std::vector<char> t_buf;
    // the parse threads belong to a crew that lives as long as the process: a buffer that grew for one big file is
    // handed back when this file is done (region files are a few hundred KB: those keep their buffer)
    struct Shrink {
        std::vector<char> &b;
        ~Shrink() { if (b.capacity() > ((size_t)1 << 20)) std::vector<char>().swap(b); }
    } shrink{t_buf};
    void *map = nullptr;
    const char *p = nullptr;
    if (len <= kReadLimit) {
        t_buf.resize(len);
        size_t got = 0;
        while (got < len) {
            const ssize_t r = read(fd, t_buf.data() + got, len - got);
            if (r <= 0) break;
            got += (size_t)r;
        }
        close(fd);
        if (got != len) { out.error = std::string("Unable to read ") + path; return; }
        p = t_buf.data();
    } else {
        map = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (map == MAP_FAILED) { out.error = std::string("Unable to mmap ") + path; return; }
        madvise(map, len, MADV_SEQUENTIAL);
        p = static_cast<const char *>(map);
    }
    const char *end = p + len;
    const size_t guess = len / (size_t)(2 * W + 60) + 16;
    out.kmers.reserve(guess * (size_t)W);
    out.start.reserve(guess); out.stop.reserve(guess); out.freq.reserve(guess);
    out.strand.reserve(guess); out.is_ref.reserve(guess); out.local_name.reserve(guess);
    std::unordered_map<std::string, int32_t> name_ix;
    const char *last_name = nullptr;
    size_t last_len = 0;
    int32_t last_id = -1;
    int64_t lineno = 0;
    while (p < end) {
        const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
        const char *le = nl ? nl : end;
        ++lineno;
        // split the first six whitespace-separated fields
        const char *fb[6], *fe[6];
        int nf = 0;
        const char *c = p;
        while (c < le && nf < 6) {
            while (c < le && is_ws(*c)) ++c;
            if (c >= le) break;
            fb[nf] = c;
            while (c < le && !is_ws(*c)) ++c;
            fe[nf] = c;
            ++nf;
        }
        const char *next = nl ? nl + 1 : end;
        if (nf == 0) { p = next; continue; }  // blank line
        auto bad = [&](const char *what) {
            char buf[256];
            snprintf(buf, sizeof buf, "%s:%lld: %s", path, (long long)lineno, what);
            out.error = buf;
        };
        if (nf < 6) { bad("expected at least 6 columns"); break; }
        int64_t st = 0, sp = 0, fr = 0;
        char s1 = 0, s2 = 0;
        if (!parse_pos(fb[2], fe[2], &st, &s1)) { bad("malformed start column"); break; }
        if (skip_rev && s1 == '-') { p = next; continue; }
        if (!parse_pos(fb[3], fe[3], &sp, &s2)) { bad("malformed stop column"); break; }
        if (fe[1] - fb[1] != W) { bad("k-mer length differs from the motif width"); break; }
        if (!parse_int(fb[4], fe[4], &fr)) { bad("malformed haplotype count"); break; }
        const size_t nlen = (size_t)(fe[0] - fb[0]);
        int32_t nid;
        if (last_name && nlen == last_len && memcmp(last_name, fb[0], nlen) == 0) {
            nid = last_id;
        } else {
            std::string key(fb[0], nlen);
            auto it = name_ix.find(key);
            if (it == name_ix.end()) {
                nid = (int32_t)out.names.size();
                out.names.push_back(key);
                name_ix.emplace(std::move(key), nid);
            } else {
                nid = it->second;
            }
            last_name = fb[0]; last_len = nlen; last_id = nid;
        }
        out.kmers.insert(out.kmers.end(), reinterpret_cast<const uint8_t *>(fb[1]),
                         reinterpret_cast<const uint8_t *>(fe[1]));
        out.start.push_back(st);
        out.stop.push_back(sp);
        out.freq.push_back(fr);
        out.strand.push_back((uint8_t)s1);
        const bool is_ref_str = (fe[5] - fb[5] == 3) && memcmp(fb[5], "ref", 3) == 0;
        const int64_t dist = sp > st ? sp - st : st - sp;
        out.is_ref.push_back((uint8_t)(is_ref_str && dist == W));  // score_sequences.py:305-307
        out.local_name.push_back(nid);
        p = next;
    }
    if (map) munmap(map, len);
}

// More threads than this lose to their own coordination: 2e6 rows in 1000 files (184 MB) on a 256-thread host
// through gfm_scan_tsv, median of 7 runs: 32 threads 6.6 ms, 64 5.6, 96 5.0, 128 5.6, 175 8.0 with 80 ms outliers
// (profiles/r02_scan_trace.txt).
static constexpr int kMaxParseThreads = 96;

int gfm_tsv_detail::pick_threads(const char *const *paths, int n_paths, int requested)
{
    int nt = requested > 0 ? requested : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > n_paths) nt = n_paths;
    if (nt <= 1) return 1;
    // total size from at most 64 files spread over the list (a stat per file was 0.8 ms for 1000 files)
    unsigned long long bytes = 0;
    const int step = (n_paths + 63) / 64;
    int sampled = 0;
    for (int i = 0; i < n_paths; i += step, ++sampled) {
        struct stat sb;
        if (stat(paths[i], &sb) == 0) bytes += (unsigned long long)sb.st_size;
    }
    bytes = bytes / (unsigned long long)sampled * (unsigned long long)n_paths;
    const unsigned long long by_size = bytes >> 20;
    if ((unsigned long long)nt > by_size) nt = (int)(by_size < 1 ? 1 : by_size);
    if (nt > kMaxParseThreads) nt = kMaxParseThreads;
    return nt;
}

void gfm_tsv::index_rows()
{
    std::unordered_map<std::string, int32_t> gix;
    const size_t n_paths = files.size();
    row_base.resize(n_paths);
    remap.resize(n_paths);
    names.clear();
    int64_t total = 0;
    for (size_t i = 0; i < n_paths; ++i) {
        FileCols &f = files[i];
        row_base[i] = total;
        total += (int64_t)f.start.size();
        auto &rm = remap[i];
        rm.resize(f.names.size());
        for (size_t k = 0; k < f.names.size(); ++k) {
            auto it = gix.find(f.names[k]);
            if (it == gix.end()) {
                rm[k] = (int32_t)names.size();
                gix.emplace(f.names[k], rm[k]);
                names.push_back(f.names[k]);
            } else {
                rm[k] = it->second;
            }
        }
    }
    n = total;
}

GFM_API int gfm_tsv_open(const char *const *paths, int n_paths, int width, int skip_reverse,
                         int n_threads, gfm_tsv_t *out, int64_t *n_rows)
{
    if (!out || !n_rows || (n_paths > 0 && !paths)) { t_err = "NULL argument"; return GFM_ERR_INVALID; }
    *out = nullptr;
    *n_rows = 0;
    if (width < 1 || n_paths < 0) { t_err = "bad width or path count"; return GFM_ERR_INVALID; }
    gfm_tsv *t = new (std::nothrow) gfm_tsv();
    if (!t) { t_err = "out of memory"; return GFM_ERR_NOMEM; }
    t->W = width;
    t->files.resize((size_t)n_paths);
    const int nt = gfm_tsv_detail::pick_threads(paths, n_paths, n_threads);
    std::atomic<int> next{0};
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n_paths) break;
            try {
                gfm_tsv_detail::parse_file(paths[i], width, skip_reverse != 0, t->files[(size_t)i]);
            } catch (const std::bad_alloc &) {
                t->files[(size_t)i].error = "out of memory";
            }
        }
    };
    gfm_workers::run(nt, work);
    for (auto &f : t->files)
        if (!f.error.empty()) {
            t_err = f.error;
            delete t;
            return GFM_ERR_IO;
        }
    t->index_rows();
    *n_rows = t->n;
    *out = t;
    return GFM_OK;
}

GFM_API int gfm_tsv_read(gfm_tsv_t t, uint8_t *kmers, int64_t *start, int64_t *stop, uint8_t *strand,
                         int64_t *freq, uint8_t *is_ref, int32_t *file_id, int32_t *name_id)
{
    if (!t) { t_err = "NULL handle"; return GFM_ERR_INVALID; }
    for (size_t i = 0; i < t->files.size(); ++i) {
        const FileCols &f = t->files[i];
        const int64_t b = t->row_base[i];
        const size_t m = f.start.size();
        if (!m) continue;
        if (kmers) memcpy(kmers + (size_t)b * (size_t)t->W, f.kmers.data(), m * (size_t)t->W);
        if (start) memcpy(start + b, f.start.data(), m * sizeof(int64_t));
        if (stop) memcpy(stop + b, f.stop.data(), m * sizeof(int64_t));
        if (freq) memcpy(freq + b, f.freq.data(), m * sizeof(int64_t));
        if (strand) memcpy(strand + b, f.strand.data(), m);
        if (is_ref) memcpy(is_ref + b, f.is_ref.data(), m);
        if (file_id) for (size_t k = 0; k < m; ++k) file_id[b + (int64_t)k] = (int32_t)i;
        if (name_id) {
            const auto &rm = t->remap[i];
            for (size_t k = 0; k < m; ++k) name_id[b + (int64_t)k] = rm[(size_t)f.local_name[k]];
        }
    }
    return GFM_OK;
}

GFM_API int gfm_tsv_name_count(gfm_tsv_t t) { return t ? (int)t->names.size() : 0; }

GFM_API int64_t gfm_tsv_names_bytes(gfm_tsv_t t)
{
    if (!t) return 0;
    int64_t s = 0;
    for (const auto &x : t->names) s += (int64_t)x.size();
    return s;
}

GFM_API int gfm_tsv_names(gfm_tsv_t t, int64_t *offsets, char *bytes)
{
    if (!t || !offsets || (!bytes && gfm_tsv_names_bytes(t))) { t_err = "NULL argument"; return GFM_ERR_INVALID; }
    int64_t o = 0;
    for (size_t i = 0; i < t->names.size(); ++i) {
        offsets[i] = o;
        memcpy(bytes + o, t->names[i].data(), t->names[i].size());
        o += (int64_t)t->names[i].size();
    }
    offsets[t->names.size()] = o;
    return GFM_OK;
}

GFM_API void gfm_tsv_close(gfm_tsv_t t) { delete t; }

