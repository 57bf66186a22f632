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
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
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
using gfm_tsv_detail::is_ws;
using gfm_tsv_detail::NameTable;
using gfm_tsv_detail::parse_rows;

namespace {

struct ErrSlot {
    ErrSlot &operator=(const std::string &m) { gfm_set_error_(m.c_str()); return *this; }
    ErrSlot &operator=(const char *m) { gfm_set_error_(m); return *this; }
} t_err;

}  // namespace

// ---------------------------------------------------------------------------------------------- file bytes
bool gfm_tsv_detail::FileBuf::load(const char *path, std::string &err)
{
    drop();
    int fd = open(path, O_RDONLY);
    if (fd < 0) { err = std::string("Unable to open ") + path; return false; }
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); err = std::string("Unable to stat ") + path; return false; }
    const size_t len = (size_t)sb.st_size;
    if (len == 0) { close(fd); return true; }
    constexpr size_t kReadLimit = (size_t)32 << 20;
    if (len <= kReadLimit) {
        if (buf_.size() < len) {
            buf_.clear();                       // nothing to carry over into a new allocation
            buf_.resize(len + len / 8);         // (files of one scan are about one size: no growth per file)
        }
        size_t got = 0;
        while (got < len) {
            const ssize_t r = read(fd, buf_.data() + got, len - got);
            if (r <= 0) break;
            got += (size_t)r;
        }
        close(fd);
        if (got != len) { err = std::string("Unable to read ") + path; return false; }
        p_ = buf_.data();
        len_ = len;
    } else {
        void *m = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { err = std::string("Unable to mmap ") + path; return false; }
        madvise(m, len, MADV_SEQUENTIAL);
        map_ = m;
        map_len_ = len;
        p_ = static_cast<const char *>(m);
        len_ = len;
    }
    return true;
}

void gfm_tsv_detail::FileBuf::drop()
{
    if (map_) munmap(map_, map_len_);
    map_ = nullptr;
    map_len_ = 0;
    p_ = nullptr;
    len_ = 0;
    // the parse threads belong to a crew that lives as long as the process: a buffer that grew for one big file
    // is handed back (region files are a few hundred KB: those keep theirs)
    if (buf_.capacity() > ((size_t)4 << 20)) std::vector<char>().swap(buf_);
}

#if defined(__x86_64__)
// Lines that hold a field, 64 bytes at a time: a line counts if a byte that is neither white space nor '\n' lies
// between two newlines (9 -> 3 ns per row; the per-line memchr + scan was a tenth of the scan's CPU time).
__attribute__((target("avx2,bmi"))) static int64_t count_rows_avx2(const char *p, const char *end)
{
    using namespace gfm_tsv_detail;
    int64_t n = 0;
    bool seen = false;                                   // the current line has shown a field byte
    for (; end - p >= 64; p += 64) {
        const __m256i v0 = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p));
        const __m256i v1 = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + 32));
        const __m256i nlv = _mm256_set1_epi8('\n');
        unsigned long long nl = (unsigned long long)(unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v0, nlv)) |
                                ((unsigned long long)(unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v1, nlv)) << 32);
        const unsigned long long ws = (unsigned long long)ws_mask32_avx2(v0) | ((unsigned long long)ws_mask32_avx2(v1) << 32);
        unsigned long long field = ~(ws | nl);           // bytes of fields
        while (nl) {
            const int t = __builtin_ctzll(nl);
            nl &= nl - 1;
            const unsigned long long upto = (1ull << t) - 1ull;          // bits below the newline (t <= 63)
            if (seen || (field & upto)) ++n;
            seen = false;
            field &= ~upto;                              // what is left belongs to the following lines
        }
        if (field) seen = true;
    }
    for (; p < end; ++p) {
        if (*p == '\n') { if (seen) ++n; seen = false; }
        else if (!is_ws(*p)) seen = true;
    }
    return n + (seen ? 1 : 0);
}

// The same with one 64-byte load and mask-register compares per block (3 -> 2 ns per row).
__attribute__((target("avx512f,avx512bw,bmi"))) static int64_t count_rows_avx512(const char *p, const char *end)
{
    using namespace gfm_tsv_detail;
    int64_t n = 0;
    bool seen = false;                                   // the current line has shown a field byte
    for (; end - p >= 64; p += 64) {
        unsigned long long nl, ws;
        masks64_avx512(p, &nl, &ws);
        unsigned long long field = ~(ws | nl);
        while (nl) {
            const int t = __builtin_ctzll(nl);
            nl &= nl - 1;
            const unsigned long long upto = (1ull << t) - 1ull;
            if (seen || (field & upto)) ++n;
            seen = false;
            field &= ~upto;
        }
        if (field) seen = true;
    }
    for (; p < end; ++p) {
        if (*p == '\n') { if (seen) ++n; seen = false; }
        else if (!is_ws(*p)) seen = true;
    }
    return n + (seen ? 1 : 0);
}
#endif

int64_t gfm_tsv_detail::count_rows(const char *p, const char *end, bool skip_rev)
{
#if defined(__x86_64__)
    if (!skip_rev && cpu_has_avx512()) return count_rows_avx512(p, end);
    if (!skip_rev && cpu_has_avx2()) return count_rows_avx2(p, end);
    const bool wide512 = cpu_has_avx512();
#endif
    int64_t n = 0;
    while (p < end) {
#if defined(__x86_64__)
        if (wide512 && end - p >= 128) {    // --no-reverse: the strand of a well-formed line from its white-space mask (76 -> 9 ns per kept row)
            const char *kb, *ke, *le;
            char sd = 0;
            const int r = light_line_avx512(p, end, &kb, &ke, &sd, &le);
            if (r >= 0) {
                if (r == 1 && sd != '-') ++n;
                p = le < end ? le + 1 : end;
                continue;
            }
        }
#endif
        const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
        const char *le = nl ? nl : end;
        const char *c = p;
        while (c < le && is_ws(*c)) ++c;
        if (c < le) {
            if (!skip_rev) {
                ++n;
            } else {           // the strand is the last character of the third field
                for (int f = 0; f < 2 && c < le; ++f) {
                    while (c < le && !is_ws(*c)) ++c;
                    while (c < le && is_ws(*c)) ++c;
                }
                while (c < le && !is_ws(*c)) ++c;
                if (c[-1] != '-') ++n;       // (a malformed line is counted: the parse pass reports it)
            }
        }
        p = nl ? nl + 1 : end;
    }
    return n;
}

void gfm_tsv_detail::parse_file(const char *path, int W, bool skip_rev, FileCols &out)
{
    static thread_local FileBuf text;
    struct Drop {
        FileBuf &b;
        ~Drop() { b.drop(); }
    } dropper{text};
    if (!text.load(path, out.error)) return;
    const size_t len = (size_t)(text.end() - text.begin());
    if (len == 0) return;
    const size_t guess = len / (size_t)(2 * W + 60) + 16;
    out.kmers.reserve(guess * (size_t)W);
    out.start.reserve(guess); out.stop.reserve(guess); out.freq.reserve(guess);
    out.strand.reserve(guess); out.is_ref.reserve(guess); out.local_name.reserve(guess);
    NameTable names(out.names);
    parse_rows(path, text.begin(), text.end(), W, skip_rev, names,
               [&](const uint8_t *kmer, int64_t st, int64_t sp, int64_t fr, uint8_t strand, uint8_t is_ref, int32_t nid) {
                   out.kmers.insert(out.kmers.end(), kmer, kmer + W);
                   out.start.push_back(st);
                   out.stop.push_back(sp);
                   out.freq.push_back(fr);
                   out.strand.push_back(strand);
                   out.is_ref.push_back(is_ref);
                   out.local_name.push_back(nid);
               },
               out.error);
}

// More threads than this lose to their own coordination: 2e6 rows in 1000 files (184 MB) on a 256-thread host
// through gfm_scan_tsv, median of 7 runs: 32 threads 6.6 ms, 64 5.6, 96 5.0, 128 5.6, 175 8.0 with 80 ms outliers
// (profiles/r02_scan_trace.txt).
static constexpr int kMaxParseThreads = 96;

// CPU time the container may use per second of wall time, in cores (cgroup v2 cpu.max, v1 cfs quota); 0: no quota.
// Read once.  A container that shows 256 hardware threads under a quota of 16 lets 64 threads burst for a few
// milliseconds, but work that needs more CPU time than one period's allowance (100 ms x the quota) is frozen as a whole
// once the allowance is used up, until the period ends (cpu.stat nr_throttled).
static double cpu_quota_cores()
{
    static const double cached = [] {
        double quota = 0, period = 0;
        if (FILE *fh = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            const int got = std::fscanf(fh, "%63s %lf", q, &period);
            std::fclose(fh);
            if (got == 2 && std::strcmp(q, "max") != 0 && period > 0) return std::atof(q) / period;
            if (got >= 1) return 0.0;
        }
        FILE *fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
        FILE *fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (fq && fp && std::fscanf(fq, "%lf", &quota) == 1 && std::fscanf(fp, "%lf", &period) == 1 && quota > 0 && period > 0) {
            std::fclose(fq);
            std::fclose(fp);
            return quota / period;
        }
        if (fq) std::fclose(fq);
        if (fp) std::fclose(fp);
        return 0.0;
    }();
    return cached;
}

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
#ifdef GFM_LAB
    if (const char *e = std::getenv("GRAFIMO_PARSE_THREADS_EXACT")) {   // no caps at all
        if (*e == '1') return nt;
    }
#endif
    if (nt > kMaxParseThreads) nt = kMaxParseThreads;
    // Big inputs under a CPU quota.  On the GPU boxes (256 hardware threads, quota 16 = 1.6 CPU-seconds per 100 ms) a
    // scan whose CPU time exceeds one period's allowance is frozen as a whole until the period ends, the sooner the more
    // threads it runs (a trace shows all workers asleep for 20-80 ms with nothing to wait for; profiles/r03_cpu_quota.txt),
    // and every thread beyond the quota makes the same work dearer (SMT siblings, cache sharing: 1.84 GB in 10 000 files cost
    // 0.58 CPU-s with 16 threads, 0.65 with 24, 0.75 with 32, 1.0 with 48, 1.37 with 64; profiles/r04_ingest_cpu.txt: wall
    // 44 / 38 / 34 / 32 / 36 ms rested, 44 / 46 / 34-61 / 75 / 95 ms back to back).  So: twice the quota for a scan that
    // finds the allowance untouched, one and a half times for one that starts within 150 ms of the one before it -- then
    // it is the allowance per period that bounds the throughput.  Without a quota the cap above stands (the bare read +
    // parse of those files scales to 96 threads: 28 ms, profiles/r03_file_read_scaling.txt).
    const double quota = cpu_quota_cores();
    if (quota > 0 && nt > (int)(4.0 * quota + 0.5)) nt = std::max(16, (int)(4.0 * quota + 0.5));   // (2e6 rows in 1000 files: 4.6 ms with 64 threads, 4.9 with 96)
    if (quota > 0 && bytes > (1ull << 30)) {
        static std::atomic<long long> last_big_ns{0};
        const long long now_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(
                                     std::chrono::steady_clock::now().time_since_epoch()).count();
        const long long prev = last_big_ns.exchange(now_ns);
        const bool sustained = prev != 0 && now_ns - prev < 150000000ll;
        const int cap = std::max(16, std::min(kMaxParseThreads, (int)((sustained ? 1.5 : 2.0) * quota + 0.5)));
        if (nt > cap) nt = cap;
    }
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
        total += f.rows();
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

GFM_API int gfm_tsv_count_rows(const char *path, int skip_reverse, int64_t *n_rows)
{
    if (!path || !n_rows) { t_err = "NULL argument"; return GFM_ERR_INVALID; }
    gfm_tsv_detail::FileBuf text;
    std::string err;
    if (!text.load(path, err)) { t_err = err; return GFM_ERR_IO; }
    *n_rows = gfm_tsv_detail::count_rows(text.begin(), text.end(), skip_reverse != 0);
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

