// scan_stream.cpp -- compute_results' numeric core as ONE streamed pass (score_sequences.py:113-157,194-205):
//   TSV parse threads -> pinned chunk buffers -> hipMemcpyAsync on a copy stream -> score kernel per chunk
//   (one histogram and one hit list per motif, row ids global) -> q-value table + selection once at the end -> the
//   hits, with the columns of their rows, back to the host.
// The reference forks `cores` workers that parse and score line by line and then merges pickled lists; here
// the parse threads run ahead of the GPU and nothing but the hits ever comes back.  Buffers (pinned chunk
// slots, device slots, score blocks, tables, hit lists, the host columns of the rows) live in a per-device pool that
// only grows: a second call of the same size allocates nothing -- and touches no fresh page: at 2e7 rows the page
// faults of freshly allocated per-file columns (1.4 GB, 96 threads in one address space) were most of the scan.
//
// Two phases, so that a sharded scan can put its collective between them (distributed.py): gfm_scan_tsv_begin
// returns when the last chunk is scored, with every motif's histogram complete on the device; the caller may
// all-reduce the histograms (its own buffers, if it passed some); gfm_scan_tsv_finish derives q-tables and cutoffs
// from them, selects and brings the hits back.  gfm_scan_tsv is the two in one call.  Several motifs of ONE width
// share the pass: each chunk is scored by gfm_score_kmers_multi (one read of the k-mers per group of up to three).
//
// Reference lines cited as file:line are relative to /root/reference/src/grafimo/.

#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "gfm_hit_sort.hpp"
#include "gfm_tsv_internal.hpp"
#include "gfm_workers.hpp"

#define GFM_API extern "C" __attribute__((visibility("default")))

using gfm_tsv_detail::FileCols;

namespace {
// Phase times to stderr: lab builds only (scripts/lab_build.sh -DGFM_LAB); the product reads no timing knob.
inline bool scan_trace_on()
{
#ifdef GFM_LAB
    static const bool on = std::getenv("GRAFIMO_SCAN_TRACE") != nullptr;
    return on;
#else
    return false;
#endif
}
}   // namespace

namespace {

int sfail(int code, const std::string &msg)
{
    gfm_set_error_(msg.c_str());
    return code;
}

#define S_TRY(expr)                                                                              \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return sfail(GFM_ERR_HIP, std::string(#expr " failed: ") + hipGetErrorString(e_));   \
    } while (0)
#define S_RC(expr)              \
    do {                        \
        const int rc_ = (expr); \
        if (rc_) return rc_;    \
    } while (0)

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

constexpr int kSlots = 3;                          // chunk slots: one being staged, one in flight, one spare
constexpr int64_t kDefaultChunkRows = 1 << 20;     // rows per chunk (x W bytes pinned + device, per slot)

// What the host keeps of the rows of one chunk: where each row's line starts in its file (8 bytes per row).  The
// columns of a row -- coordinates, haplotype count, ref flag, REGION -- are read from there for the hit rows only, at
// gfm_scan_tsv_finish (round 3 kept 30 + W bytes of parsed columns per row: 1 GB of stores for 2e7 rows of which two
// thousand were ever looked at).  One allocation, kept in the pool.
struct MetaChunk {
    void *block = nullptr;
    uint64_t *line_off = nullptr;
    bool alloc(int64_t rows, int)
    {
        // (2 MiB aligned and advised to use huge pages, like the text arena: the hit rows' offsets are picked out of these
        // blocks at random)
        const size_t bytes = (((size_t)rows * sizeof(uint64_t) + 64) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        if (posix_memalign(&block, (size_t)2 << 20, bytes) != 0) { block = nullptr; return false; }
#ifdef MADV_HUGEPAGE
        (void)::madvise(block, bytes, MADV_HUGEPAGE);
#endif
        line_off = static_cast<uint64_t *>(block);
        return true;
    }
    void release() { std::free(block); *this = MetaChunk(); }
};

// per-motif device buffers of a scan
struct MotifBufs {
    uint64_t *d_hist = nullptr;         // the library's own histogram (a caller may pass its own instead)
    double *d_q = nullptr;
    size_t table_len = 0;
    int32_t *d_cutoff = nullptr;
    uint64_t *d_count = nullptr;
    int64_t *d_hits = nullptr;
    uint64_t *d_cand_count = nullptr;   // q-value threshold: the p < t candidates collected while scoring
    int64_t *d_cand = nullptr;          // (same capacity as d_hits)
    int64_t hit_cap = 0;
    int init()
    {
        S_TRY(hipMalloc(&d_cutoff, sizeof(int32_t)));
        S_TRY(hipMalloc(&d_count, sizeof(uint64_t)));
        S_TRY(hipMalloc(&d_cand_count, sizeof(uint64_t)));
        return GFM_OK;
    }
    int reserve_tables(size_t len)
    {
        if (len <= table_len) return GFM_OK;
        if (d_hist) (void)hipFree(d_hist);
        if (d_q) (void)hipFree(d_q);
        d_hist = nullptr;
        d_q = nullptr;
        table_len = 0;
        S_TRY(hipMalloc(&d_hist, sizeof(uint64_t) * len));
        S_TRY(hipMalloc(&d_q, sizeof(double) * len));
        table_len = len;
        return GFM_OK;
    }
    // a list that was grown for a scan that has to be run again (too_short at finish): the capacity the retry needs.  trim()
    // keeps such a list whatever its size -- the retry closes the scan (-> trim) before it begins the next one (with
    // the list dropped there, the second begin reserved the default again and a scan of more than keep_bytes / 16 passing rows
    // overflowed on every attempt) -- and the next begin reserves at least this much; a finish that holds clears it.
    int64_t want_cap = 0;
    void release_hits()
    {
        if (d_hits) (void)hipFree(d_hits);
        if (d_cand) (void)hipFree(d_cand);
        d_hits = d_cand = nullptr;
        hit_cap = 0;
    }
    int reserve_hits(int64_t cap)
    {
        if (cap <= hit_cap) return GFM_OK;
        if (d_hits) (void)hipFree(d_hits);
        if (d_cand) (void)hipFree(d_cand);
        d_hits = d_cand = nullptr;
        hit_cap = 0;
        S_TRY(hipMalloc(&d_hits, sizeof(int64_t) * (size_t)cap));
        S_TRY(hipMalloc(&d_cand, sizeof(int64_t) * (size_t)cap));
        hit_cap = cap;
        return GFM_OK;
    }
    void release()
    {
        if (d_hist) (void)hipFree(d_hist);
        if (d_q) (void)hipFree(d_q);
        if (d_cutoff) (void)hipFree(d_cutoff);
        if (d_count) (void)hipFree(d_count);
        if (d_hits) (void)hipFree(d_hits);
        if (d_cand_count) (void)hipFree(d_cand_count);
        if (d_cand) (void)hipFree(d_cand);
        *this = MotifBufs();
    }
};

// Per-device buffers of the streamed scan; they only grow.
struct ScanPool {
    int device = -1;
    hipStream_t copy = nullptr, score = nullptr;
    hipEvent_t copied[kSlots] = {}, scored[kSlots] = {};
    hipEvent_t c0[kSlots] = {}, c1[kSlots] = {};   // H2D timing
    uint8_t *h_pin[kSlots] = {};
    uint8_t *d_kmers[kSlots] = {};
    size_t slot_bytes = 0;
    std::vector<MotifBufs> mb;              // one per motif of the scan
    std::vector<MetaChunk> meta;            // one per chunk index
    // The TEXT of the scanned files, kept until the scan closes: the files are read straight into this arena (no per-thread
    // buffer, no copy), and the hit rows' columns are parsed from it at finish -- at the bench's hit density (1 % of the
    // rows are planted motif instances) reading the lines back from the files was 193 000 pread()s and 10 000 open()s for
    // a 2e7-row scan: 22 ms of a 73 ms scan, under a CPU quota that the scan's own work already exhausts.  Grow-only like
    // the slots; files that do not fit (or directories beyond kTextMax) go through the threads' own buffers and their
    // hit rows are read back from the files.
    char *text = nullptr;
    size_t text_cap = 0, text_map = 0;
    int64_t meta_rows = 0;                  // rows (and width) the column blocks were made for
    int meta_W = 0;
    bool in_use = false;

    int init(int dev)
    {
        device = dev;
        S_TRY(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
        S_TRY(hipStreamCreateWithFlags(&score, hipStreamNonBlocking));
        for (int s = 0; s < kSlots; ++s) {
            S_TRY(hipEventCreateWithFlags(&copied[s], hipEventDisableTiming));
            S_TRY(hipEventCreateWithFlags(&scored[s], hipEventDisableTiming));
            S_TRY(hipEventCreate(&c0[s]));
            S_TRY(hipEventCreate(&c1[s]));
        }
        return GFM_OK;
    }
    int reserve_motifs(size_t n)
    {
        while (mb.size() < n) {
            mb.emplace_back();
            S_RC(mb.back().init());
        }
        return GFM_OK;
    }
    int reserve_slots(size_t bytes)
    {
        if (bytes <= slot_bytes) return GFM_OK;
        for (int s = 0; s < kSlots; ++s) {
            if (h_pin[s]) (void)hipHostFree(h_pin[s]);
            if (d_kmers[s]) (void)hipFree(d_kmers[s]);
            h_pin[s] = nullptr;
            d_kmers[s] = nullptr;
        }
        slot_bytes = 0;
        for (int s = 0; s < kSlots; ++s) {
            S_TRY(hipHostMalloc(reinterpret_cast<void **>(&h_pin[s]), bytes, hipHostMallocDefault));
            S_TRY(hipMalloc(&d_kmers[s], bytes));
        }
        slot_bytes = bytes;
        return GFM_OK;
    }
    // column block of chunk k (called by the parse threads under the scan's mutex)
    MetaChunk *meta_chunk(size_t k, int64_t rows, int W)
    {
        if (rows > meta_rows || W > meta_W) {
            for (auto &m : meta) m.release();
            meta.clear();
            meta_rows = std::max(rows, meta_rows);
            meta_W = std::max(W, meta_W);
        }
        while (meta.size() <= k) {
            meta.emplace_back();
            if (!meta.back().alloc(meta_rows, meta_W)) { meta.pop_back(); return nullptr; }
        }
        return &meta[k];
    }
    // What a scan holds while it runs: three chunk slots on each side, hit lists, per row 8 bytes of host memory (line
    // offsets) and -- up to kTextMax -- the files' text (the hit rows' columns are parsed from it at finish; files beyond it are
    // read back with pread).  NO score is stored (round 5: the cutoff is known before scoring, hits and histogram come out of
    // the score kernel; rounds 1-4 kept one int32 block per chunk and motif, 80 MB per 2e7 rows, read only by a fallback).
    // What a CLOSED scan leaves behind is bounded by keep_bytes() per side (GRAFIMO_SCAN_KEEP_BYTES, default 256 MiB): a
    // next scan of that size is allocation-free, a bigger one maps its arena again (huge pages, touched by all workers).
    static constexpr size_t kTextMax = (size_t)4 << 30;
    static size_t keep_bytes()
    {
        static const size_t v = [] {
            const char *e = std::getenv("GRAFIMO_SCAN_KEEP_BYTES");
            return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)256 << 20;
        }();
        return v;
    }
    // The text arena is its own mapping, advised to use huge pages: 1.8 GB of 4 KiB pages is 450 000 TLB entries'
    // worth of text that is written once by read(), read once by the scan and then picked at by the hit rows -- each
    // of those a page walk (the hit rows' columns: 208 -> 130 ms of CPU at 193 000 hits once the walks were gone).
    void free_text()
    {
        if (text) (void)::munmap(text, text_map);
        text = nullptr;
        text_cap = text_map = 0;
    }
    int reserve_text(size_t bytes)
    {
        if (bytes <= text_cap) return GFM_OK;
        free_text();
        if (bytes == 0) return GFM_OK;
        const size_t huge = (size_t)2 << 20;
        const size_t len = (bytes + huge - 1) / huge * huge;
        void *p = ::mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) return sfail(GFM_ERR_NOMEM, "out of host memory");
#ifdef MADV_HUGEPAGE
        (void)::madvise(p, len, MADV_HUGEPAGE);      // (refused where the kernel has none: small pages then, as before)
#endif
        text = static_cast<char *>(p);
        text_map = len;
        text_cap = bytes;
        return GFM_OK;
    }
    void trim()
    {
        const size_t keep = keep_bytes();
        if (text_cap > keep) free_text();
        const size_t per_meta = (size_t)meta_rows * sizeof(uint64_t) + 64;
        while (!meta.empty() && meta.size() * per_meta + text_cap > keep) {
            meta.back().release();
            meta.pop_back();
        }
        for (auto &m : mb)          // hit lists that an overflow made large: back to the default next time
            if ((size_t)m.hit_cap * 2 * sizeof(int64_t) > keep && m.want_cap == 0) m.release_hits();
    }
    void release()
    {
        for (int s = 0; s < kSlots; ++s) {
            if (h_pin[s]) (void)hipHostFree(h_pin[s]);
            if (d_kmers[s]) (void)hipFree(d_kmers[s]);
            if (copied[s]) (void)hipEventDestroy(copied[s]);
            if (scored[s]) (void)hipEventDestroy(scored[s]);
            if (c0[s]) (void)hipEventDestroy(c0[s]);
            if (c1[s]) (void)hipEventDestroy(c1[s]);
            h_pin[s] = d_kmers[s] = nullptr;
            copied[s] = scored[s] = c0[s] = c1[s] = nullptr;
        }
        for (auto &m : mb) m.release();
        mb.clear();
        for (auto &m : meta) m.release();
        meta.clear();
        free_text();
        if (copy) (void)hipStreamDestroy(copy);
        if (score) (void)hipStreamDestroy(score);
        *this = ScanPool();
    }
};

std::mutex g_pool_mu;
std::vector<ScanPool *> g_pools;   // one per device that was used

// the calling thread's pool for the current device (callers are single-threaded per device; a second
// concurrent scan on one device gets a pool of its own)
int acquire_pool(ScanPool **out)
{
    int dev = 0;
    S_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto *p : g_pools)
        if (p->device == dev && !p->in_use) {
            p->in_use = true;
            *out = p;
            return GFM_OK;
        }
    ScanPool *p = new (std::nothrow) ScanPool();
    if (!p) return sfail(GFM_ERR_NOMEM, "out of host memory");
    const int rc = p->init(dev);
    if (rc) {
        p->release();
        delete p;
        return rc;
    }
    p->in_use = true;
    g_pools.push_back(p);
    *out = p;
    return GFM_OK;
}

void release_pool(ScanPool *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    p->trim();
    p->in_use = false;
}

struct MotifHits {
    std::vector<int64_t> rows;        // hit rows, ascending (global row ids in sorted-file order)
    std::vector<int32_t> scaled;
    std::vector<double> logodds, pvalue, qvalue;
    // the columns of the hit rows' TSV lines (read back from the files at finish)
    std::vector<uint8_t> kmers, strand, is_ref;
    std::vector<int64_t> start, stop, freq;
    std::vector<int32_t> name_id;
};

}  // namespace

struct gfm_scan {
    gfm_tsv table;                    // per file: row count and REGION names (the rows' columns live in the pool)
    int W = 0, L = 0;
    bool have_q = false;
    std::vector<MotifHits> hits;      // one per motif
    std::vector<std::string> paths;   // the files, in scan order (the hit rows' lines are read from them at finish)
    struct Kept { const char *p = nullptr; size_t len = 0; };
    std::vector<Kept> kept;           // per file: its text in the pool's arena (nullptr: not kept, read it back from the file)
    int n_threads = 1;
    gfm_scan_stats_t stats{};
    // ---- what gfm_scan_tsv_finish needs from gfm_scan_tsv_begin
    std::vector<gfm_motif_t> motifs;
    std::vector<uint64_t *> d_hist;   // per motif: the caller's buffer or the pool's
    std::vector<int32_t> cutoffs;     // p-value cutoffs (known before scoring)
    ScanPool *pool = nullptr;         // leased from begin to close
    double threshold = 0.0;
    bool on_qvalue = false, want_qvalues = false, finished = false;
    int64_t chunk_rows = 0, total_rows = 0;
    size_t n_chunks = 0;
    std::vector<int64_t> chunk_n;     // rows of every submitted chunk
    double t_begin = 0.0, t_parsed = 0.0, begin_s = 0.0;
    ~gfm_scan() { release_pool(pool); }
};

namespace {

// The columns of the hit rows, read back from their files (score_sequences.py:285-293, :305-307 for the rows that are
// reported).  Hits ascend by row, rows by file: per motif and file ONE job -- open, then either a pread of a kilobyte
// per hit (the usual case: one hit in ten thousand rows) or, when a file holds so many hits that this would read most
// of it anyway, the whole file once; the jobs are spread over the crew.  REGION strings become ids in the scan's name
// table (only names of hit rows are in it).
struct HitJob {
    size_t motif, file, i0, i1;               // hits [i0, i1) of the motif lie in this file
    std::vector<std::string> names;           // distinct REGION strings of the job's rows, local ids in name_id[]
};

int fetch_hit_columns(gfm_scan *sc)
{
    const gfm_tsv &t = sc->table;
    const ScanPool *P = sc->pool;
    const size_t W = (size_t)sc->W;
    std::vector<HitJob> jobs;
    for (size_t j = 0; j < sc->hits.size(); ++j) {
        MotifHits &h = sc->hits[j];
        const size_t k = h.rows.size();
        h.kmers.resize(k * W);
        h.start.resize(k); h.stop.resize(k); h.freq.resize(k);
        h.strand.resize(k); h.is_ref.resize(k); h.name_id.resize(k);
        size_t fi = 0;
        for (size_t i = 0; i < k;) {
            while (fi + 1 < t.files.size() && t.row_base[fi + 1] <= h.rows[i]) ++fi;
            const int64_t file_end = fi + 1 < t.files.size() ? t.row_base[fi + 1] : t.n;
            size_t e = i;
            while (e < k && h.rows[e] < file_end) ++e;
            jobs.push_back(HitJob{j, fi, i, e, {}});
            i = e;
        }
    }
    if (jobs.empty()) return GFM_OK;
    const bool trace = scan_trace_on();
    const double t_jobs = trace ? now_s() : 0.0;
    std::atomic<int64_t> cpu_offsets{0}, cpu_lines{0};
    auto tcpu = []() -> int64_t {
        timespec ts{};
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
        return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
    };
    std::atomic<size_t> next{0};
    const int n_workers = std::max(1, std::min<int>(sc->n_threads, (int)((jobs.size() + 3) / 4)));
    const size_t job_run = std::max<size_t>(1, std::min<size_t>(64, jobs.size() / ((size_t)n_workers * 4)));
    std::mutex err_mu;
    std::string err;
    auto fail_job = [&](const std::string &msg) {
        std::lock_guard<std::mutex> lk(err_mu);
        if (err.empty()) err = msg;
    };
    std::atomic<int64_t> w_first{INT64_MAX}, w_last_start{0}, w_last_end{0};      // trace: when the workers got going (ns after t_run)
    std::atomic<int> w_with_work{0};
    auto work = [&]() {
        static thread_local gfm_tsv_detail::FileBuf whole;
        std::vector<char> buf;
        size_t jx_at = 0, jx_end = 0;
        const int64_t w_t0 = trace ? (int64_t)((now_s() - t_jobs) * 1e9) : 0;
        bool w_any = false;
        struct Stamp {
            std::atomic<int64_t> &first, &last_start, &last_end; std::atomic<int> &n; int64_t t0; const bool &any; bool on; double base;
            ~Stamp()
            {
                if (!on || !any) return;
                const int64_t t1 = (int64_t)((now_s() - base) * 1e9);
                int64_t v = first.load(); while (t0 < v && !first.compare_exchange_weak(v, t0)) {}
                v = last_start.load(); while (t0 > v && !last_start.compare_exchange_weak(v, t0)) {}
                v = last_end.load(); while (t1 > v && !last_end.compare_exchange_weak(v, t1)) {}
                n.fetch_add(1);
            }
        } stamp_{w_first, w_last_start, w_last_end, w_with_work, w_t0, w_any, trace, t_jobs};
        for (;;) {
            // Jobs are taken a run at a time: neighbouring jobs write neighbouring entries of the hit columns, and handed out
            // one by one they made 24 threads share every cache line of the one-byte columns (660 ns of CPU per hit row).
            if (jx_at == jx_end) {
                jx_at = next.fetch_add(job_run, std::memory_order_relaxed);
                if (jx_at >= jobs.size()) break;
                jx_end = std::min(jobs.size(), jx_at + job_run);
            }
            const size_t jx = jx_at++;
            w_any = true;
            HitJob &job = jobs[jx];
            MotifHits &h = sc->hits[job.motif];
            const std::string &path = sc->paths[job.file];
            gfm_tsv_detail::NameTable names(job.names);
            const int64_t base_row = t.row_base[job.file];
            auto store = [&](size_t i, const gfm_tsv_detail::LineCols &c) {
                std::memcpy(h.kmers.data() + i * W, c.kmer, W);
                h.start[i] = c.start; h.stop[i] = c.stop; h.freq[i] = c.freq;
                h.strand[i] = c.strand; h.is_ref[i] = c.is_ref;
                h.name_id[i] = names.id(c.name, c.name_len);
            };
            auto offset_of = [&](size_t i) {
                const int64_t r = h.rows[i];
                return P->meta[(size_t)(r / sc->chunk_rows)].line_off[(size_t)(r % sc->chunk_rows)];
            };
            auto complain = [&](size_t i, const char *what) {
                fail_job(path + ": row " + std::to_string((long long)(h.rows[i] - base_row + 1)) + " of the file: " + what);
            };
            if (const char *text = sc->kept[job.file].p) {       // the file's text is still in the scan's arena
                const char *tend = text + sc->kept[job.file].len;
                const int64_t jc0 = trace ? tcpu() : 0;
                // (the lines lie anywhere in 1.8 GB of text that has left every cache: ask for them all before the first is parsed)
                for (size_t i = job.i0; i < job.i1; ++i) {
                    const uint64_t off = offset_of(i);
                    if (off < (uint64_t)(tend - text)) {
                        __builtin_prefetch(text + off);
                        __builtin_prefetch(text + off + 64);
                        __builtin_prefetch(text + off + 127);     // (the splitter looks at 128 bytes from wherever the line starts)
                    }
                }
                const int64_t jc1 = trace ? tcpu() : 0;
                struct Acc {
                    std::atomic<int64_t> &a, &b; int64_t t0, t1; bool on; int64_t (*clk)();
                    ~Acc() { if (on) { a.fetch_add(t1 - t0, std::memory_order_relaxed); b.fetch_add(clk() - t1, std::memory_order_relaxed); } }
                } acc{cpu_offsets, cpu_lines, jc0, jc1, trace, +tcpu};
                for (size_t i = job.i0; i < job.i1; ++i) {
                    const uint64_t off = offset_of(i);
                    gfm_tsv_detail::LineCols c;
                    const char *what = "internal error: line offset outside the file";
                    if (off >= (uint64_t)(tend - text) || gfm_tsv_detail::parse_line(text + off, tend, true, (int)W, c, &what) != 1) {
                        complain(i, what);
                        break;
                    }
                    store(i, c);
                }
                continue;
            }
            const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
            if (fd < 0) { fail_job("cannot open " + path + " again for its hit rows"); continue; }
            struct stat sb {};
            const bool have_size = ::fstat(fd, &sb) == 0;
            const size_t n_hits = job.i1 - job.i0;
            if (have_size && (uint64_t)n_hits * 2048u >= (uint64_t)sb.st_size) {
                ::close(fd);
                std::string lerr;
                if (!whole.load(path.c_str(), lerr)) { fail_job(lerr); continue; }
                for (size_t i = job.i0; i < job.i1; ++i) {
                    const uint64_t off = offset_of(i);
                    gfm_tsv_detail::LineCols c;
                    const char *what = "the file changed during the scan";
                    if (off >= (uint64_t)(whole.end() - whole.begin()) ||
                        gfm_tsv_detail::parse_line(whole.begin() + off, whole.end(), true, (int)W, c, &what) != 1) {
                        complain(i, what);
                        break;
                    }
                    store(i, c);
                }
                whole.drop();
                continue;
            }
            for (size_t i = job.i0; i < job.i1; ++i) {
                const uint64_t off = offset_of(i);
                size_t want = 1024;
                for (;;) {
                    buf.resize(want + 64);
                    size_t got = 0;
                    bool eof = false, io_bad = false;
                    while (got < want) {
                        const ssize_t r = ::pread(fd, buf.data() + got, want - got, (off_t)(off + got));
                        if (r < 0) { if (errno == EINTR) continue; io_bad = true; break; }
                        if (r == 0) { eof = true; break; }
                        got += (size_t)r;
                    }
                    if (io_bad) { complain(i, "read error"); break; }
                    std::memset(buf.data() + got, 0, 64);
                    gfm_tsv_detail::LineCols c;
                    const char *what = "the file changed during the scan";
                    const int rc = got ? gfm_tsv_detail::parse_line(buf.data(), buf.data() + got, eof, (int)W, c, &what) : 0;
                    if (rc == 1) { store(i, c); break; }
                    if (rc == 0 || eof) { complain(i, what); break; }
                    want *= 8;                 // the sixth column ends behind what was read (a long line): read more
                }
            }
            ::close(fd);
        }
    };
    const double t_run = trace ? now_s() : 0.0;
    gfm_workers::run(n_workers, work);
    const double t_merge = trace ? now_s() : 0.0;
    if (!err.empty()) return sfail(GFM_ERR_IO, err);
    // local name ids -> ids in the scan's table
    gfm_tsv_detail::NameTable global(sc->table.names);
    global.ix.reserve(jobs.size() + jobs.size() / 4);          // (a file is mostly one region: about a name per job)
    sc->table.names.reserve(jobs.size() + jobs.size() / 4);
    std::vector<int32_t> map;
    for (auto &job : jobs) {
        map.resize(job.names.size());
        for (size_t q = 0; q < job.names.size(); ++q) map[q] = global.id(job.names[q].data(), job.names[q].size());
        MotifHits &h = sc->hits[job.motif];
        if (job.names.size() == 1) {
            const int32_t only = map[0];
            for (size_t i = job.i0; i < job.i1; ++i) h.name_id[i] = only;
        } else {
            for (size_t i = job.i0; i < job.i1; ++i) h.name_id[i] = map[(size_t)h.name_id[i]];
        }
    }
    if (trace)
        std::fprintf(stderr, "[scan] hit columns: %zu jobs listed in %.3f ms, run %.3f ms (worker CPU: offsets + prefetch %.1f ms, lines %.1f ms), "
                             "names merged in %.3f ms; %d of %d workers had jobs, the first started at %.3f ms, the last at %.3f ms, the "
                             "last finished at %.3f ms\n", jobs.size(), (t_run - t_jobs) * 1e3, (t_merge - t_run) * 1e3,
                     cpu_offsets.load() * 1e-6, cpu_lines.load() * 1e-6, (now_s() - t_merge) * 1e3, w_with_work.load(), n_workers,
                     w_first.load() * 1e-6, w_last_start.load() * 1e-6, w_last_end.load() * 1e-6);
    return GFM_OK;
}

}  // namespace

GFM_API int gfm_scan_tsv_begin(const gfm_motif_t *motifs, int n_motifs, const char *const *paths, int n_paths,
                               int skip_reverse, int n_threads, double threshold, int on_qvalue, int want_qvalues,
                               int64_t chunk_rows, uint64_t *const *d_hist_ext, gfm_scan_t *out, int64_t *n_rows)
{
    if (!motifs || n_motifs < 1 || !out || !n_rows || (n_paths > 0 && !paths)) return sfail(GFM_ERR_INVALID, "NULL argument");
    *out = nullptr;
    *n_rows = 0;
    if (n_paths < 0) return sfail(GFM_ERR_INVALID, "negative path count");
    if (!(threshold > 0 && threshold <= 1)) return sfail(GFM_ERR_INVALID, "threshold must be in (0, 1]");
    if (on_qvalue && !want_qvalues) return sfail(GFM_ERR_INVALID, "q-value threshold without q-values");
    for (int j = 0; j < n_motifs; ++j) {
        if (!motifs[j]) return sfail(GFM_ERR_INVALID, "motif is NULL");
        if (gfm_motif_width(motifs[j]) != gfm_motif_width(motifs[0]))
            return sfail(GFM_ERR_INVALID, "the motifs of one scan must share their width");
        if (d_hist_ext && want_qvalues && !d_hist_ext[j]) return sfail(GFM_ERR_INVALID, "a histogram buffer is NULL");
    }
    const int W = gfm_motif_width(motifs[0]);
    const int L = gfm_motif_table_len(motifs[0]);
    const size_t M = (size_t)n_motifs;
    if (chunk_rows <= 0) chunk_rows = kDefaultChunkRows;
    chunk_rows = (chunk_rows + 255) & ~(int64_t)255;   // whole 256-row score chunks, 16-byte aligned slices

    gfm_scan *sc = new (std::nothrow) gfm_scan();
    if (!sc) return sfail(GFM_ERR_NOMEM, "out of host memory");
    struct Guard {
        gfm_scan *p;
        ~Guard() { delete p; }
    } guard{sc};
    sc->W = W;
    sc->L = L;
    sc->have_q = want_qvalues != 0;
    sc->want_qvalues = want_qvalues != 0;
    sc->on_qvalue = on_qvalue != 0;
    sc->threshold = threshold;
    sc->chunk_rows = chunk_rows;
    sc->motifs.assign(motifs, motifs + n_motifs);
    sc->hits.resize(M);
    sc->table.W = W;
    sc->table.files.resize((size_t)n_paths);
    sc->paths.reserve((size_t)n_paths);
    for (int i = 0; i < n_paths; ++i) sc->paths.emplace_back(paths[i]);
    sc->kept.assign((size_t)n_paths, gfm_scan::Kept());
    const double t_begin = now_s();
    sc->t_begin = t_begin;
    const bool trace = scan_trace_on();   // development aid: phase times to stderr
    auto stamp = [&](const char *what) {
        if (trace) std::fprintf(stderr, "[scan] %8.3f ms  %s\n", (now_s() - t_begin) * 1e3, what);
    };

    // ---- device side
    ScanPool *P = nullptr;
    S_RC(acquire_pool(&P));
    sc->pool = P;                        // released by ~gfm_scan
    // Any exit that is not the successful one leaves copies and score kernels of earlier chunks in flight: they still read
    // the pool's pinned slots and add into the histograms (the caller's own buffers, possibly freed on the exception).
    // Declared behind `guard` and before the crew: on the way out the workers are joined first, then the streams
    // drained, then the pool goes back.
    struct DrainOnFail {
        ScanPool *p;
        bool armed = true;
        ~DrainOnFail()
        {
            if (!armed || !p) return;
            (void)hipStreamSynchronize(p->copy);
            (void)hipStreamSynchronize(p->score);
        }
    } drain{P};
    stamp("pool acquired");
    S_RC(P->reserve_slots((size_t)chunk_rows * (size_t)W + 16));
    int64_t est_rows = 0;
    {   // the text arena: the directory's size estimated from a sample of its files (a stat per file was 0.8 ms per thousand)
        unsigned long long bytes = 0;
        const int step = (n_paths + 63) / 64;
        int sampled = 0;
        for (int i = 0; i < n_paths; i += std::max(1, step), ++sampled) {
            struct stat sb {};
            if (::stat(paths[i], &sb) == 0) bytes += (unsigned long long)sb.st_size;
        }
        if (sampled) bytes = bytes / (unsigned long long)sampled * (unsigned long long)n_paths;
        est_rows = (int64_t)(bytes / (unsigned long long)(2 * W + 56));      // a row: two W-mers' worth of text and change
        size_t want = (size_t)(bytes + bytes / 16) + (size_t)n_paths * 64 + (1u << 20);
        if (const char *e = std::getenv("GRAFIMO_SCAN_TEXT_BYTES")) want = (size_t)strtoull(e, nullptr, 10);   // test aid (0: keep nothing)
        if (want > ScanPool::kTextMax) want = ScanPool::kTextMax;
        if (want && P->reserve_text(want) != GFM_OK) (void)P->reserve_text(0);      // no memory for it: the files are read back instead
    }
    std::atomic<size_t> text_used{0};
    S_RC(P->reserve_motifs(M));
    sc->n_threads = gfm_tsv_detail::pick_threads(paths, n_paths, n_threads);
    sc->d_hist.assign(M, nullptr);
    sc->cutoffs.assign(M, GFM_NO_SELECT);
    for (size_t j = 0; j < M; ++j) {
        MotifBufs &b = P->mb[j];
        if (want_qvalues) S_RC(b.reserve_tables((size_t)L));
        // room for one row in sixteen (a p < 1e-4 scan reports one in ten thousand; the bench plants one in a hundred); a
        // list that turns out too short is grown at finish and the scan asked for again (GFM_ERR_OVERFLOW)
        S_RC(b.reserve_hits(std::max<int64_t>(std::max<int64_t>(std::max<int64_t>(b.hit_cap, 1 << 20), est_rows / 16), b.want_cap)));
        sc->d_hist[j] = want_qvalues ? (d_hist_ext ? d_hist_ext[j] : b.d_hist) : nullptr;
        // p-value threshold: the cutoff is known before scoring and the score kernel selects the hits.  q-value
        // threshold: q >= p, so the score kernel collects the p < t CANDIDATES the same way, and the selection behind
        // the q-table filters those instead of reading every score again (gfm_select_hits_from).
        S_RC(gfm_motif_pvalue_cutoff(motifs[j], threshold, &sc->cutoffs[j]));
    }
    const bool fused = !on_qvalue;
    stamp("pool sized");

    // ---- host pipeline.  The files are taken in path order.  A worker READS a file into its own small text buffer and
    // SCANS it at once (k-mer + line offset per row into a stage of the worker): that fixes the file's global row offset
    // as soon as every earlier file is scanned.  The text goes to the scan's arena (non-temporal copy) for the hit rows'
    // columns later; the stage's rows are COPIED to where they belong -- the k-mers into the pinned slot of their chunk,
    // the line offsets into the chunk's block -- once the offset is known.  One read() per file, no allocation once the
    // buffers have their size.  What was measured on the way (CPU seconds per 2e7 rows in 10 000 files, 16-CPU quota):
    // round 2 parsed into per-file columns (page faults of 1.4 GB of fresh vectors: 250 ms wall); round 3 read into ring
    // buffers, counted, parsed every column in place (1.1 s); round 4 first read straight into the arena, counted, and
    // scanned k-mers only (0.86 s: the read() into cold memory alone 0.46 s, profiles/r04_ingest_cpu.txt), then this.
    // The calling thread only sequences chunks: chunk k goes to the device once every row of it is in place; its slot is
    // handed back to the workers when the score kernel has read it.
    const int nt = sc->n_threads;            // (pick_threads() once per scan: it notes when it was last asked)
    constexpr int kRing = 4;                 // files a worker may hold scanned, waiting for their offsets
    // Coordination is lock-light on purpose: with one mutex taken ~5 times per file, 96 workers on 10 000 files spent
    // more time handing the mutex around than parsing (300 ms at 96 threads against 110 ms at 32).  Files are claimed
    // with an atomic counter; row counts are published through per-file flags and the offsets advanced by whichever
    // worker gets a try-lock; rows in place are atomic adds per chunk, and only the add that completes a chunk (20 of
    // them at 2e7 rows) wakes the calling thread.
    constexpr size_t kMaxChunks = 1 << 16;   // chunk counters (6.9e10 rows at the default chunk size)
    std::mutex mu;                           // guards the two condition variables and the failure message only
    std::condition_variable cv_work, cv_main;
    std::atomic<int> next_count{0};          // next file to read + count
    std::atomic<int> next_assign{0};         // files [0, next_assign) have their row offset
    std::mutex assign_mu;                    // try-lock: one worker at a time advances next_assign
    int64_t assigned_rows = 0;               // under assign_mu; final once next_assign == n_paths
    // == assigned_rows once every file is counted; with no file at all that is now (a rank of a sharded scan whose
    // shard is empty: nobody would ever set it, and the calling thread would poll for chunk 0 forever)
    std::atomic<int64_t> total_assigned{n_paths == 0 ? 0 : -1};
    std::unique_ptr<std::atomic<char>[]> counted(new std::atomic<char>[(size_t)n_paths + 1]);
    for (int i = 0; i <= n_paths; ++i) counted[(size_t)i].store(0, std::memory_order_relaxed);
    std::vector<int64_t> file_off((size_t)n_paths, 0);
    std::unique_ptr<std::atomic<int64_t>[]> chunk_staged(new std::atomic<int64_t>[kMaxChunks]);
    for (size_t k = 0; k < kMaxChunks; ++k) chunk_staged[k].store(0, std::memory_order_relaxed);
    std::atomic<int64_t> released{0};        // chunks whose slot may be written again: chunk k needs k < released + kSlots
    std::atomic<bool> failed{false};
    std::string fail_msg;
    std::atomic<int64_t> parse_end_ns{0};    // steady-clock time of the last finished parse
    // development aid (GRAFIMO_SCAN_TRACE): the longest single wait of each kind, in ns
    std::atomic<int64_t> tr_read{0}, tr_slot{0}, tr_offset{0}, tr_parse{0}, tr_block{0};
    // ... and the CPU time (this thread's clock) the workers spent per phase, summed over the workers, in ns
    std::atomic<int64_t> cpu_read{0}, cpu_count{0}, cpu_parse{0}, cpu_keep{0}, cpu_yield{0}, cpu_all{0};
    auto tcpu = []() -> int64_t {
        timespec ts{};
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
        return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
    };
    auto tr_max = [](std::atomic<int64_t> &a, int64_t v) {
        int64_t prev = a.load(std::memory_order_relaxed);
        while (prev < v && !a.compare_exchange_weak(prev, v, std::memory_order_relaxed)) {}
    };
    auto fail_with = [&](const std::string &msg) {
        {
            std::lock_guard<std::mutex> g(mu);
            if (!failed.load()) fail_msg = msg;
            failed.store(true);
        }
        cv_work.notify_all();
        cv_main.notify_all();
    };
    auto wake_main = [&]() {
        { std::lock_guard<std::mutex> g(mu); }
        cv_main.notify_one();
    };
    auto advance_offsets = [&]() {           // any worker, after publishing a count
        if (!assign_mu.try_lock()) return;   // (whoever holds it sees the flag set before this call, or the next caller does)
        int na = next_assign.load(std::memory_order_relaxed);
        bool moved = false;
        while (na < n_paths && counted[(size_t)na].load(std::memory_order_acquire)) {
            file_off[(size_t)na] = assigned_rows;
            assigned_rows += sc->table.files[(size_t)na].n_rows;
            ++na;
            moved = true;
        }
        if (moved) {
            if (na == n_paths) total_assigned.store(assigned_rows, std::memory_order_release);
            next_assign.store(na, std::memory_order_release);
        }
        assign_mu.unlock();
        if (moved && na == n_paths) wake_main();     // the last chunk may be a short one
    };
    // trace only: what every worker is doing (state * 2^32 + file index); states 1 read, 2 parse, 3 slot wait, 4 yield
    std::unique_ptr<std::atomic<long long>[]> w_state(new std::atomic<long long>[(size_t)nt + 1]);
    for (int i = 0; i <= nt; ++i) w_state[(size_t)i].store(0);
    std::atomic<int> w_next{0};
    auto work = [&]() {
        const int me = w_next.fetch_add(1);
        const int64_t cpu_in = trace ? tcpu() : 0;
        struct CpuAll {
            std::atomic<int64_t> &a; int64_t t0; bool on; int64_t (*clk)();
            ~CpuAll() { if (on) a.fetch_add(clk() - t0, std::memory_order_relaxed); }
        } cpu_total{cpu_all, cpu_in, trace, +tcpu};
        auto set_state = [&](int st, long long file) {
            if (trace && me < nt) w_state[(size_t)me].store(((long long)st << 32) | (file & 0xffffffffll), std::memory_order_relaxed);
        };
        // A file is read into this thread's one text buffer (the same few hundred KB for every file: the kernel's copy out of
        // the page cache lands in L2, 24 us a file against 35 into the arena's cold memory), scanned there at once -- k-mers and
        // line offsets into a STAGE, which also gives the row count that fixes the file's global row offset -- and its text
        // goes to the arena with non-temporal stores (kept until the scan closes: the hit rows' columns are parsed from it).
        // Stages wait, oldest first, until every earlier file is counted; then their rows are copied to where they belong.
        // (The crew's threads live as long as the process: so do these buffers.)
        struct Stage {
            std::vector<uint8_t> kmers;
            std::vector<uint64_t> offs;
            int64_t rows = 0;
        };
        static thread_local gfm_tsv_detail::FileBuf hot;
        static thread_local Stage stage[kRing];
        int held[kRing], n_held = 0, head = 0;         // file index per stage, in order from `head`
        int64_t cached_k = -1;                         // the chunk this worker wrote into last, and its column block
        MetaChunk cached_mc;
        struct DropAll {
            gfm_tsv_detail::FileBuf &h;
            Stage *st;
            ~DropAll()
            {
                h.drop();
                for (int k = 0; k < kRing; ++k)        // (a stage that grew for one big file is handed back)
                    if (st[k].kmers.capacity() > ((size_t)8 << 20)) { std::vector<uint8_t>().swap(st[k].kmers); std::vector<uint64_t>().swap(st[k].offs); }
            }
        } drop_all{hot, stage};
        for (;;) {
            if (failed.load(std::memory_order_relaxed)) return;
            if (n_held > 0 && held[head] < next_assign.load(std::memory_order_acquire)) {
                // my oldest file has its offset: its rows go into their chunk(s)
                const int i = held[head];
                set_state(2, i);
                const Stage &st = stage[head];
                const int64_t rows = st.rows, off = file_off[(size_t)i];
                std::string err;
                const int64_t pc0 = trace ? tcpu() : 0;
                for (int64_t done = 0; done < rows && err.empty();) {
                    const int64_t g = off + done, k = g / chunk_rows, in = g % chunk_rows;
                    const int64_t cnt = std::min(rows - done, chunk_rows - in);
                    if ((size_t)k >= kMaxChunks) { err = "too many rows for one scan"; break; }
                    if (k >= released.load(std::memory_order_acquire) + kSlots) {   // rare: far ahead of the GPU
                        const double w0 = trace ? now_s() : 0.0;
                        set_state(3, (long long)i | ((long long)k << 20));
                        {
                            std::unique_lock<std::mutex> g2(mu);
                            cv_work.wait(g2, [&] { return failed.load() || k < released.load() + kSlots; });
                        }
                        if (trace) tr_max(tr_slot, (int64_t)((now_s() - w0) * 1e9));
                        set_state(2, i);
                    }
                    if (failed.load()) return;
                    if (k != cached_k) {        // consecutive files of a worker mostly stay in one chunk
                        const MetaChunk *got;
                        const double b0 = trace ? now_s() : 0.0;
                        {
                            std::lock_guard<std::mutex> g3(assign_mu);      // the pool's block table
                            got = P->meta_chunk((size_t)k, chunk_rows, W);
                            if (got) cached_mc = *got;      // by value: the pool's vector of blocks may grow meanwhile
                        }
                        if (trace) tr_max(tr_block, (int64_t)((now_s() - b0) * 1e9));
                        if (!got) { err = "out of memory"; break; }
                        cached_k = k;
                    }
                    std::memcpy(P->h_pin[k % kSlots] + (size_t)in * (size_t)W, st.kmers.data() + (size_t)done * (size_t)W,
                                (size_t)cnt * (size_t)W);
                    std::memcpy(cached_mc.line_off + in, st.offs.data() + done, (size_t)cnt * sizeof(uint64_t));
                    const int64_t now = chunk_staged[(size_t)k].fetch_add(cnt, std::memory_order_acq_rel) + cnt;
                    // the add that completes a chunk wakes the calling thread (a short last chunk: whoever makes
                    // it reach the total, or advance_offsets when the total becomes known)
                    const int64_t tot = total_assigned.load(std::memory_order_acquire);
                    if (now == chunk_rows || (tot >= 0 && now == tot - k * chunk_rows)) wake_main();
                    done += cnt;
                }
                if (trace) cpu_count.fetch_add(tcpu() - pc0, std::memory_order_relaxed);
                head = (head + 1) % kRing;
                --n_held;
                if (!err.empty()) { fail_with(err); return; }
                const int64_t t_ns = (int64_t)(now_s() * 1e9);
                int64_t prev = parse_end_ns.load(std::memory_order_relaxed);
                while (prev < t_ns && !parse_end_ns.compare_exchange_weak(prev, t_ns, std::memory_order_relaxed)) {}
                continue;
            }
            if (n_held < kRing && next_count.load(std::memory_order_relaxed) < n_paths) {
                const int i = next_count.fetch_add(1, std::memory_order_relaxed);    // read and scan the next file
                if (i < n_paths) {
                    const int slot_ix = (head + n_held) % kRing;
                    Stage &st = stage[slot_ix];
                    std::string err;
                    const double r0 = trace ? now_s() : 0.0;
                    const int64_t rc0 = trace ? tcpu() : 0;
                    set_state(1, i);
                    st.rows = 0;
                    if (!hot.load(paths[i], err)) { fail_with(err); return; }
                    const int64_t rc1 = trace ? tcpu() : 0;
                    if (trace) tr_max(tr_read, (int64_t)((now_s() - r0) * 1e9));
                    const size_t len = (size_t)(hot.end() - hot.begin());
                    if (len > 0) {
                        // room for the rows this text can hold at most (a row is a k-mer, five more columns and six separators)
                        const size_t most = len / ((size_t)W + 11) + 1;
                        if (st.offs.size() < most || st.kmers.size() < most * (size_t)W + 64) {      // (the stage outlives the scan: W may have changed)
                            st.offs.resize(most + most / 8);
                            st.kmers.resize((most + most / 8) * (size_t)W + 64);
                        }
                        uint8_t *kd = st.kmers.data();
                        uint64_t *od = st.offs.data();
                        int64_t n = 0;
                        auto sink = [&](const uint8_t *kmer, uint64_t line_off) {
                            gfm_tsv_detail::copy_kmer(kd + (size_t)n * (size_t)W, kmer, W);
                            od[n] = line_off;
                            ++n;
                        };
                        const double p0 = trace ? now_s() : 0.0;
                        const bool ok = gfm_tsv_detail::scan_rows(paths[i], hot.begin(), hot.end(), W, skip_reverse != 0, sink, err);
                        if (trace) tr_max(tr_parse, (int64_t)((now_s() - p0) * 1e9));
                        if (!ok) { fail_with(err); return; }
                        st.rows = n;
                    }
                    const int64_t rc2 = trace ? tcpu() : 0;
                    held[slot_ix] = i;
                    ++n_held;
                    sc->table.files[(size_t)i].n_rows = st.rows;
                    counted[(size_t)i].store(1, std::memory_order_release);
                    advance_offsets();
                    // the text into the scan's arena while the file fits what is left of it
                    if (P->text_cap && len > 0 && len <= ((size_t)32 << 20)) {
                        const size_t at = text_used.fetch_add(len + 64, std::memory_order_relaxed);
                        if (at + len + 64 <= P->text_cap) {
                            char *dst = P->text + at;
                            gfm_tsv_detail::copy_streaming(dst, hot.begin(), len);
                            std::memset(dst + len, 0, 64);
                            sc->kept[(size_t)i].p = dst;
                            sc->kept[(size_t)i].len = len;
                        }
                    }
                    if (trace) {
                        const int64_t rc3 = tcpu();
                        cpu_read.fetch_add(rc1 - rc0, std::memory_order_relaxed);
                        cpu_parse.fetch_add(rc2 - rc1, std::memory_order_relaxed);
                        cpu_keep.fetch_add(rc3 - rc2, std::memory_order_relaxed);
                    }
                    continue;
                }
            }
            if (n_held == 0) { set_state(0, 0); return; }        // nothing left for this thread to take
            // my oldest file waits for earlier counts (they are being made right now): help, then give way
            advance_offsets();
            if (held[head] >= next_assign.load(std::memory_order_acquire)) {
                const double y0 = trace ? now_s() : 0.0;
                const int64_t yc0 = trace ? tcpu() : 0;
                set_state(4, held[head]);
                std::this_thread::yield();
                if (trace) tr_max(tr_offset, (int64_t)((now_s() - y0) * 1e9));
                if (trace) cpu_yield.fetch_add(tcpu() - yc0, std::memory_order_relaxed);
            }
        }
    };
    stamp("pipeline set up");
    // the workers come from the process-wide crew (gfm_workers.hpp): starting threads per call cost more than
    // the parsing they did
    struct Crew {
        gfm_workers::Run run;
        std::mutex &mu;
        std::condition_variable &cv;
        std::atomic<bool> &failed;
        ~Crew()
        {
            {
                std::lock_guard<std::mutex> lk(mu);
                failed.store(true);     // whoever still waits gives up (after a normal run nobody does)
            }
            cv.notify_all();
            run.wait();
        }
    } crew{{}, mu, cv_work, failed};
    crew.run.start(nt, work);
    stamp("workers started");
    // (behind the start of the workers: the memsets take host time to enqueue)
    for (size_t j = 0; j < M; ++j) {
        MotifBufs &b = P->mb[j];
        if (want_qvalues) S_TRY(hipMemsetAsync(sc->d_hist[j], 0, sizeof(uint64_t) * (size_t)L, P->score));
        S_TRY(hipMemsetAsync(b.d_count, 0, sizeof(uint64_t), P->score));
        S_TRY(hipMemsetAsync(b.d_cand_count, 0, sizeof(uint64_t), P->score));
    }
    stamp("buffers cleared");

    int64_t total_rows = 0;
    size_t n_chunks = 0;
    double h2d_ms = 0.0;
    int64_t h2d_bytes = 0;
    auto release_chunk = [&](int64_t k) -> int {   // chunk k has been copied AND scored: its slot is free again
        const int s = (int)(k % kSlots);
        S_TRY(hipEventSynchronize(P->scored[s]));
        float ms = 0.f;
        S_TRY(hipEventElapsedTime(&ms, P->c0[s], P->c1[s]));
        h2d_ms += ms;
        {
            std::lock_guard<std::mutex> lk(mu);
            released.store(k + 1, std::memory_order_release);
        }
        cv_work.notify_all();
        return GFM_OK;
    };
    std::vector<int32_t *> d_sc(M, nullptr);          // no score is stored: hits and histograms are all a scan needs
    std::vector<uint64_t *> v_hist(M), v_count(M);
    std::vector<int64_t *> v_sel(M);
    std::vector<int64_t> v_cap(M);
    double m_wait = 0, m_submit = 0, m_release = 0;     // trace: the main thread's longest wait / submit / release
    for (int64_t k = 0;; ++k) {
        int64_t rows_k = 0;
        const double t_w0 = trace ? now_s() : 0.0;
        {   // wait until chunk k is fully in place (or turns out not to exist)
            auto ready = [&]() {
                if (failed.load()) return true;
                if ((size_t)k >= kMaxChunks) return true;
                const int64_t have = chunk_staged[(size_t)k].load(std::memory_order_acquire);
                if (have == chunk_rows) return true;
                const int64_t tot = total_assigned.load(std::memory_order_acquire);
                if (tot >= 0)                               // every row count is known: the last chunk may be short
                    return have == std::min(chunk_rows, std::max<int64_t>(0, tot - k * chunk_rows));
                return false;
            };
            std::unique_lock<std::mutex> lk(mu);
            bool dumped = false;
            while (!ready()) {
                cv_main.wait_for(lk, std::chrono::microseconds(500));   // (woken by the completing add)
                // stuck for 10 ms: who holds what.  (What this showed on the GPU boxes of round 3: every worker asleep in its
                // slot wait or in the middle of a parse, for 20-80 ms at a time, with nothing in the pipeline to wait for --
                // the container's CPU quota, 16 CPUs' worth of time per 100 ms on a 256-thread host, was used up and the
                // kernel had frozen all threads until the next period: /sys/fs/cgroup/cpu.stat nr_throttled.)
                if (trace && !dumped && now_s() - t_w0 > 0.010) {
                    dumped = true;
                    int cnt[5] = {0, 0, 0, 0, 0};
                    long long lo_file[5] = {1ll << 40, 1ll << 40, 1ll << 40, 1ll << 40, 1ll << 40};
                    for (int i = 0; i < nt; ++i) {
                        const long long v = w_state[(size_t)i].load();
                        const int st = (int)(v >> 32);
                        if (st >= 0 && st < 5) { ++cnt[st]; lo_file[st] = std::min(lo_file[st], v & 0xffffffffll); }
                    }
                    for (int i = 0; i < nt; ++i) {
                        const long long v = w_state[(size_t)i].load();
                        if ((int)(v >> 32) == 3 && ((v >> 20) & 0xfff) <= k)
                            std::fprintf(stderr, "[scan]   worker %d waits for a slot with file %lld, chunk %lld\n", i, v & 0xfffff,
                                         (v >> 20) & 0xfff);
                    }
                    std::fprintf(stderr, "[scan] STUCK at chunk %lld: have %lld of %lld rows; next_count %d next_assign %d released %lld; "
                                         "workers idle %d, reading %d (lowest file %lld), parsing %d (lowest %lld), slot wait %d (lowest %lld), "
                                         "yield %d (lowest %lld)\n",
                                 (long long)k, (long long)chunk_staged[(size_t)k].load(), (long long)chunk_rows, next_count.load(),
                                 next_assign.load(), (long long)released.load(), cnt[0], cnt[1], lo_file[1], cnt[2], lo_file[2], cnt[3],
                                 lo_file[3], cnt[4], lo_file[4]);
                }
            }
            if (failed.load()) return sfail(GFM_ERR_IO, fail_msg);
            rows_k = (size_t)k < kMaxChunks ? chunk_staged[(size_t)k].load(std::memory_order_acquire) : 0;
        }
        if (rows_k == 0) break;
        stamp("chunk staged");
        const double t_s0 = trace ? now_s() : 0.0;
        if (trace) m_wait = std::max(m_wait, t_s0 - t_w0);
        const int slot = (int)(k % kSlots);
        const size_t bytes = (size_t)rows_k * (size_t)W;
        for (size_t j = 0; j < M; ++j) {
            MotifBufs &b = P->mb[j];
            v_hist[j] = sc->d_hist[j];
            v_sel[j] = fused ? b.d_hits : b.d_cand;             // the list the score kernel appends to
            v_count[j] = fused ? b.d_count : b.d_cand_count;
            v_cap[j] = b.hit_cap;
        }
        S_TRY(hipEventRecord(P->c0[slot], P->copy));
        S_TRY(hipMemcpyAsync(P->d_kmers[slot], P->h_pin[slot], bytes, hipMemcpyHostToDevice, P->copy));
        S_TRY(hipEventRecord(P->c1[slot], P->copy));
        S_TRY(hipEventRecord(P->copied[slot], P->copy));
        S_TRY(hipStreamWaitEvent(P->score, P->copied[slot], 0));
        if (M == 1)
            S_RC(gfm_score_kmers(motifs[0], P->d_kmers[slot], rows_k, d_sc[0], v_hist[0], sc->cutoffs[0], total_rows,
                                 v_sel[0], v_cap[0], v_count[0], 0, P->score, nullptr));
        else
            S_RC(gfm_score_kmers_multi(motifs, n_motifs, P->d_kmers[slot], rows_k, nullptr,
                                       want_qvalues ? v_hist.data() : nullptr, sc->cutoffs.data(), total_rows, v_sel.data(),
                                       v_cap.data(), v_count.data(), 0, P->score));
        S_TRY(hipEventRecord(P->scored[slot], P->score));
        h2d_bytes += (int64_t)bytes;
        sc->chunk_n.push_back(rows_k);
        total_rows += rows_k;
        ++n_chunks;
        const double t_r0 = trace ? now_s() : 0.0;
        if (trace) m_submit = std::max(m_submit, t_r0 - t_s0);
        if (k + 1 >= kSlots) S_RC(release_chunk(k + 1 - kSlots));   // the slot chunk k+1 will be written into
        if (trace) m_release = std::max(m_release, now_s() - t_r0);
        if (rows_k < chunk_rows) break;                             // a short chunk is the last one
    }
    stamp("chunks submitted");
    // every row is parsed by now (the last chunk was complete): the workers are done or about to be.  The host-side
    // index of the rows is made while the GPU still copies and scores the last chunks; their slots are waited for after it
    crew.run.wait();
    stamp("workers done");
    {
        std::lock_guard<std::mutex> lk(mu);
        if (failed.load()) {
            (void)hipStreamSynchronize(P->copy);         // nothing of this scan may still use the pool's slots
            (void)hipStreamSynchronize(P->score);
            return sfail(GFM_ERR_IO, fail_msg);
        }
    }
    sc->table.index_rows();
    stamp("rows indexed");
    for (int64_t k = std::max<int64_t>(0, (int64_t)n_chunks - (kSlots - 1)); k < (int64_t)n_chunks; ++k)
        S_RC(release_chunk(k));
    stamp("chunks released");
    const double t_parse_end = std::max(t_begin, (double)parse_end_ns.load() * 1e-9);
    if (trace) {
        std::fprintf(stderr, "[scan] %8.3f ms  (last file parsed)\n", (t_parse_end - t_begin) * 1e3);
        std::fprintf(stderr, "[scan] main thread, longest: wait for a chunk %.3f ms, submit %.3f ms, release %.3f ms\n",
                     m_wait * 1e3, m_submit * 1e3, m_release * 1e3);
        std::fprintf(stderr, "[scan] longest single: read+count %.3f ms, parse of a file %.3f ms, wait for a slot %.3f ms, "
                             "column block lookup %.3f ms, yield %.3f ms\n", tr_read.load() * 1e-6, tr_parse.load() * 1e-6,
                     tr_slot.load() * 1e-6, tr_block.load() * 1e-6, tr_offset.load() * 1e-6);
        const double rows_d = (double)std::max<int64_t>(total_rows, 1);
        std::fprintf(stderr, "[scan] worker CPU (thread clocks, summed): read %.1f ms (%.1f ns/row), scan %.1f ms (%.1f), text to the arena "
                             "%.1f ms (%.1f), rows into their chunks %.1f ms (%.1f), yield %.1f ms (%.1f), all %.1f ms (%.1f ns/row)\n",
                     cpu_read.load() * 1e-6, cpu_read.load() / rows_d, cpu_parse.load() * 1e-6, cpu_parse.load() / rows_d,
                     cpu_keep.load() * 1e-6, cpu_keep.load() / rows_d, cpu_count.load() * 1e-6, cpu_count.load() / rows_d,
                     cpu_yield.load() * 1e-6, cpu_yield.load() / rows_d,
                     cpu_all.load() * 1e-6, cpu_all.load() / rows_d);
    }
    if (sc->table.n != total_rows) return sfail(GFM_ERR_IO, "internal error: row count mismatch");
    S_TRY(hipStreamSynchronize(P->score));      // the histograms are complete for whoever reads them next
    sc->total_rows = total_rows;
    sc->n_chunks = n_chunks;
    sc->t_parsed = t_parse_end;
    sc->begin_s = now_s() - t_begin;
    sc->stats.n_rows = total_rows;
    sc->stats.n_chunks = (int64_t)n_chunks;
    sc->stats.h2d_bytes = h2d_bytes;
    sc->stats.h2d_s = h2d_ms * 1e-3;
    sc->stats.parse_s = t_parse_end - t_begin;
    sc->stats.parse_threads = nt;
    *n_rows = total_rows;
    drain.armed = false;
    guard.p = nullptr;
    *out = sc;
    return GFM_OK;
}

GFM_API int gfm_scan_tsv_finish(gfm_scan_t sc, int64_t *n_hits)
{
    if (!sc) return sfail(GFM_ERR_INVALID, "NULL handle");
    if (sc->finished) return sfail(GFM_ERR_INVALID, "gfm_scan_tsv_finish was already called on this scan");
    ScanPool *P = sc->pool;
    const size_t M = sc->motifs.size();
    const int L = sc->L;
    const double t0 = now_s();
    const bool trace = scan_trace_on();
    auto proc_cpu = []() {
        timespec ts{};
        clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts);
        return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    };
    const double c0 = trace ? proc_cpu() : 0.0;
    double c_sorted = c0, t_sorted = t0;
    {
        int dev = -1;
        S_TRY(hipGetDevice(&dev));
        if (dev != P->device) return sfail(GFM_ERR_INVALID, "the scan lives on another device than the current one");
    }
    if (sc->total_rows > 0) {
        // ---- tables
        if (sc->want_qvalues) {
            std::vector<double *> v_q(M);
            std::vector<int32_t *> v_cut(M);
            for (size_t j = 0; j < M; ++j) { v_q[j] = P->mb[j].d_q; v_cut[j] = P->mb[j].d_cutoff; }
            if (M == 1)
                S_RC(gfm_qvalue_table(sc->motifs[0], sc->d_hist[0], sc->threshold, sc->on_qvalue, v_q[0], v_cut[0], nullptr, 0,
                                      P->score));
            else
                S_RC(gfm_qvalue_table_multi(sc->motifs.data(), (int)M, sc->d_hist.data(), sc->threshold, sc->on_qvalue,
                                            v_q.data(), v_cut.data(), nullptr, 0, P->score));
        }
        // ---- were the lists long enough?  Asked for ALL motifs before any is selected from: every list that was short is grown
        // in this one go, so that the one retry the callers make (StreamScan.finish, _scan_width_sharded) serves a motif set too
        {
            std::vector<uint64_t> counted(M, 0);
            for (size_t j = 0; j < M; ++j)
                S_TRY(hipMemcpyAsync(&counted[j], sc->on_qvalue ? P->mb[j].d_cand_count : P->mb[j].d_count, sizeof(uint64_t),
                                     hipMemcpyDeviceToHost, P->score));
            S_TRY(hipStreamSynchronize(P->score));
            uint64_t worst = 0;
            for (size_t j = 0; j < M; ++j) {
                MotifBufs &b = P->mb[j];
                if ((int64_t)counted[j] <= b.hit_cap) continue;
                const int64_t cap = (int64_t)counted[j] + (int64_t)(counted[j] >> 3) + 1024;
                S_RC(b.reserve_hits(cap));
                b.want_cap = cap;
                worst = std::max(worst, counted[j]);
            }
            if (worst)
                return sfail(GFM_ERR_OVERFLOW, "the hit list of this scan was too short (" + std::to_string(worst) + " rows pass the threshold); it "
                                               "has been grown: run the scan again");
        }
        // ---- selection, hits back: motif by motif
        for (size_t j = 0; j < M; ++j) {
            MotifBufs &b = P->mb[j];
            gfm_motif_t m = sc->motifs[j];
            // A list that was too short (the score kernel counts what it could not store): no score was kept to select from
            // again, so the list is grown to what the count asks for and the caller runs the scan once more.
            auto too_short = [&](uint64_t need) -> int {
                const int64_t cap = (int64_t)need + (int64_t)(need >> 3) + 1024;
                S_RC(b.reserve_hits(cap));
                b.want_cap = cap;
                return sfail(GFM_ERR_OVERFLOW, "the hit list of this scan was too short (" + std::to_string(need) + " rows pass the threshold); it "
                                               "has been grown: run the scan again");
            };
            if (sc->on_qvalue) {
                uint64_t ccnt = 0;
                S_TRY(hipMemcpyAsync(&ccnt, b.d_cand_count, sizeof ccnt, hipMemcpyDeviceToHost, P->score));
                S_TRY(hipStreamSynchronize(P->score));
                if ((int64_t)ccnt > b.hit_cap) return too_short(ccnt);
                // the p < t candidates are complete: those that reach the q-value cutoff are the hits
                S_RC(gfm_select_hits_from(m, nullptr, sc->total_rows, b.d_cutoff, 0, b.d_cand, b.hit_cap, b.d_cand_count, b.d_hits,
                                          b.hit_cap, b.d_count, P->score));
            }
            uint64_t cnt = 0;
            S_TRY(hipMemcpyAsync(&cnt, b.d_count, sizeof cnt, hipMemcpyDeviceToHost, P->score));
            S_TRY(hipStreamSynchronize(P->score));
            if ((int64_t)cnt > b.hit_cap) return too_short(cnt);
            b.want_cap = 0;                    // the list held: a closed scan's trim() may drop it again
            std::vector<int64_t> packed((size_t)cnt);
            std::vector<double> q;
            if (cnt)
                S_TRY(hipMemcpyAsync(packed.data(), b.d_hits, sizeof(int64_t) * (size_t)cnt, hipMemcpyDeviceToHost, P->score));
            if (sc->want_qvalues && cnt) {
                q.resize((size_t)L);
                S_TRY(hipMemcpyAsync(q.data(), b.d_q, sizeof(double) * (size_t)L, hipMemcpyDeviceToHost, P->score));
            }
            S_TRY(hipStreamSynchronize(P->score));
            gfm_hit_sort::sort_packed(packed.data(), packed.size(), GFM_HIT_SCORE_BITS);   // ascending by row
            MotifHits &h = sc->hits[j];
            h.rows.resize((size_t)cnt);
            h.scaled.resize((size_t)cnt);
            h.logodds.resize((size_t)cnt);
            h.pvalue.resize((size_t)cnt);
            for (size_t i = 0; i < (size_t)cnt; ++i) {
                h.rows[i] = packed[i] >> GFM_HIT_SCORE_BITS;
                h.scaled[i] = (int32_t)(packed[i] & ((1ll << GFM_HIT_SCORE_BITS) - 1));
            }
            if (cnt) S_RC(gfm_motif_annotate(m, h.scaled.data(), (int64_t)cnt, h.logodds.data(), h.pvalue.data()));
            if (sc->want_qvalues) {
                h.qvalue.resize((size_t)cnt);
                for (size_t i = 0; i < (size_t)cnt; ++i) h.qvalue[i] = q[(size_t)h.scaled[i]];
            }
        }
        if (trace) { c_sorted = proc_cpu(); t_sorted = now_s(); }
        S_RC(fetch_hit_columns(sc));
    } else {
        S_TRY(hipStreamSynchronize(P->score));
    }
    sc->finished = true;
    const double t_end = now_s();
    if (trace)
        std::fprintf(stderr, "[scan] finish: tables + selection + hits back + order %.3f ms (process CPU %.1f ms), columns of the hit rows "
                             "%.3f ms (process CPU %.1f ms)\n", (t_sorted - t0) * 1e3, (c_sorted - c0) * 1e3, (t_end - t_sorted) * 1e3,
                     (proc_cpu() - c_sorted) * 1e3);
    sc->stats.n_hits = (int64_t)sc->hits[0].rows.size();
    sc->stats.total_s = sc->begin_s + (t_end - t0);          // the caller's time between the two phases is not the scan's
    sc->stats.tail_s = sc->stats.total_s - sc->stats.parse_s;
    if (n_hits)
        for (size_t j = 0; j < M; ++j) n_hits[j] = (int64_t)sc->hits[j].rows.size();
    return GFM_OK;
}

GFM_API int gfm_scan_tsv(gfm_motif_t m, const char *const *paths, int n_paths, int skip_reverse, int n_threads,
                         double threshold, int on_qvalue, int want_qvalues, int64_t chunk_rows, gfm_scan_t *out,
                         int64_t *n_rows, int64_t *n_hits)
{
    if (!m || !out || !n_rows || !n_hits) return sfail(GFM_ERR_INVALID, "NULL argument");
    *n_hits = 0;
    gfm_scan_t sc = nullptr;
    S_RC(gfm_scan_tsv_begin(&m, 1, paths, n_paths, skip_reverse, n_threads, threshold, on_qvalue, want_qvalues, chunk_rows,
                            nullptr, &sc, n_rows));
    const int rc = gfm_scan_tsv_finish(sc, n_hits);
    if (rc) {
        delete sc;
        *n_rows = 0;
        return rc;
    }
    *out = sc;
    return GFM_OK;
}

GFM_API int gfm_scan_hits_of(gfm_scan_t s, int motif, int64_t *rows, int32_t *scaled, double *logodds, double *pvalue,
                             double *qvalue, uint8_t *kmers, int64_t *start, int64_t *stop, uint8_t *strand,
                             int64_t *freq, uint8_t *is_ref, int32_t *name_id)
{
    if (!s) return sfail(GFM_ERR_INVALID, "NULL handle");
    if (!s->finished) return sfail(GFM_ERR_INVALID, "gfm_scan_tsv_finish has not run on this scan");
    if (motif < 0 || (size_t)motif >= s->hits.size()) return sfail(GFM_ERR_INVALID, "motif index outside the scan");
    const MotifHits &h = s->hits[(size_t)motif];
    const size_t k = h.rows.size();
    if (rows) std::memcpy(rows, h.rows.data(), k * sizeof(int64_t));
    if (scaled) std::memcpy(scaled, h.scaled.data(), k * sizeof(int32_t));
    if (logodds) std::memcpy(logodds, h.logodds.data(), k * sizeof(double));
    if (pvalue) std::memcpy(pvalue, h.pvalue.data(), k * sizeof(double));
    if (qvalue) {
        if (!s->have_q) return sfail(GFM_ERR_INVALID, "the scan computed no q-values");
        std::memcpy(qvalue, h.qvalue.data(), k * sizeof(double));
    }
    const size_t W = (size_t)s->W;
    if (kmers) std::memcpy(kmers, h.kmers.data(), k * W);
    if (start) std::memcpy(start, h.start.data(), k * sizeof(int64_t));
    if (stop) std::memcpy(stop, h.stop.data(), k * sizeof(int64_t));
    if (strand) std::memcpy(strand, h.strand.data(), k);
    if (freq) std::memcpy(freq, h.freq.data(), k * sizeof(int64_t));
    if (is_ref) std::memcpy(is_ref, h.is_ref.data(), k);
    if (name_id) std::memcpy(name_id, h.name_id.data(), k * sizeof(int32_t));
    return GFM_OK;
}

GFM_API int gfm_scan_hits(gfm_scan_t s, int64_t *rows, int32_t *scaled, double *logodds, double *pvalue,
                          double *qvalue, uint8_t *kmers, int64_t *start, int64_t *stop, uint8_t *strand,
                          int64_t *freq, uint8_t *is_ref, int32_t *name_id)
{
    return gfm_scan_hits_of(s, 0, rows, scaled, logodds, pvalue, qvalue, kmers, start, stop, strand, freq, is_ref, name_id);
}

GFM_API int gfm_scan_stats(gfm_scan_t s, gfm_scan_stats_t *out)
{
    if (!s || !out) return sfail(GFM_ERR_INVALID, "NULL argument");
    *out = s->stats;
    return GFM_OK;
}

GFM_API gfm_tsv_t gfm_scan_table(gfm_scan_t s) { return s ? &s->table : nullptr; }

GFM_API void gfm_scan_close(gfm_scan_t s) { delete s; }

GFM_API void gfm_scan_release_buffers(void)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto it = g_pools.begin(); it != g_pools.end();) {
        if ((*it)->in_use) { ++it; continue; }
        int cur = 0;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        if (have && (*it)->device != cur) (void)hipSetDevice((*it)->device);
        (*it)->release();
        if (have) (void)hipSetDevice(cur);
        delete *it;
        it = g_pools.erase(it);
    }
}
