// scan_stream.cpp -- compute_results' numeric core as ONE streamed pass (score_sequences.py:113-157,194-205):
//   TSV parse threads -> pinned chunk buffers -> hipMemcpyAsync on a copy stream -> score kernel per chunk
//   (one histogram, one hit list, row ids global) -> q-value table + selection once at the end -> the hits,
//   with the columns of their rows, back to the host.
// The reference forks `cores` workers that parse and score line by line and then merges pickled lists; here
// the parse threads run ahead of the GPU and nothing but the hits ever comes back.  Buffers (pinned chunk
// slots, device slots, score blocks, tables, hit list) live in a per-device pool that only grows: a second
// call of the same size allocates nothing.
//
// Reference lines cited as file:line are relative to /root/reference/src/grafimo/.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "gfm_tsv_internal.hpp"

#define GFM_API extern "C" __attribute__((visibility("default")))

using gfm_tsv_detail::FileCols;

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

constexpr int kSlots = 2;                          // chunk slots in flight (double buffering)
constexpr int64_t kDefaultChunkRows = 4 << 20;     // rows per chunk (x W bytes pinned + device, per slot)

// Per-device buffers of the streamed scan; they only grow.
struct ScanPool {
    int device = -1;
    hipStream_t copy = nullptr, score = nullptr;
    hipEvent_t copied[kSlots] = {nullptr, nullptr}, scored[kSlots] = {nullptr, nullptr};
    hipEvent_t c0[kSlots] = {nullptr, nullptr}, c1[kSlots] = {nullptr, nullptr};   // H2D timing
    uint8_t *h_pin[kSlots] = {nullptr, nullptr};
    uint8_t *d_kmers[kSlots] = {nullptr, nullptr};
    size_t slot_bytes = 0;
    int64_t block_rows = 0;                 // rows per score block (== chunk rows they were made for)
    std::vector<int32_t *> score_blocks;    // one per chunk index
    uint64_t *d_hist = nullptr;
    double *d_q = nullptr;
    size_t table_len = 0;
    int32_t *d_cutoff = nullptr;
    uint64_t *d_count = nullptr;
    int64_t *d_hits = nullptr;
    int64_t hit_cap = 0;
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
        S_TRY(hipMalloc(&d_cutoff, sizeof(int32_t)));
        S_TRY(hipMalloc(&d_count, sizeof(uint64_t)));
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
    int reserve_hits(int64_t cap)
    {
        if (cap <= hit_cap) return GFM_OK;
        if (d_hits) (void)hipFree(d_hits);
        d_hits = nullptr;
        hit_cap = 0;
        S_TRY(hipMalloc(&d_hits, sizeof(int64_t) * (size_t)cap));
        hit_cap = cap;
        return GFM_OK;
    }
    int score_block(size_t k, int64_t rows, int32_t **out)
    {
        if (rows > block_rows) {   // chunk size grew: the old blocks are too small
            for (auto *p : score_blocks) (void)hipFree(p);
            score_blocks.clear();
            block_rows = rows;
        }
        while (score_blocks.size() <= k) {
            int32_t *p = nullptr;
            S_TRY(hipMalloc(&p, sizeof(int32_t) * (size_t)block_rows));
            score_blocks.push_back(p);
        }
        *out = score_blocks[k];
        return GFM_OK;
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
        for (auto *p : score_blocks) (void)hipFree(p);
        score_blocks.clear();
        if (d_hist) (void)hipFree(d_hist);
        if (d_q) (void)hipFree(d_q);
        if (d_cutoff) (void)hipFree(d_cutoff);
        if (d_count) (void)hipFree(d_count);
        if (d_hits) (void)hipFree(d_hits);
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

struct PoolLease {
    ScanPool *p = nullptr;
    ~PoolLease()
    {
        if (!p) return;
        std::lock_guard<std::mutex> lk(g_pool_mu);
        p->in_use = false;
    }
};

}  // namespace

struct gfm_scan {
    gfm_tsv table;                    // the parsed columns of every file (hits are looked up here)
    int W = 0;
    bool have_q = false;
    std::vector<int64_t> rows;        // hit rows, ascending (global row ids in sorted-file order)
    std::vector<int32_t> scaled;
    std::vector<double> logodds, pvalue, qvalue;
    gfm_scan_stats_t stats{};
};

GFM_API int gfm_scan_tsv(gfm_motif_t m, const char *const *paths, int n_paths, int skip_reverse, int n_threads,
                         double threshold, int on_qvalue, int want_qvalues, int64_t chunk_rows, gfm_scan_t *out,
                         int64_t *n_rows, int64_t *n_hits)
{
    if (!m || !out || !n_rows || !n_hits || (n_paths > 0 && !paths)) return sfail(GFM_ERR_INVALID, "NULL argument");
    *out = nullptr;
    *n_rows = *n_hits = 0;
    if (n_paths < 0) return sfail(GFM_ERR_INVALID, "negative path count");
    if (!(threshold > 0 && threshold <= 1)) return sfail(GFM_ERR_INVALID, "threshold must be in (0, 1]");
    if (on_qvalue && !want_qvalues) return sfail(GFM_ERR_INVALID, "q-value threshold without q-values");
    const int W = gfm_motif_width(m);
    const int L = gfm_motif_table_len(m);
    if (chunk_rows <= 0) chunk_rows = kDefaultChunkRows;
    chunk_rows = (chunk_rows + 255) & ~(int64_t)255;   // whole 256-row score chunks, 16-byte aligned slices

    gfm_scan *sc = new (std::nothrow) gfm_scan();
    if (!sc) return sfail(GFM_ERR_NOMEM, "out of host memory");
    struct Guard {
        gfm_scan *p;
        ~Guard() { delete p; }
    } guard{sc};
    sc->W = W;
    sc->have_q = want_qvalues != 0;
    sc->table.W = W;
    sc->table.files.resize((size_t)n_paths);
    const double t_begin = now_s();

    // ---- parse threads: files are taken in order, so that they also finish roughly in order
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min(nt, std::max(n_paths, 1)));
    std::atomic<int> next{0};
    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> done((size_t)n_paths, 0);
    std::atomic<bool> cancel{false};
    double t_parse_end = t_begin;
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n_paths || cancel.load()) break;
            FileCols &f = sc->table.files[(size_t)i];
            try {
                gfm_tsv_detail::parse_file(paths[i], W, skip_reverse != 0, f);
            } catch (const std::bad_alloc &) {
                f.error = "out of memory";
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                done[(size_t)i] = 1;
                t_parse_end = std::max(t_parse_end, now_s());
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> pool;
    for (int k = 0; k < nt; ++k) pool.emplace_back(work);
    struct Joiner {
        std::vector<std::thread> &pool;
        std::atomic<bool> &cancel;
        ~Joiner()
        {
            cancel.store(true);
            for (auto &th : pool)
                if (th.joinable()) th.join();
        }
    } joiner{pool, cancel};

    // ---- device side
    ScanPool *P = nullptr;
    S_RC(acquire_pool(&P));
    PoolLease lease{P};
    S_RC(P->reserve_slots((size_t)chunk_rows * (size_t)W + 16));
    S_RC(P->reserve_tables((size_t)L));
    S_RC(P->reserve_hits(std::max<int64_t>(P->hit_cap, 1 << 20)));
    const bool fused = !on_qvalue;     // p-value threshold: the cutoff is known before scoring
    int32_t cutoff = GFM_NO_SELECT;
    if (fused) S_RC(gfm_motif_pvalue_cutoff(m, threshold, &cutoff));
    if (want_qvalues) S_TRY(hipMemsetAsync(P->d_hist, 0, sizeof(uint64_t) * (size_t)L, P->score));
    S_TRY(hipMemsetAsync(P->d_count, 0, sizeof(uint64_t), P->score));

    int64_t total_rows = 0, slot_rows = 0;
    size_t n_chunks = 0;
    std::vector<int64_t> chunk_n;          // rows of every submitted chunk
    int slot = 0;
    bool slot_busy[kSlots] = {false, false};
    double h2d_ms = 0.0;
    int64_t h2d_bytes = 0;
    auto wait_slot = [&](int s) -> int {   // the slot's last chunk has been copied AND scored
        if (!slot_busy[s]) return GFM_OK;
        S_TRY(hipEventSynchronize(P->scored[s]));
        float ms = 0.f;
        S_TRY(hipEventElapsedTime(&ms, P->c0[s], P->c1[s]));
        h2d_ms += ms;
        slot_busy[s] = false;
        return GFM_OK;
    };
    auto submit = [&]() -> int {
        if (slot_rows == 0) return GFM_OK;
        const size_t bytes = (size_t)slot_rows * (size_t)W;
        int32_t *d_sc = nullptr;
        S_RC(P->score_block(n_chunks, chunk_rows, &d_sc));
        S_TRY(hipEventRecord(P->c0[slot], P->copy));
        S_TRY(hipMemcpyAsync(P->d_kmers[slot], P->h_pin[slot], bytes, hipMemcpyHostToDevice, P->copy));
        S_TRY(hipEventRecord(P->c1[slot], P->copy));
        S_TRY(hipEventRecord(P->copied[slot], P->copy));
        S_TRY(hipStreamWaitEvent(P->score, P->copied[slot], 0));
        S_RC(gfm_score_kmers(m, P->d_kmers[slot], slot_rows, d_sc, want_qvalues ? P->d_hist : nullptr, cutoff,
                             total_rows, fused ? P->d_hits : nullptr, fused ? P->hit_cap : 0,
                             fused ? P->d_count : nullptr, 0, P->score, nullptr));
        S_TRY(hipEventRecord(P->scored[slot], P->score));
        slot_busy[slot] = true;
        h2d_bytes += (int64_t)bytes;
        chunk_n.push_back(slot_rows);
        total_rows += slot_rows;
        ++n_chunks;
        slot_rows = 0;
        slot = (slot + 1) % kSlots;
        return wait_slot(slot);            // the slot that is filled next must be free again
    };

    for (int i = 0; i < n_paths; ++i) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return done[(size_t)i] != 0; });
        }
        const FileCols &f = sc->table.files[(size_t)i];
        if (!f.error.empty()) return sfail(GFM_ERR_IO, f.error);
        const int64_t rows = (int64_t)f.start.size();
        int64_t at = 0;
        while (at < rows) {
            const int64_t take = std::min(rows - at, chunk_rows - slot_rows);
            std::memcpy(P->h_pin[slot] + (size_t)slot_rows * (size_t)W, f.kmers.data() + (size_t)at * (size_t)W,
                        (size_t)take * (size_t)W);
            slot_rows += take;
            at += take;
            if (slot_rows == chunk_rows) S_RC(submit());
        }
    }
    S_RC(submit());
    for (auto &th : pool) th.join();
    const double t_parsed = t_parse_end;
    sc->table.index_rows();
    if (sc->table.n != total_rows) return sfail(GFM_ERR_IO, "internal error: row count mismatch");

    // ---- tables, selection
    if (total_rows > 0) {
        if (want_qvalues)
            S_RC(gfm_qvalue_table(m, P->d_hist, threshold, on_qvalue, P->d_q, P->d_cutoff, nullptr, 0, P->score));
        auto select_all = [&]() -> int {   // separate selection pass over every score block
            int64_t base = 0;
            for (size_t k = 0; k < n_chunks; ++k) {
                S_RC(gfm_select_hits(m, P->score_blocks[k], chunk_n[k], P->d_cutoff, base, P->d_hits, P->hit_cap,
                                     P->d_count, k == 0 ? GFM_FLAG_RESET_HITS : 0, P->score));
                base += chunk_n[k];
            }
            return GFM_OK;
        };
        if (on_qvalue) S_RC(select_all());
        uint64_t cnt = 0;
        S_TRY(hipMemcpyAsync(&cnt, P->d_count, sizeof cnt, hipMemcpyDeviceToHost, P->score));
        S_TRY(hipStreamSynchronize(P->score));
        if ((int64_t)cnt > P->hit_cap) {   // the list was too short: size it from the count and select again
            S_RC(P->reserve_hits((int64_t)cnt + (int64_t)(cnt >> 3) + 1024));
            if (fused) S_TRY(hipMemcpyAsync(P->d_cutoff, &cutoff, sizeof cutoff, hipMemcpyHostToDevice, P->score));
            S_RC(select_all());
            S_TRY(hipMemcpyAsync(&cnt, P->d_count, sizeof cnt, hipMemcpyDeviceToHost, P->score));
            S_TRY(hipStreamSynchronize(P->score));
            if ((int64_t)cnt > P->hit_cap) return sfail(GFM_ERR_OVERFLOW, "hit list overflow");
        }
        for (int s = 0; s < kSlots; ++s) S_RC(wait_slot(s));
        std::vector<int64_t> packed((size_t)cnt);
        std::vector<double> q;
        if (cnt)
            S_TRY(hipMemcpyAsync(packed.data(), P->d_hits, sizeof(int64_t) * (size_t)cnt, hipMemcpyDeviceToHost, P->score));
        if (want_qvalues && cnt) {
            q.resize((size_t)L);
            S_TRY(hipMemcpyAsync(q.data(), P->d_q, sizeof(double) * (size_t)L, hipMemcpyDeviceToHost, P->score));
        }
        S_TRY(hipStreamSynchronize(P->score));
        std::sort(packed.begin(), packed.end());   // (row << 20 | score): ascending by row
        sc->rows.resize((size_t)cnt);
        sc->scaled.resize((size_t)cnt);
        sc->logodds.resize((size_t)cnt);
        sc->pvalue.resize((size_t)cnt);
        for (size_t i = 0; i < (size_t)cnt; ++i) {
            sc->rows[i] = packed[i] >> GFM_HIT_SCORE_BITS;
            sc->scaled[i] = (int32_t)(packed[i] & ((1ll << GFM_HIT_SCORE_BITS) - 1));
        }
        if (cnt) S_RC(gfm_motif_annotate(m, sc->scaled.data(), (int64_t)cnt, sc->logodds.data(), sc->pvalue.data()));
        if (want_qvalues) {
            sc->qvalue.resize((size_t)cnt);
            for (size_t i = 0; i < (size_t)cnt; ++i) sc->qvalue[i] = q[(size_t)sc->scaled[i]];
        }
    } else {
        S_TRY(hipStreamSynchronize(P->score));
    }
    const double t_end = now_s();
    sc->stats.n_rows = total_rows;
    sc->stats.n_hits = (int64_t)sc->rows.size();
    sc->stats.n_chunks = (int64_t)n_chunks;
    sc->stats.h2d_bytes = h2d_bytes;
    sc->stats.total_s = t_end - t_begin;
    sc->stats.parse_s = t_parsed - t_begin;
    sc->stats.h2d_s = h2d_ms * 1e-3;
    sc->stats.tail_s = t_end - t_parsed;
    sc->stats.parse_threads = nt;
    *n_rows = total_rows;
    *n_hits = (int64_t)sc->rows.size();
    guard.p = nullptr;
    *out = sc;
    return GFM_OK;
}

GFM_API int gfm_scan_hits(gfm_scan_t s, int64_t *rows, int32_t *scaled, double *logodds, double *pvalue,
                          double *qvalue, uint8_t *kmers, int64_t *start, int64_t *stop, uint8_t *strand,
                          int64_t *freq, uint8_t *is_ref, int32_t *name_id)
{
    if (!s) return sfail(GFM_ERR_INVALID, "NULL handle");
    const size_t k = s->rows.size();
    if (rows) std::memcpy(rows, s->rows.data(), k * sizeof(int64_t));
    if (scaled) std::memcpy(scaled, s->scaled.data(), k * sizeof(int32_t));
    if (logodds) std::memcpy(logodds, s->logodds.data(), k * sizeof(double));
    if (pvalue) std::memcpy(pvalue, s->pvalue.data(), k * sizeof(double));
    if (qvalue) {
        if (!s->have_q) return sfail(GFM_ERR_INVALID, "the scan computed no q-values");
        std::memcpy(qvalue, s->qvalue.data(), k * sizeof(double));
    }
    const gfm_tsv &t = s->table;
    size_t fi = 0;
    const size_t W = (size_t)s->W;
    for (size_t i = 0; i < k; ++i) {          // hits ascend by row: walk the files once
        const int64_t r = s->rows[i];
        while (fi + 1 < t.files.size() && t.row_base[fi + 1] <= r) ++fi;
        const FileCols &f = t.files[fi];
        const size_t j = (size_t)(r - t.row_base[fi]);
        if (kmers) std::memcpy(kmers + i * W, f.kmers.data() + j * W, W);
        if (start) start[i] = f.start[j];
        if (stop) stop[i] = f.stop[j];
        if (strand) strand[i] = f.strand[j];
        if (freq) freq[i] = f.freq[j];
        if (is_ref) is_ref[i] = f.is_ref[j];
        if (name_id) name_id[i] = t.remap[fi][(size_t)f.local_name[j]];
    }
    return GFM_OK;
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
