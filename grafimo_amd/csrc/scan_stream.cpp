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
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "gfm_tsv_internal.hpp"
#include "gfm_workers.hpp"

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

constexpr int kSlots = 3;                          // chunk slots: one being staged, one in flight, one spare
constexpr int64_t kDefaultChunkRows = 1 << 20;     // rows per chunk (x W bytes pinned + device, per slot)

// Per-device buffers of the streamed scan; they only grow.
struct ScanPool {
    int device = -1;
    hipStream_t copy = nullptr, score = nullptr;
    hipEvent_t copied[kSlots] = {}, scored[kSlots] = {};
    hipEvent_t c0[kSlots] = {}, c1[kSlots] = {};   // H2D timing
    uint8_t *h_pin[kSlots] = {};
    uint8_t *d_kmers[kSlots] = {};
    size_t slot_bytes = 0;
    int64_t block_rows = 0;                 // rows per score block (== chunk rows they were made for)
    std::vector<int32_t *> score_blocks;    // one per chunk index
    uint64_t *d_hist = nullptr;
    double *d_q = nullptr;
    size_t table_len = 0;
    int32_t *d_cutoff = nullptr;
    uint64_t *d_count = nullptr;
    int64_t *d_hits = nullptr;
    uint64_t *d_cand_count = nullptr;   // q-value threshold: the p < t candidates collected while scoring
    int64_t *d_cand = nullptr;          // (same capacity as d_hits)
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
        S_TRY(hipMalloc(&d_cand_count, sizeof(uint64_t)));
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
        if (d_cand) (void)hipFree(d_cand);
        d_hits = d_cand = nullptr;
        hit_cap = 0;
        S_TRY(hipMalloc(&d_hits, sizeof(int64_t) * (size_t)cap));
        S_TRY(hipMalloc(&d_cand, sizeof(int64_t) * (size_t)cap));
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
        if (d_cand_count) (void)hipFree(d_cand_count);
        if (d_cand) (void)hipFree(d_cand);
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
    const bool trace = std::getenv("GRAFIMO_SCAN_TRACE") != nullptr;   // development aid: phase times to stderr
    auto stamp = [&](const char *what) {
        if (trace) std::fprintf(stderr, "[scan] %8.3f ms  %s\n", (now_s() - t_begin) * 1e3, what);
    };

    // ---- device side
    ScanPool *P = nullptr;
    S_RC(acquire_pool(&P));
    PoolLease lease{P};
    stamp("pool acquired");
    S_RC(P->reserve_slots((size_t)chunk_rows * (size_t)W + 16));
    S_RC(P->reserve_tables((size_t)L));
    S_RC(P->reserve_hits(std::max<int64_t>(P->hit_cap, 1 << 20)));
    // p-value threshold: the cutoff is known before scoring and the score kernel selects the hits.  q-value
    // threshold: q >= p, so the score kernel collects the p < t CANDIDATES the same way, and the selection behind
    // the q-table filters those instead of reading every score again (gfm_select_hits_from).
    const bool fused = !on_qvalue;
    int32_t cutoff = GFM_NO_SELECT;
    stamp("pool sized");
    S_RC(gfm_motif_pvalue_cutoff(m, threshold, &cutoff));
    int64_t *d_sel = fused ? P->d_hits : P->d_cand;             // the list the score kernel appends to
    uint64_t *d_sel_count = fused ? P->d_count : P->d_cand_count;

    // ---- host pipeline.  Worker threads parse the files (taken in path order) AND stage them: as soon as
    // the row counts of all earlier files are known a file's global row offset is fixed, and whichever worker
    // comes by next copies its k-mers into the pinned slot of the chunk(s) they fall into.  (One thread doing
    // all staging copies -- 38 MB out of other cores' caches for 2e6 rows -- took longer than 256 threads
    // needed to parse.)  The calling thread only sequences chunks: chunk k goes to the device once every row
    // of it has been staged; its slot is handed back to the workers when the score kernel has read it.
    const int nt = gfm_tsv_detail::pick_threads(paths, n_paths, n_threads);
    std::mutex mu;
    std::condition_variable cv_work, cv_main;
    int next_parse = 0;                      // next file to parse
    int next_assign = 0;                     // files [0, next_assign) have their row offset
    int next_stage = 0;                      // next file to stage (offsets are assigned in order, so a counter)
    int staged_files = 0;
    int64_t assigned_rows = 0;
    std::vector<char> counted((size_t)n_paths, 0);
    std::vector<int64_t> file_off((size_t)n_paths, 0);
    std::vector<int64_t> chunk_staged;       // rows staged so far per chunk
    int64_t released = 0;                    // chunks whose slot may be written again: chunk k needs k < released + kSlots
    bool failed = false;
    std::string fail_msg;
    double t_parse_end = t_begin;
    auto work = [&]() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            if (failed) return;
            if (next_stage < next_assign) {                     // stage a file whose offset is known
                const int i = next_stage++;
                const FileCols &f = sc->table.files[(size_t)i];
                const int64_t rows = (int64_t)f.start.size();
                int64_t at = 0;
                while (at < rows) {
                    const int64_t g = file_off[(size_t)i] + at;
                    const int64_t k = g / chunk_rows, in = g % chunk_rows;
                    const int64_t take = std::min(rows - at, chunk_rows - in);
                    cv_work.wait(lk, [&] { return failed || k < released + kSlots; });
                    if (failed) return;
                    lk.unlock();
                    std::memcpy(P->h_pin[k % kSlots] + (size_t)in * (size_t)W, f.kmers.data() + (size_t)at * (size_t)W,
                                (size_t)take * (size_t)W);
                    lk.lock();
                    if ((int64_t)chunk_staged.size() <= k) chunk_staged.resize((size_t)k + 1, 0);
                    chunk_staged[(size_t)k] += take;
                    at += take;
                    cv_main.notify_one();
                }
                ++staged_files;
                cv_main.notify_one();
                continue;
            }
            if (next_parse < n_paths) {                          // parse the next file
                const int i = next_parse++;
                FileCols &f = sc->table.files[(size_t)i];
                lk.unlock();
                try {
                    gfm_tsv_detail::parse_file(paths[i], W, skip_reverse != 0, f);
                } catch (const std::bad_alloc &) {
                    f.error = "out of memory";
                }
                lk.lock();
                if (!f.error.empty()) {
                    if (!failed) fail_msg = f.error;
                    failed = true;
                    cv_work.notify_all();
                    cv_main.notify_all();
                    return;
                }
                counted[(size_t)i] = 1;
                t_parse_end = std::max(t_parse_end, now_s());
                while (next_assign < n_paths && counted[(size_t)next_assign]) {
                    file_off[(size_t)next_assign] = assigned_rows;
                    assigned_rows += (int64_t)sc->table.files[(size_t)next_assign].start.size();
                    ++next_assign;
                }
                cv_work.notify_all();
                cv_main.notify_one();
                continue;
            }
            if (staged_files >= n_paths || next_stage >= n_paths) return;   // nothing left for this thread to take
            cv_work.wait(lk);
        }
    };
    stamp("pipeline set up");
    // the workers come from the process-wide crew (gfm_workers.hpp): starting threads per call cost more than
    // the parsing they did
    struct Crew {
        gfm_workers::Run run;
        std::mutex &mu;
        std::condition_variable &cv;
        bool &failed;
        ~Crew()
        {
            {
                std::lock_guard<std::mutex> lk(mu);
                failed = true;          // whoever still waits gives up (after a normal run nobody does)
            }
            cv.notify_all();
            run.wait();
        }
    } crew{{}, mu, cv_work, failed};
    crew.run.start(nt, work);
    stamp("workers started");
    // (behind the start of the workers: the two memsets take 0.7 ms of host time to enqueue)
    if (want_qvalues) S_TRY(hipMemsetAsync(P->d_hist, 0, sizeof(uint64_t) * (size_t)L, P->score));
    S_TRY(hipMemsetAsync(P->d_count, 0, sizeof(uint64_t), P->score));
    S_TRY(hipMemsetAsync(P->d_cand_count, 0, sizeof(uint64_t), P->score));
    stamp("buffers cleared");

    int64_t total_rows = 0;
    size_t n_chunks = 0;
    std::vector<int64_t> chunk_n;          // rows of every submitted chunk
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
            released = k + 1;
        }
        cv_work.notify_all();
        return GFM_OK;
    };
    for (int64_t k = 0;; ++k) {
        int64_t rows_k = 0;
        {   // wait until chunk k is fully staged (or turns out not to exist)
            std::unique_lock<std::mutex> lk(mu);
            cv_main.wait(lk, [&] {
                if (failed) return true;
                const int64_t have = (int64_t)chunk_staged.size() > k ? chunk_staged[(size_t)k] : 0;
                if (have == chunk_rows) return true;
                if (next_assign == n_paths)                 // every row count is known: the last chunk may be short
                    return have == std::min(chunk_rows, std::max<int64_t>(0, assigned_rows - k * chunk_rows));
                return false;
            });
            if (failed) return sfail(GFM_ERR_IO, fail_msg);
            rows_k = (int64_t)chunk_staged.size() > k ? chunk_staged[(size_t)k] : 0;
        }
        if (rows_k == 0) break;
        stamp("chunk staged");
        const int slot = (int)(k % kSlots);
        const size_t bytes = (size_t)rows_k * (size_t)W;
        int32_t *d_sc = nullptr;
        S_RC(P->score_block((size_t)k, chunk_rows, &d_sc));
        S_TRY(hipEventRecord(P->c0[slot], P->copy));
        S_TRY(hipMemcpyAsync(P->d_kmers[slot], P->h_pin[slot], bytes, hipMemcpyHostToDevice, P->copy));
        S_TRY(hipEventRecord(P->c1[slot], P->copy));
        S_TRY(hipEventRecord(P->copied[slot], P->copy));
        S_TRY(hipStreamWaitEvent(P->score, P->copied[slot], 0));
        S_RC(gfm_score_kmers(m, P->d_kmers[slot], rows_k, d_sc, want_qvalues ? P->d_hist : nullptr, cutoff, total_rows,
                             d_sel, P->hit_cap, d_sel_count, 0, P->score, nullptr));
        S_TRY(hipEventRecord(P->scored[slot], P->score));
        h2d_bytes += (int64_t)bytes;
        chunk_n.push_back(rows_k);
        total_rows += rows_k;
        ++n_chunks;
        if (k + 1 >= kSlots) S_RC(release_chunk(k + 1 - kSlots));   // the slot chunk k+1 will be staged into
        if (rows_k < chunk_rows) break;                             // a short chunk is the last one
    }
    for (int64_t k = std::max<int64_t>(0, (int64_t)n_chunks - (kSlots - 1)); k < (int64_t)n_chunks; ++k)
        S_RC(release_chunk(k));
    stamp("chunks submitted and released");
    crew.run.wait();
    stamp("workers done");
    {
        std::lock_guard<std::mutex> lk(mu);
        if (failed) return sfail(GFM_ERR_IO, fail_msg);
    }
    const double t_parsed = t_parse_end;
    if (trace) std::fprintf(stderr, "[scan] %8.3f ms  (last file parsed)\n", (t_parsed - t_begin) * 1e3);
    sc->table.index_rows();
    stamp("rows indexed");
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
        if (on_qvalue) {
            uint64_t ccnt = 0;
            S_TRY(hipMemcpyAsync(&ccnt, P->d_cand_count, sizeof ccnt, hipMemcpyDeviceToHost, P->score));
            S_TRY(hipStreamSynchronize(P->score));
            if ((int64_t)ccnt <= P->hit_cap)      // the candidates are complete: filter them (the gated pass over
                S_RC(gfm_select_hits_from(m, P->score_blocks[0], chunk_n[0], P->d_cutoff, 0, P->d_cand, P->hit_cap,   // the scores exits at once)
                                          P->d_cand_count, P->d_hits, P->hit_cap, P->d_count, P->score));
            else
                S_RC(select_all());
        }
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
        std::vector<int64_t> packed((size_t)cnt);
        std::vector<double> q;
        if (cnt)
            S_TRY(hipMemcpyAsync(packed.data(), P->d_hits, sizeof(int64_t) * (size_t)cnt, hipMemcpyDeviceToHost, P->score));
        if (want_qvalues && cnt) {
            q.resize((size_t)L);
            S_TRY(hipMemcpyAsync(q.data(), P->d_q, sizeof(double) * (size_t)L, hipMemcpyDeviceToHost, P->score));
        }
        S_TRY(hipStreamSynchronize(P->score));
        stamp("hits and q-table on the host");
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
    stamp("hits annotated");
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
