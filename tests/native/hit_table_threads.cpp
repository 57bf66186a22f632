// TEST INFRASTRUCTURE: drives the host half of the fused path (csrc/hit_table.cpp: gfm_graph_hit_columns, the threaded
// gfm_graph_hit_columns_start / _wait, gfm_region_labels) on random records, built with -fsanitize=address,undefined and with
// -fsanitize=thread by tests/test_native_sanitizers.py: no out-of-bounds access, no data race between the library's host
// threads and the caller, and the threaded run's columns == the synchronous call's, job by job.
//   hit_table_threads <rounds> <seed>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "grafimo_hip.h"

// the two functions of grafimo_hip.hip that hit_table.cpp links against (the thread-local error slot)
static thread_local std::string g_err;
extern "C" void gfm_set_error_(const char *msg) { g_err = msg ? msg : ""; }
extern "C" const char *gfm_last_error(void) { return g_err.c_str(); }

struct Cols {
    std::vector<int64_t> start, stop, freq, region;
    std::vector<double> score, pvalue, qvalue;
    std::vector<uint8_t> strand, ref, kmers;
    int64_t n = 0;
    void room(size_t rows, int W)
    {
        start.assign(rows, -1); stop.assign(rows, -1); freq.assign(rows, -1); region.assign(rows, -1);
        score.assign(rows, -1); pvalue.assign(rows, -1); qvalue.assign(rows, -1);
        strand.assign(rows, 9); ref.assign(rows, 9); kmers.assign(rows * (size_t)(W + 1), 0);
    }
    bool same(const Cols &o, int W) const
    {
        if (n != o.n) return false;
        const size_t k = (size_t)n;
        if (k == 0) return true;
        auto eq = [k](const auto &a, const auto &b, size_t per = 1) { return std::memcmp(a.data(), b.data(), k * per * sizeof(a[0])) == 0; };
        return eq(start, o.start) && eq(stop, o.stop) && eq(freq, o.freq) && eq(region, o.region) && eq(score, o.score) && eq(pvalue, o.pvalue) &&
               eq(qvalue, o.qvalue) && eq(strand, o.strand) && eq(ref, o.ref) && eq(kmers, o.kmers, (size_t)(W + 1));
    }
};

struct Case {
    int W = 0, L = 0, n_parts = 0;
    uint32_t flags = 0;
    std::vector<double> ptable;
    std::vector<std::vector<gfm_graph_hit_t>> recs;
    std::vector<std::vector<int64_t>> entry_of;
    std::vector<int64_t> region_base, n_recs;
    std::vector<const gfm_graph_hit_t *> rec_p;
    std::vector<const int64_t *> eo_p;
    size_t total = 0;
};

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 6;
    std::mt19937_64 rng(argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1);
    auto below = [&](uint64_t m) { return (int64_t)(rng() % m); };
    long jobs_run = 0, rows = 0;
    for (int round = 0; round < rounds; ++round) {
        const int n_jobs = (int)below(24);
        std::vector<Case> cases((size_t)n_jobs);
        for (Case &c : cases) {
            c.W = 1 + (int)below(GFM_MAX_WIDTH);
            c.L = 1000 * c.W + 1;
            c.ptable.resize((size_t)c.L);
            double p = 1.0;
            for (int i = 0; i < c.L; ++i) {            // non-increasing, with flat stretches (scores that share a p-value)
                c.ptable[(size_t)i] = p;
                if (below(3)) p *= 0.9995;
            }
            c.n_parts = 1 + (int)below(3);
            c.flags = (uint32_t)below(4);
            c.region_base.push_back(0);
            for (int q = 0; q < c.n_parts; ++q) {
                const int n_regions = 1 + (int)below(300);
                const int n = below(5) ? (int)below(round % 3 == 2 ? 12000 : 3000) : 0;      // (>= 4 096 rows: the synchronous call uses threads)
                std::vector<gfm_graph_hit_t> v((size_t)n);
                for (gfm_graph_hit_t &r : v) {
                    std::memset(&r, 0, sizeof r);
                    r.region = (int32_t)below((uint64_t)n_regions);
                    r.w = (int32_t)below(200000);
                    r.q2 = below(1ull << 40);
                    r.score = (int32_t)(c.L / 2 + below((uint64_t)(c.L - c.L / 2)));
                    r.start = below(1ull << 40);
                    r.stop = below(2) ? r.start + c.W : r.start - c.W - below(5);
                    r.freq = below(3) ? 1 + below(5000) : 0;
                    r.qvalue = (double)below(1000) / 1000.0;
                    r.strand = below(2) ? '+' : '-';
                    r.is_ref = (uint8_t)below(2);
                    r.keep = (uint8_t)(below(10) != 0);
                    for (int j = 0; j < c.W; ++j) r.kmer[j] = (uint8_t)"ACGTN"[below(5)];
                }
                c.total += v.size();
                c.recs.push_back(std::move(v));
                std::vector<int64_t> eo((size_t)n_regions);
                for (int64_t &e : eo) e = 10 * q + below(3);
                c.entry_of.push_back(std::move(eo));
                c.region_base.push_back(c.region_base.back() + n_regions);
            }
            for (int q = 0; q < c.n_parts; ++q) {
                c.rec_p.push_back(c.recs[(size_t)q].empty() ? nullptr : c.recs[(size_t)q].data());
                c.n_recs.push_back((int64_t)c.recs[(size_t)q].size());
                c.eo_p.push_back(c.entry_of[(size_t)q].data());
            }
        }
        if (n_jobs > 2 && below(3) == 0) {             // one job with a score outside its table: it fails alone
            Case &c = cases[1];
            if (!c.recs[0].empty()) {
                c.recs[0][0].keep = 1;
                c.recs[0][0].score = c.L;
            }
        }
        std::vector<Cols> want((size_t)n_jobs), got((size_t)n_jobs);
        std::vector<int> want_rc((size_t)n_jobs);
        std::vector<gfm_hit_columns_job_t> jobs((size_t)n_jobs);
        for (int i = 0; i < n_jobs; ++i) {
            Case &c = cases[(size_t)i];
            Cols &w = want[(size_t)i], &g = got[(size_t)i];
            w.room(c.total, c.W);
            g.room(c.total, c.W);
            want_rc[(size_t)i] = gfm_graph_hit_columns(c.ptable.data(), c.L, 37, -12.0, c.W, c.n_parts, c.rec_p.data(), c.n_recs.data(), c.eo_p.data(),
                                                       c.region_base.data(), c.flags, &w.n, w.start.data(), w.stop.data(), w.freq.data(), w.region.data(),
                                                       w.score.data(), w.pvalue.data(), w.qvalue.data(), w.strand.data(), w.ref.data(), w.kmers.data());
            gfm_hit_columns_job_t &j = jobs[(size_t)i];
            std::memset(&j, 0, sizeof j);
            j.h_ptable = c.ptable.data(); j.table_len = c.L; j.scale = 37; j.offset = -12.0; j.width = c.W; j.n_parts = c.n_parts;
            j.h_recs = c.rec_p.data(); j.n_recs = c.n_recs.data(); j.h_entry_of = c.eo_p.data(); j.region_base = c.region_base.data(); j.flags = c.flags;
            j.o_start = g.start.data(); j.o_stop = g.stop.data(); j.o_freq = g.freq.data(); j.o_region = g.region.data(); j.o_score = g.score.data();
            j.o_pvalue = g.pvalue.data(); j.o_qvalue = g.qvalue.data(); j.o_strand = g.strand.data(); j.o_ref = g.ref.data(); j.o_kmers = g.kmers.data();
        }
        gfm_hit_columns_run_t run = nullptr;
        if (gfm_graph_hit_columns_start(jobs.data(), n_jobs, &run) != GFM_OK || !run) {
            std::printf("MISMATCH: start failed: %s\n", gfm_last_error());
            return 1;
        }
        // the caller goes on meanwhile: labels for a run of regions (host work of the same library)
        std::vector<int64_t> s(500), e(500);
        for (int i = 0; i < 500; ++i) { s[(size_t)i] = below(1ull << 40); e[(size_t)i] = s[(size_t)i] + 200; }
        std::vector<char> buf((size_t)gfm_region_labels("chr22", s.data(), e.data(), 500, nullptr, 0));
        if (gfm_region_labels("chr22", s.data(), e.data(), 500, buf.data(), (int64_t)buf.size()) <= 0) return 1;
        // ... and a table of its own, synchronously, while the library's threads are busy with the jobs: large ones find no
        // help (gfm_workers::run_if_idle) and are built by this thread alone -- the same table
        for (int i = 0; i < n_jobs && i < 3; ++i) {
            Case &c = cases[(size_t)i];
            Cols again;
            again.room(c.total, c.W);
            const int rc2 = gfm_graph_hit_columns(c.ptable.data(), c.L, 37, -12.0, c.W, c.n_parts, c.rec_p.data(), c.n_recs.data(), c.eo_p.data(),
                                                  c.region_base.data(), c.flags, &again.n, again.start.data(), again.stop.data(), again.freq.data(),
                                                  again.region.data(), again.score.data(), again.pvalue.data(), again.qvalue.data(),
                                                  again.strand.data(), again.ref.data(), again.kmers.data());
            if (rc2 != want_rc[(size_t)i] || (rc2 == GFM_OK && !again.same(want[(size_t)i], c.W))) { std::printf("MISMATCH: synchronous table %d beside the run\n", i); return 1; }
        }
        const int rc = gfm_graph_hit_columns_wait(run);
        int first_bad = GFM_OK;
        for (int i = 0; i < n_jobs; ++i) {
            got[(size_t)i].n = jobs[(size_t)i].n_out;
            if (jobs[(size_t)i].status != want_rc[(size_t)i]) { std::printf("MISMATCH: job %d status %d, the synchronous call said %d\n", i, jobs[(size_t)i].status, want_rc[(size_t)i]); return 1; }
            if (want_rc[(size_t)i] != GFM_OK) { if (first_bad == GFM_OK) first_bad = want_rc[(size_t)i]; continue; }
            if (!got[(size_t)i].same(want[(size_t)i], cases[(size_t)i].W)) { std::printf("MISMATCH: job %d of round %d\n", i, round); return 1; }
            rows += (long)want[(size_t)i].n;
        }
        if ((rc == GFM_OK) != (first_bad == GFM_OK)) { std::printf("MISMATCH: wait returned %d\n", rc); return 1; }
        if (rc != GFM_OK && !std::strstr(gfm_last_error(), "outside the table")) { std::printf("MISMATCH: message '%s'\n", gfm_last_error()); return 1; }
        jobs_run += n_jobs;
    }
    std::printf("hit table: %d rounds, %ld jobs, %ld rows: threaded == synchronous\n", rounds, jobs_run, rows);
    return 0;
}
