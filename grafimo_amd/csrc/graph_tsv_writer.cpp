// graph_tsv_writer.cpp -- the files scan_graph leaves for GRAFIMO's own compute_results, written by host threads from the
// rows of gfm_graph_emit: out_dir/width_W/CHR_S-E.tsv (extract_regions.py:165-170,180), seven tab-separated columns per
// row as `vg find -p CHR:S-E -x XG -H GBWT -K W -E` prints them (consumed by score_sequences.py:273-321):
//     REGION  KMER  CHR:START(+|-)  CHR:STOP(+|-)  COUNT  ref|non.ref  NODE(+|-),NODE(+|-),...
// Rounds 1-4 formatted these rows in a Python loop (4 us a row: 24 s for the bench's 6e6 rows); what made it slow is
// column 7, the path of node ids, which needs for every row the walk its k-mer spells -- here re-derived per WINDOW: the
// layouts of a window (one for a plain window; for a window that touches an insertion / deletion the vectors of decisions in
// the enumeration order of the kernels, graph_extract.hip) are enumerated once, each as its list of node ids with the slots
// of its SNP sites marked; a row's walk number then only picks the layout and patches the alternate alleles' nodes in.
// Part of libgrafimo_hip.so.
#include "gfm_graph_host.hpp"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstring>

#include <fcntl.h>
#include <unistd.h>

#include "gfm_workers.hpp"
#include "grafimo_hip.h"

namespace gfm_host {

void HostGraph::build_nodes()
{
    const int n = n_sites();
    std::vector<long long> c;
    c.reserve(2 * (size_t)n + 2);
    c.push_back(0);
    c.push_back(ref_len);
    for (int i = 0; i < n; ++i) {
        const long long p = pos[(size_t)i];
        if (del_len[(size_t)i] > 0) { c.push_back(p + 1); c.push_back(p + del_len[(size_t)i] + 1); }
        else if (ins_len[(size_t)i] > 0) c.push_back(p + 1);
        else { c.push_back(p); c.push_back(p + 1); }
    }
    std::sort(c.begin(), c.end());
    c.erase(std::unique(c.begin(), c.end()), c.end());
    c.erase(std::remove_if(c.begin(), c.end(), [&](long long x) { return x < 0 || x > ref_len; }), c.end());
    cuts.swap(c);
    const size_t iv = cuts.size() - 1;
    first.assign(iv, 0);
    site_of.assign(iv, -1);
    ins_first.assign((size_t)n, -1);
    long long nid = 1;
    int s = 0;                                     // first site at or behind the interval's start
    for (size_t j = 0; j < iv; ++j) {
        const long long b = cuts[j], e = cuts[j + 1];
        while (s < n && pos[(size_t)s] < b) ++s;
        int snp = -1;
        if (e - b == 1)
            for (int k = s; k < n && pos[(size_t)k] == b; ++k)
                if (del_len[(size_t)k] == 0 && ins_len[(size_t)k] == 0) snp = k;
        first[j] = nid;
        if (snp >= 0) {                            // alternates first[j] .. first[j] + n_alts - 1, the reference allele after them
            site_of[j] = snp;
            nid += (long long)n_alts[(size_t)snp] + 1;
        } else {
            nid += (e - b + kNodeMax - 1) / kNodeMax;
        }
        // the insertions anchored on the interval's last base, in site order
        for (int k = s; k < n && pos[(size_t)k] < e; ++k)
            if (pos[(size_t)k] == e - 1 && ins_len[(size_t)k] > 0) {
                ins_first[(size_t)k] = nid;
                nid += (ins_len[(size_t)k] + kNodeMax - 1) / kNodeMax;
            }
    }
}

namespace {

using clk = std::chrono::steady_clock;

struct Step { int a, b; };            // b < 0: the reference base at a; else base b of the insertion site a
struct Slot { int idx, radix; long long alt_base; };     // node `idx` of the layout is SNP site's: allele k > 0 -> alt_base + k - 1
struct Layout { long long cum_end; int node_off, n_nodes, slot_off, n_slots; };

struct WinCache {
    long long p = -1;
    int region = -1;
    std::vector<Layout> lay;
    std::vector<long long> nodes;
    std::vector<Slot> slots;
    std::vector<Step> plan;           // scratch of the enumeration
    void reset(long long p_, int r_) { p = p_; region = r_; lay.clear(); nodes.clear(); slots.clear(); }
};

int lower_site(const HostGraph &g, long long x)
{
    return (int)(std::lower_bound(g.pos.begin(), g.pos.end(), x, [](int a, long long b) { return (long long)a < b; }) - g.pos.begin());
}

// GraphIndex.touches_deletion: a deletion or an insertion inside [p, p + W), an insertion anchored at p - 1, a start on
// deleted bases -- the windows whose walks the kernels enumerate layout by layout
bool touches_indel(const HostGraph &g, long long p, int W)
{
    const int i0 = lower_site(g, p), n = g.n_sites();
    for (int i = i0; i < n && g.pos[(size_t)i] < p + W; ++i)
        if (g.del_len[(size_t)i] || g.ins_len[(size_t)i]) return true;
    for (int k = i0 - 1; k >= 0 && g.pos[(size_t)k] == p - 1; --k)
        if (g.ins_len[(size_t)k] > 0) return true;
    return g.max_reach[(size_t)i0] >= p;
}

// one layout (W steps) -> its node list with the SNP slots marked
void add_layout(const HostGraph &g, WinCache &wc, const Step *steps, int W)
{
    Layout L{};
    L.node_off = (int)wc.nodes.size();
    L.slot_off = (int)wc.slots.size();
    long long prod = 1, last = -1;
    long j = -1;                                   // interval of the last reference base (they ascend along a walk)
    for (int k = 0; k < W; ++k) {
        const Step st = steps[k];
        long long nid;
        if (st.b >= 0) {
            nid = g.ins_first[(size_t)st.a] + st.b / kNodeMax;
        } else {
            const long long x = st.a;
            if (j < 0) j = (long)(std::upper_bound(g.cuts.begin(), g.cuts.end(), x) - g.cuts.begin()) - 1;
            while (g.cuts[(size_t)j + 1] <= x) ++j;
            const int si = g.site_of[(size_t)j];
            if (si >= 0) {
                nid = g.first[(size_t)j] + g.n_alts[(size_t)si];
                wc.slots.push_back(Slot{(int)wc.nodes.size() - L.node_off, 1 + (int)g.n_alts[(size_t)si], g.first[(size_t)j]});
                prod *= 1 + (long long)g.n_alts[(size_t)si];
                wc.nodes.push_back(nid);           // (a SNP's node is one base long: never merged with a neighbour)
                last = nid;
                continue;
            }
            nid = g.first[(size_t)j] + (x - g.cuts[(size_t)j]) / kNodeMax;
        }
        if (nid != last) { wc.nodes.push_back(nid); last = nid; }
    }
    L.n_nodes = (int)wc.nodes.size() - L.node_off;
    L.n_slots = (int)wc.slots.size() - L.slot_off;
    L.cum_end = (wc.lay.empty() ? 0 : wc.lay.back().cum_end) + prod;
    wc.lay.push_back(L);
}

// The layouts of a window that touches an insertion / deletion, in the kernels' enumeration order (graph_extract.hip,
// oracle/extract_oracle.py): behind the base at x -- read insertion k anchored at x? (a yes ends the site), then: jump the
// deletion anchored at x? for every deletion anchored there (a yes ends the site); 0 before 1; starts inside an insertion
// anchored at p - 1 follow the plain start (site order, offsets ascending).  A walk must end inside its region.
struct LayoutEnum {
    const HostGraph &g;
    WinCache &wc;
    int W;
    long long limit;
    void emit() { add_layout(g, wc, wc.plan.data(), W); }
    void layouts(long long x)
    {
        if (x >= g.ref_len) return;
        wc.plan.push_back(Step{(int)x, -1});
        if ((int)wc.plan.size() == W) {
            if (x + 1 <= limit) emit();
            wc.plan.pop_back();
            return;
        }
        const int s0 = lower_site(g, x);
        after_ins(x, s0, s0);
        wc.plan.pop_back();
    }
    // insertion sites at x from k on (sites at one position: substitution, insertions, deletions)
    void after_ins(long long x, int s0, int k)
    {
        const int n = g.n_sites();
        while (k < n && g.pos[(size_t)k] == x && g.ins_len[(size_t)k] <= 0) {
            if (g.del_len[(size_t)k] > 0) break;               // insertions come before the deletions
            ++k;
        }
        if (k >= n || g.pos[(size_t)k] != x || g.ins_len[(size_t)k] <= 0) { after_del(x, s0); return; }
        after_ins(x, s0, k + 1);                               // 0: do not read insertion k
        const int take = std::min(g.ins_len[(size_t)k], W - (int)wc.plan.size());
        for (int t = 0; t < take; ++t) wc.plan.push_back(Step{k, t});
        if ((int)wc.plan.size() == W) { if (x + 1 <= limit) emit(); }
        else layouts(x + 1);
        wc.plan.resize(wc.plan.size() - (size_t)take);
    }
    void after_del(long long x, int k)
    {
        const int n = g.n_sites();
        while (k < n && g.pos[(size_t)k] == x && g.del_len[(size_t)k] <= 0) ++k;
        if (k >= n || g.pos[(size_t)k] != x) { layouts(x + 1); return; }
        after_del(x, k + 1);                                   // 0: along the reference
        layouts(x + g.del_len[(size_t)k] + 1);                 // 1: jump
    }
    void window(long long p)
    {
        wc.plan.clear();
        layouts(p);
        const int i0 = lower_site(g, p);
        int k = i0 - 1;
        while (k >= 0 && g.pos[(size_t)k] == p - 1) --k;
        for (++k; k < i0; ++k) {
            const int len = g.ins_len[(size_t)k];
            if (len <= 0) continue;
            for (int t = 0; t < len; ++t) {
                const int take = std::min(len - t, W);
                wc.plan.clear();
                for (int j = 0; j < take; ++j) wc.plan.push_back(Step{k, t + j});
                if (take == W) { if (p <= limit) emit(); }
                else layouts(p);
            }
        }
        wc.plan.clear();
    }
};

void build_window(HostGraph &g, WinCache &wc, long long p, int region, int W, long long limit)
{
    wc.reset(p, region);
    if (!touches_indel(g, p, W)) {
        wc.plan.clear();
        for (int j = 0; j < W; ++j) wc.plan.push_back(Step{(int)(p + j), -1});
        add_layout(g, wc, wc.plan.data(), W);
        return;
    }
    LayoutEnum en{g, wc, W, std::min(limit, g.ref_len)};
    en.window(p);
}

struct Out {
    std::vector<char> buf;
    size_t n = 0;
    void need(size_t k) { if (n + k > buf.size()) buf.resize(std::max(buf.size() * 2, n + k + (1u << 16))); }
    void bytes(const void *p, size_t k) { std::memcpy(buf.data() + n, p, k); n += k; }
    void ch(char c) { buf[n++] = c; }
    void num(long long v)
    {
        char tmp[24];
        int k = 0;
        unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
        do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) buf[n++] = '-';
        while (k) buf[n++] = tmp[--k];
    }
};

struct Shared {
    HostGraph &g;
    const WriteJob &job;
    const RowChunk &rows;
    const std::vector<std::pair<long long, long long>> &pieces;
    Shared(HostGraph &g_, const WriteJob &j_, const RowChunk &r_, const std::vector<std::pair<long long, long long>> &p_)
        : g(g_), job(j_), rows(r_), pieces(p_) {}
    std::atomic<size_t> next{0};
    std::atomic<long long> bytes{0}, files{0};
    std::atomic<int> failed{0};
    std::mutex err_mu;
    std::string err;
    void fail(const std::string &m)
    {
        std::lock_guard<std::mutex> lk(err_mu);
        if (!failed.exchange(1)) err = m;
    }
};

void work(Shared &sh)
{
    HostGraph &g = sh.g;
    const WriteJob &job = sh.job;
    const RowChunk &rw = sh.rows;
    const int W = job.W;
    const size_t chrom_len = std::strlen(job.chrom);
    WinCache wc;
    Out out;
    std::vector<long long> path;
    for (;;) {
        const size_t pi = sh.next.fetch_add(1);
        if (pi >= sh.pieces.size() || sh.failed.load()) return;
        const long long a = sh.pieces[pi].first, b = sh.pieces[pi].second;
        const int r = rw.region[a];
        if (r < 0 || r >= job.n_regions) { sh.fail("a row names a region outside the call's regions"); return; }
        const char *label = job.labels[r];
        const size_t label_len = std::strlen(label);
        out.n = 0;
        for (long long i = a; i < b; ++i) {
            const char sg = (char)rw.strand[i];
            const long long st = rw.start[i], sp = rw.stop[i];
            out.need(label_len + (size_t)W + 2 * (chrom_len + 24) + 48);
            out.bytes(label, label_len); out.ch('\t');
            out.bytes(rw.kmers + (size_t)i * (size_t)W, (size_t)W); out.ch('\t');
            out.bytes(job.chrom, chrom_len); out.ch(':'); out.num(st); out.ch(sg); out.ch('\t');
            out.bytes(job.chrom, chrom_len); out.ch(':'); out.num(sp); out.ch(sg); out.ch('\t');
            out.num(rw.freq[i]); out.ch('\t');
            if (rw.is_ref[i]) out.bytes("ref", 3); else out.bytes("non.ref", 7);
            out.ch('\t');
            if (job.node_paths) {
                const long long p = sg == '+' ? st : sp;
                if (wc.p != p || wc.region != r) build_window(g, wc, p, r, W, job.region_stop[r]);
                long long q = rw.walk[i];
                // the layout that holds walk q (layout-major numbering)
                size_t li = 0;
                if (wc.lay.size() > 1) {
                    size_t lo = 0, hi = wc.lay.size() - 1;
                    while (lo < hi) {
                        const size_t mid = (lo + hi) / 2;
                        if (wc.lay[mid].cum_end > q) hi = mid; else lo = mid + 1;
                    }
                    li = lo;
                }
                if (wc.lay.empty() || q < 0 || q >= wc.lay.back().cum_end) {
                    sh.fail("a row's walk number lies outside its window's walks (rows and graph do not belong together)");
                    return;
                }
                const Layout &L = wc.lay[li];
                if (li) q -= wc.lay[li - 1].cum_end;
                path.assign(wc.nodes.begin() + L.node_off, wc.nodes.begin() + L.node_off + L.n_nodes);
                for (int s = L.n_slots - 1; s >= 0 && q; --s) {          // mixed radix, the last SNP fastest
                    const Slot &sl = wc.slots[(size_t)L.slot_off + (size_t)s];
                    const long long al = q % sl.radix;
                    q /= sl.radix;
                    if (al) path[(size_t)sl.idx] = sl.alt_base + al - 1;
                }
                out.need(path.size() * 22 + 2);
                if (sg == '+') for (size_t k = 0; k < path.size(); ++k) { out.num(path[k]); out.ch('+'); out.ch(','); }
                else for (size_t k = path.size(); k-- > 0;) { out.num(path[k]); out.ch('-'); out.ch(','); }
            }
            out.ch('\n');
        }
        const bool append = job.seen[r] != 0;
        const int fd = ::open(job.paths[r], O_WRONLY | O_CREAT | (append ? O_APPEND : O_TRUNC), 0644);
        if (fd < 0) { sh.fail(std::string("cannot open ") + job.paths[r] + ": " + std::strerror(errno)); return; }
        size_t done = 0;
        while (done < out.n) {
            const ssize_t w = ::write(fd, out.buf.data() + done, out.n - done);
            if (w < 0) {
                if (errno == EINTR) continue;
                sh.fail(std::string("cannot write ") + job.paths[r] + ": " + std::strerror(errno));
                ::close(fd);
                return;
            }
            done += (size_t)w;
        }
        if (::close(fd) != 0) { sh.fail(std::string("cannot close ") + job.paths[r] + ": " + std::strerror(errno)); return; }
        if (!append) sh.files.fetch_add(1);
        job.seen[r] = 1;
        sh.bytes.fetch_add((long long)out.n);
    }
}

}  // namespace

int write_chunk(HostGraph &g, const WriteJob &job, const RowChunk &rows, WriteStats &stats, std::string &err)
{
    if (rows.n <= 0) return GFM_OK;
    if (job.node_paths) std::call_once(g.nodes_once, [&] { g.build_nodes(); });
    const auto t0 = clk::now();
    // pieces: runs of one region (rows are region-major)
    std::vector<std::pair<long long, long long>> pieces;
    long long a = 0;
    for (long long i = 1; i <= rows.n; ++i)
        if (i == rows.n || rows.region[i] != rows.region[a]) { pieces.emplace_back(a, i); a = i; }
    Shared sh(g, job, rows, pieces);
    const int n_threads = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, job.threads), pieces.size()));
    gfm_workers::run(n_threads, [&] { work(sh); });
    stats.n_rows += rows.n;
    stats.bytes += sh.bytes.load();
    stats.n_files += sh.files.load();
    stats.format_s += std::chrono::duration<double>(clk::now() - t0).count();
    stats.threads = std::max(stats.threads, n_threads);
    if (sh.failed.load()) { err = sh.err; return GFM_ERR_IO; }
    return GFM_OK;
}

}  // namespace gfm_host
