// graph_extract.hip -- k-mer extraction over a SNP variation graph on the MI355X (gfx950): the
// rows `vg find -p CHR:S-E -x XG -H GBWT -K W -E` hands to GRAFIMO
// (src/grafimo/extract_regions.py:180,225; consumed at src/grafimo/score_sequences.py:273-307),
// produced straight into the row-major uint8 [N][W] matrix the score kernel reads, with the
// per-row metadata beside it -- no TSV round trip.
//
// Graph = linear reference + biallelic/multi-allelic SNP sites + (optionally) one bitset of
// haplotypes per alternate allele.  A "walk" is one choice of allele at every site of a window
// [p, p+W); per walk two rows come out (forward, reverse complement on '-' with start/stop
// swapped); freq = number of haplotypes that carry every allele of the walk (GBWT count of -H).
// Semantics pinned by the reference's expected_seqs.tsv through oracle/extract_oracle.py; what is
// not pinned is listed there.
//
// Both kernels are byte/bit work bound by HBM/L2 reads of the haplotype bitsets
// (sites-in-window x ceil(H/64) x 8 B per walk, re-used across the W overlapping windows of a site).

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "grafimo_hip.h"
#include "gfm_graph_host.hpp"

#define GFM_API extern "C" __attribute__((visibility("default")))
extern "C" void gfm_set_error_(const char *msg);   // thread-local slot of grafimo_hip.hip

namespace {

int gfail(int code, const std::string &msg)
{
    gfm_set_error_(msg.c_str());
    return code;
}

#define GX_TRY(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return gfail(GFM_ERR_HIP, std::string(#expr " failed: ") + hipGetErrorString(e_));    \
    } while (0)

constexpr int kMaxAlts = 3;
constexpr int kCountThreads = 256;
constexpr int kEmitThreads = 256;                    // walks (threads) per workgroup of the emit kernel
// walks of one window that the MATERIALISING path can hold: what one plan can (2^30 walks = 2^31 rows: the row indices of
// its kernels are 32-bit, and 2^31 rows of W + 34 bytes are what one MI355X's HBM holds at W = 64).  A window beyond it
// has no rows to give -- the fused path (gfm_graph_score) scores up to 2^40 walks per window without writing any.
constexpr long long kMaxWalksPerWindow = 1ll << 30;

// One site as the walk simulation reads it: one 16-byte load instead of four loads from four arrays.  The array holds
// kSitePad records behind the last site whose pos is kNoSitePos, so a scan for "the sites at x" ends by itself.
struct SiteRec { int pos, del_len, ins_len, n_alts; };
constexpr int kNoSitePos = 0x7fffffff;
constexpr int kSiteCache = 8;        // records of the window's first sites an emit thread keeps in LDS
constexpr int kSitePad = kSiteCache + 1;

struct GraphDev {
    const SiteRec *site_rec;    // [n_sites + kSitePad]
    const uint8_t *ref;         // [ref_len]
    long long ref_len;
    int n_sites;
    const int *pos;             // [n_sites] non-decreasing, 0-based: SNP position / anchor of a deletion
    const uint8_t *n_alts;      // [n_sites] SNP: 1..3 alternate bases; deletion: 1
    const uint8_t *alt_bases;   // [n_sites][3]
    const unsigned long long *alt_bits;   // [n_sites][3][hw] or nullptr (deletion carriers in slot 0)
    int n_hap, hw;
    const int *del_len;         // [n_sites] 0 for a SNP, else the number of deleted bases after the anchor
    int n_dels;
    const int *prev_del;        // [n_sites + 1] index of the last deletion among sites [0, i), or -1
    const long long *max_reach; // [n_sites + 1] last reference position removed by ANY deletion among sites [0, i), or -1
                                // (deletions may overlap: several lengths at one anchor, anchors inside another's span)
    // insertions (round 2; semantics unpinned, stated in oracle/extract_oracle.py): a site with ins_len > 0 holds
    // ins_len bases behind its anchor `pos`; several may share an anchor (after the SNP, before the deletion)
    const int *ins_len;         // [n_sites]
    const int *ins_off;         // [n_sites] offset of the inserted bases in ins_bases
    const uint8_t *ins_bases;
    int n_ins;
    // haplotype counts of allele combinations over CONSECUTIVE sites, built once per graph (allele 0 = none of the
    // site's alternates): pair_count[i][a][b] = haplotypes with allele a at site i and b at site i + 1,
    // triple_count[i][a][b][c] the same over sites i, i + 1, i + 2.  The sites one window holds are consecutive, so
    // a walk through two or three of them looks its count up instead of ANDing bitsets (nullptr: no haplotypes).
    const int *pair_count;      // [n_sites][4][4]
    const int *triple_count;    // [n_sites][4][4][4]
    // site_rec again with the alternate bases in the upper bytes of n_alts (count | a0 << 8 | a1 << 16 | a2 << 24): what the
    // fused kernels stage per tile, one 16-byte load per site instead of that plus three byte loads
    const SiteRec *site_pk;     // [n_sites + kSitePad]
};

// -------------------------------------------------------------------------------------------
// Walks through windows that touch a deletion.  While W bases are consumed from p a walk decides, at
// the anchor of every deletion it meets with bases still to go, whether to go on along the reference
// (0) or to jump the deleted bases (1): that vector of jumps is its LAYOUT -- the reference positions
// it uses -- and on a layout the SNP alleles are free, a plain mixed-radix product.  The walks of a
// window are ordered layout-major: layouts in lexicographic order of their jump vectors (an odometer
// over simulate()), inside a layout the allele digits with the last SNP varying fastest.  So counting
// a window costs one pass per layout, not per walk, and walk q is found by skipping whole layouts.
// oracle/extract_oracle.py enumerate_region_graph produces the same order.
constexpr int kMaxDecisions = 24;      // jumps decided by one walk
constexpr int kMaxConstraints = 96;    // allele constraints of one walk: <= 64 substitution sites + decisions + skipped / covering deletions
struct WalkState {
    int nd = 0;
    long long last = 0;                       // reference position of the last base
    unsigned choice = 0;                      // bit d: 0 / 1 at the d-th decision (a register: as a byte array it lived in
                                              // scratch memory, and every read waited for the scratch stores before it)
};
enum { WALK_OK = 0, WALK_DEAD = 1, WALK_OVERFLOW = 2 };

struct NoVisitor {
    static constexpr bool kWantsBases = false;     // stretches without a site are skipped in one step
    __device__ void base(int, long long, int, int, int) {}
    __device__ void ins_base(int, int, int) {}
    __device__ void took(int) {}
    __device__ void passed(int) {}
};

// One mixed-radix digit (radix 1..4: the alleles of a site) off a walk number, last site first.  The compiler's 64-bit
// division is a hundred instructions; walk numbers nearly always fit 32 bits, and a radix of 2 or 4 is a shift, 3 a multiply.
// (graph_score_kernel spent about half of its vector instructions in `rest % nall; rest /= nall`.)
__device__ inline int take_digit(unsigned long long &rest, int nall)
{
    if (rest <= 0xffffffffull) {
        const unsigned r = (unsigned)rest;
        const unsigned q3 = __umulhi(r, 0xAAAAAAABu) >> 1;
        const unsigned q = nall == 1 ? r : (nall == 2 ? r >> 1 : (nall == 3 ? q3 : r >> 2));
        rest = q;
        return (int)(r - q * (unsigned)nall);
    }
    const int d = (int)(rest % (unsigned long long)nall);
    rest /= (unsigned long long)nall;
    return d;
}

// ... and where the walk number is known to fit 32 bits (graph_score_kernel's own walks: at most kHeavyWalks per window)
__device__ inline int take_digit32(unsigned &rest, int nall)
{
    const unsigned r = rest;
    const unsigned q3 = __umulhi(r, 0xAAAAAAABu) >> 1;
    const unsigned q = nall == 1 ? r : (nall == 2 ? r >> 1 : (nall == 3 ? q3 : r >> 2));
    rest = q;
    return (int)(r - q * (unsigned)nall);
}

// Inclusive prefix sum over the wavefront's 64 lanes in the vector ALU alone (DPP: shifts inside the rows of 16 lanes, then the
// last lane of a row broadcast into the rows behind it): __shfl_up goes through the LDS unit, six dependent round trips per tile.
__device__ __forceinline__ int wave_prefix_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return v;
}

// A walk may start inside an insertion anchored at p - 1 (start coordinate p): `pre_site` = that site (or -1),
// `pre_t` = offset of its first base inside the inserted string.  next_start() steps through the starts of
// window p in enumeration order: the plain start, then per insertion anchored at p - 1 (site order) t = 0, 1, ...
struct WalkStart { int site = -1, t = 0; };
__device__ inline bool next_start(const GraphDev &g, long long p, int i0, WalkStart &ws)
{
    if (g.n_ins == 0) return false;
    int k;
    if (ws.site < 0) {
        k = i0 - 1;
        while (k >= 0 && g.pos[k] == p - 1) --k;      // first site anchored at p - 1
        ++k;
    } else {
        if (ws.t + 1 < g.ins_len[ws.site]) { ++ws.t; return true; }
        k = ws.site + 1;
    }
    for (; k < i0; ++k)
        if (g.ins_len[k] > 0) { ws.site = k; ws.t = 0; return true; }
    return false;
}

// Follows st.choice[0 .. prefix) at the deletions met and does not jump at later ones.  A walk whose
// last base lies at or beyond `limit` (the region's end) does not exist, like one that runs off the
// chromosome: vg reports a walk only if both of its ends map into the region.  `prod` returns
// the number of allele combinations of the layout.  When the visitor wants bases, walk `q` of the
// layout is decoded on the way: rem starts as the layout's product and is divided at every SNP.
// Where simulate() reads site records: straight from the array, or from the copy of the window's first kSiteCache
// records an emit thread made in LDS (one batch of loads) -- a walk is a chain of "what is at x" questions, and as
// global loads each answer was an L2 round trip.
struct GlobalSites {
    const SiteRec *rec;
    __device__ SiteRec at(int k) const { return rec[k]; }
};
struct CachedSites {
    const SiteRec *rec;
    const SiteRec *lds;      // records of sites first .. first + kSiteCache - 1, record d at lds[d * stride]
    int first, stride;       // stride = threads of the workgroup: record d of all threads lies side by side, so that the
                             // 16-byte reads of a wave fall on all banks (thread-major -- 128 bytes per thread -- every
                             // read was a 32-way bank conflict: SQ_LDS_BANK_CONFLICT 89 % of the kernel's LDS cycles)
    __device__ SiteRec at(int k) const
    {
        const unsigned d = (unsigned)(k - first);
        return d < (unsigned)kSiteCache ? lds[d * stride] : rec[k];
    }
};

template <class V, class S, long long MaxWalks = kMaxWalksPerWindow>
__device__ inline int simulate(const GraphDev &g, const S &sites, long long p, int W, int i0, const WalkStart &ws, int prefix,
                               WalkState &st, V &vis, long long q, long long rem, long long &prod, long long limit)
{
    long long x = p;
    int n = 0, d = 0, i = i0;
    long long next_pos = sites.at(i).pos;               // position of site i (kNoSitePos behind the last), kept in a register
    prod = 1;
    if (ws.site >= 0) {                                // the walk starts on base ws.t of an insertion behind p - 1
        int take = g.ins_len[ws.site] - ws.t;
        if (take > W) take = W;
        if constexpr (V::kWantsBases)
            for (int j = 0; j < take; ++j) vis.ins_base(j, ws.site, ws.t + j);
        vis.took(ws.site);
        n = take;
        if (n == W) { st.nd = 0; st.last = p - 1; return p <= limit ? WALK_OK : WALK_DEAD; }
    }
    for (;;) {
        if (x >= g.ref_len) { st.nd = d; return WALK_DEAD; }
        while (next_pos < x) {                         // only after a jump over deleted bases
            ++i;
            next_pos = sites.at(i).pos;
        }
        if (next_pos > x) {                            // no site here
            if constexpr (!V::kWantsBases) {
                // nobody looks at the bases: take the whole site-free stretch at once
                long long run = next_pos - x;
                if (run > W - n) run = W - n;
                if (x + run > g.ref_len) { st.nd = d; return WALK_DEAD; }
                n += (int)run;
                x += run;
                if (n == W) { st.nd = d; st.last = x - 1; return x <= limit ? WALK_OK : WALK_DEAD; }
                continue;
            } else {
                vis.base(n, x, -1, 0, 1);
                if (++n == W) { st.nd = d; st.last = x; return x + 1 <= limit ? WALK_OK : WALK_DEAD; }
                ++x;
                continue;
            }
        }
        int snp = -1, del0 = -1, del1 = -1, ins0 = -1, ins1 = -1;   // insertions / deletions anchored here: sites [a, b)
        int snp_alts = 0;
        for (int k = i; k < g.n_sites; ++k) {
            const SiteRec r = sites.at(k);
            if (r.pos != x) break;
            if (r.del_len > 0) { if (del0 < 0) del0 = k; del1 = k + 1; }
            else if (r.ins_len > 0) { if (ins0 < 0) ins0 = k; ins1 = k + 1; }
            else { snp = k; snp_alts = r.n_alts; }
        }
        int a = 0, nall = 1;
        if (snp >= 0) {
            nall = 1 + snp_alts;
            prod *= nall;
            if (prod > MaxWalks) { st.nd = d; return WALK_OVERFLOW; }
            if constexpr (V::kWantsBases) {
                rem /= nall;
                a = (int)((q / rem) % nall);
            }
        }
        vis.base(n, x, snp, a, nall);
        if (++n == W) { st.nd = d; st.last = x; return x + 1 <= limit ? WALK_OK : WALK_DEAD; }
        bool read_ins = false;
        for (int k = ins0; k >= 0 && k < ins1; ++k) {   // read insertion k?  (0 = no, 1 = yes; a yes ends the site)
            if (d >= kMaxDecisions) { st.nd = d; return WALK_OVERFLOW; }
            const int c = d < prefix ? (int)((st.choice >> d) & 1u) : 0;
            st.choice = (st.choice & ~(1u << d)) | ((unsigned)c << d);
            ++d;
            if (!c) { vis.passed(k); continue; }
            int take = sites.at(k).ins_len;
            if (take > W - n) take = W - n;
            if constexpr (V::kWantsBases)
                for (int j = 0; j < take; ++j) vis.ins_base(n + j, k, j);
            vis.took(k);
            n += take;
            if (n == W) { st.nd = d; st.last = x; return x + 1 <= limit ? WALK_OK : WALK_DEAD; }
            read_ins = true;
            break;
        }
        // the deletions anchored here (site order): jump this one?  (0 before 1; a yes ends the site).  A walk that
        // goes on behind x uses the bases of every deletion it did not jump as far as that one reaches beyond its
        // landing place: their carriers lack those bases.
        int jumped = -1;
        if (!read_ins)
            for (int k = del0; k >= 0 && k < del1; ++k) {
                if (d >= kMaxDecisions) { st.nd = d; return WALK_OVERFLOW; }
                const int c = d < prefix ? (int)((st.choice >> d) & 1u) : 0;
                st.choice = (st.choice & ~(1u << d)) | ((unsigned)c << d);
                ++d;
                if (c) { jumped = k; break; }
            }
        if (jumped < 0) {
            for (int k = del0; k >= 0 && k < del1; ++k) vis.passed(k);     // x + 1 lies inside every one of them
            ++x;
        } else {
            vis.took(jumped);
            const int len = sites.at(jumped).del_len;
            for (int k = del0; k < del1; ++k)
                if (k != jumped && sites.at(k).del_len > len) vis.passed(k);
            const long long land = x + len + 1;
            // sites inside the jumped span are never visited; a deletion anchored there that reaches beyond the
            // landing place has bases the walk uses
            for (int k = del1; k < g.n_sites; ++k) {
                const SiteRec r = sites.at(k);
                if (r.pos >= land) break;
                if (r.del_len > 0 && (long long)r.pos + r.del_len >= land) vis.passed(k);
            }
            x = land;
        }
    }
}

// next layout: the last jump decision that is still 0 becomes 1; returns the new prefix length or -1
__device__ inline int next_walk(WalkState &st)
{
    int t = st.nd - 1;
    while (t >= 0 && ((st.choice >> t) & 1u)) --t;
    if (t < 0) return -1;
    st.choice |= 1u << t;
    return t + 1;
}

// Does a deletion anchored before p remove the base at p (the window starts on deleted bases)?  i0 = first site at
// or after p.  Deletions may overlap, so this asks how far ANY of the earlier ones reaches.
__device__ inline bool covered_by_deletion(const GraphDev &g, long long p, int i0)
{
    return g.max_reach[i0] >= p;
}
// f(site) for every deletion anchored before p that removes the base at p: back along the chain of deletions for as
// long as one of those further back can still reach p
template <class F>
__device__ inline void for_covering_deletions(const GraphDev &g, long long p, int i0, F f)
{
    for (int d = g.prev_del[i0]; d >= 0 && g.max_reach[d + 1] >= p; d = g.prev_del[d])
        if ((long long)g.pos[d] + g.del_len[d] >= p) f(d);
}

__device__ inline int lower_bound_pos(const int *pos, int n, long long v)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (pos[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// window w of the plan -> (region, start p); region_off[r] = first window of region r
__device__ inline int region_of(const long long *region_off, int n_regions, long long w)
{
    int lo = 0, hi = n_regions;          // last r with region_off[r] <= w
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (region_off[mid] <= w) lo = mid; else hi = mid;
    }
    return lo;
}

// What graph_count_del_kernel learns about the first layouts of a listed window, kept for graph_emit_del_kernel: a
// walk finds its layout by comparing its rank with <= kLayoutCache cumulative counts and replays that layout once,
// instead of running the odometer from the start (2.5 simulations per walk of a one-deletion window became 1).
constexpr int kLayoutCache = 8;
constexpr int kDelMeta = 5;          // staging words per deletion walk: row, start, stop, count, flag
struct LayoutRec {
    int cum_end;         // walks of this window up to and including this layout
    unsigned choice;     // bits 0..23: the jump / insertion decisions, bits 24..31: how many were taken
    int site, t;         // WalkStart of the layout
};
// a listed window as its walks' threads need it (one 32-byte record instead of a chain of lookups by window index)
struct DelRec {
    long long p, limit;  // window start, end of its region
    int w, i0;           // window index, first site at or behind p
    int pad[2];
};
constexpr int kDelThreads = 128;     // walks (threads) per workgroup of graph_emit_del_kernel

// one thread per window: first site inside it and the number of walks (product of allele counts)
__global__ void __launch_bounds__(kCountThreads)
graph_count_kernel(GraphDev g, int n_regions, const long long *__restrict__ region_off,
                   const long long *__restrict__ first_start, const long long *__restrict__ region_stop, int W,
                   long long n_windows, int *__restrict__ first_site, long long *__restrict__ n_walks,
                   int *__restrict__ win_region, long long *__restrict__ win_start, int *__restrict__ overflow,
                   int *__restrict__ del_list, int *__restrict__ del_count, int *__restrict__ win_sites,
                   unsigned long long *__restrict__ slow_totals)
{
    const long long w = (long long)blockIdx.x * kCountThreads + threadIdx.x;
    if (w >= n_windows) return;
    const int r = region_of(region_off, n_regions, w);
    const long long p = first_start[r] + (w - region_off[r]);
    const int i0 = lower_bound_pos(g.pos, g.n_sites, p);
    long long walks = 1;
    bool touches_del = g.n_dels > 0 && covered_by_deletion(g, p, i0);
    bool over = false;      // too many walks for a plain window -- reported only if the window is kept as one
    int ns = 0;
    for (int i = i0; i < g.n_sites && g.pos[i] < p + W; ++i, ++ns) {
        if (g.del_len[i] || g.ins_len[i]) touches_del = true;
        if (!over) {
            walks *= 1 + g.n_alts[i];
            over = walks > kMaxWalksPerWindow;
        }
    }
    if (g.n_ins > 0) {
        for (int k = i0 - 1; k >= 0 && g.pos[k] == p - 1; --k)     // walks that start inside an insertion behind p - 1
            if (g.ins_len[k] > 0) touches_del = true;
        // with insertions in the graph the windows run on to the region's last base: beyond E - W only a walk
        // that reads inserted bases ends inside the region -- a plain window there holds no walk, however many
        // SNPs it spans (no overflow to report for it)
        if (!touches_del && p + W > region_stop[r]) { walks = 0; over = false; }
    }
    if (touches_del) {
        // the sites a walk meets depend on the deletions it takes: graph_count_del_kernel enumerates them (and
        // reports its own overflow).
        // (Done here, the few deletion windows of a wave made all of its lanes wait for their odometers.)
        walks = 0;
        del_list[atomicAdd(del_count, 1)] = (int)w;
    } else if (over) {
        walks = 0;
        atomicMax(overflow, 1);
    } else if (ns > 3 && walks > 0 && g.alt_bits) {
        // its walks' haplotype counts need the bitsets: room for their count jobs (exact; see CountJobs)
        atomicAdd(&slow_totals[0], (unsigned long long)walks);
        atomicAdd(&slow_totals[1], (unsigned long long)walks * (unsigned long long)ns);
    }
    first_site[w] = i0;
    n_walks[w] = walks;
    win_region[w] = r;      // the emit kernel reads these instead of repeating the search: its waves
    win_start[w] = p;       // are latency-bound, and a binary search is a chain of dependent loads
    win_sites[w] = touches_del ? 0 : ns;   // the sites a plain window's walks choose alleles at (a listed window's rows are
                                           // placeholders in the plain kernel: no work for them there)
}

// A <-> T, C <-> G in either case, everything else (N) stays.  Without branches: the emit kernels call it per base,
// and as a switch it was four to eight divergent branches each time.  Letters differ in their low five bits
// (A 1, C 3, G 7, T 20); A ^ T = 0x15, C ^ G = 0x04.
__device__ __forceinline__ uint8_t complement(uint8_t c)
{
    const unsigned t = c & 31u;
    const unsigned letter = ((c & 0xC0u) == 0x40u) ? 1u : 0u;
    const unsigned at = (((1u << 1) | (1u << 20)) >> t) & letter;
    const unsigned cg = (((1u << 3) | (1u << 7)) >> t) & letter;
    return (uint8_t)(c ^ (at * 0x15u) ^ (cg * 0x04u));
}

// unaligned loads (gfx950 under HSA handles them in hardware; the arrays read this way are padded behind their end)
__device__ __forceinline__ unsigned long long load_u64(const void *p)
{
    unsigned long long v;
    __builtin_memcpy(&v, p, sizeof v);
    return v;
}
constexpr size_t kReadPad = 16;     // bytes behind d_ref, d_n_alts, d_pos that such a load may touch

// thread per listed deletion window: number of walks by running the odometer
__global__ void __launch_bounds__(kCountThreads)
graph_count_del_kernel(GraphDev g, const int *__restrict__ del_list, const int *__restrict__ del_count,
                       const long long *__restrict__ win_start, const int *__restrict__ win_region,
                       const long long *__restrict__ region_stop, int W, const int *__restrict__ first_site,
                       long long *__restrict__ n_walks, long long *__restrict__ del_walks, int *__restrict__ overflow,
                       LayoutRec *__restrict__ layouts /* [listed][kLayoutCache] */, int *__restrict__ n_layouts,
                       DelRec *__restrict__ del_rec)
{
    const int m = blockIdx.x * kCountThreads + threadIdx.x;
    if (m >= *del_count) return;
    const int w = del_list[m];
    const long long p = win_start[w];
    const int i0 = first_site[w];
    del_rec[m] = DelRec{p, region_stop[win_region[w]], w, i0, {0, 0}};
    WalkState st;
    NoVisitor nv;
    WalkStart ws;
    long long walks = 0;
    bool bad = false;
    int nl = 0;
    do {                                             // per start: one pass per layout
        int prefix = 0;
        do {
            long long prod = 0;
            const int rc = simulate(g, GlobalSites{g.site_rec}, p, W, i0, ws, prefix, st, nv, 0, 0, prod, region_stop[win_region[w]]);
            if (rc == WALK_OK) {
                walks += prod;
                if (nl < kLayoutCache && walks <= kMaxWalksPerWindow) {
                    const unsigned bits = ((unsigned)st.nd << 24) | (st.choice & ((1u << st.nd) - 1u));
                    layouts[(size_t)m * kLayoutCache + nl] = LayoutRec{(int)walks, bits, ws.site, ws.t};
                }
                ++nl;
            }
            if (rc == WALK_OVERFLOW || walks > kMaxWalksPerWindow) { bad = true; break; }
            prefix = next_walk(st);
        } while (prefix >= 0);
    } while (!bad && next_start(g, p, i0, ws));
    if (bad) { walks = 0; nl = 0; atomicMax(overflow, 1); }
    n_walks[w] = walks;
    del_walks[m] = walks;
    n_layouts[m] = nl;
}

// thread per listed deletion window: walk t of the deletion walks belongs to list entry del_entry[t]
__global__ void __launch_bounds__(kCountThreads)
graph_del_map_kernel(const int *__restrict__ del_count, const long long *__restrict__ del_base,
                     int *__restrict__ del_entry)
{
    const int m = blockIdx.x * kCountThreads + threadIdx.x;
    if (m >= *del_count) return;
    for (long long t = del_base[m]; t < del_base[m + 1]; ++t) del_entry[t] = m;
}

// haplotypes per allele of every site (allele 0 = reference): a walk through ONE site needs no bitset
// read at all, and most windows hold at most one site
__global__ void __launch_bounds__(256)
graph_allele_count_kernel(GraphDev g, int *__restrict__ allele_count /* [n_sites][4] */)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= g.n_sites) return;
    int c[4] = {0, 0, 0, 0};
    for (int word = 0; word < g.hw; ++word) {
        const unsigned long long *b = g.alt_bits + ((size_t)i * kMaxAlts) * g.hw + word;
        unsigned long long any = 0ull;
        for (int a = 0; a < g.n_alts[i]; ++a) {
            const unsigned long long v = b[(size_t)a * g.hw];
            c[a + 1] += __popcll(v);
            any |= v;
        }
        unsigned long long valid = ~0ull;
        if (word == g.hw - 1 && (g.n_hap & 63)) valid = (1ull << (g.n_hap & 63)) - 1ull;
        c[0] += __popcll(~any & valid);
    }
    for (int a = 0; a < 4; ++a) allele_count[(size_t)i * 4 + a] = c[a];
}

// word `word` of the haplotype set "allele a at site s" (a = 0: none of its alternates)
__device__ inline unsigned long long allele_word(const GraphDev &g, int s, int a, int word)
{
    const unsigned long long *b = g.alt_bits + ((size_t)s * kMaxAlts) * g.hw + word;
    const int na = g.n_alts[s];
    if (a > 0) return a <= na ? b[(size_t)(a - 1) * g.hw] : 0ull;
    unsigned long long any = b[0];
    if (na > 1) any |= b[(size_t)g.hw];
    if (na > 2) any |= b[(size_t)2 * g.hw];
    unsigned long long valid = ~0ull;
    if (word == g.hw - 1 && (g.n_hap & 63)) valid = (1ull << (g.n_hap & 63)) - 1ull;
    return ~any & valid;
}

// thread per (site i, allele pair): haplotypes with allele a at site i and b at site i + 1
__global__ void __launch_bounds__(256)
graph_pair_count_kernel(GraphDev g, int *__restrict__ pair_count /* [n_sites][16] */)
{
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    const int i = (int)(id >> 4), a = (int)((id >> 2) & 3), b = (int)(id & 3);
    if (i >= g.n_sites) return;
    int c = 0;
    if (i + 1 < g.n_sites && a <= g.n_alts[i] && b <= g.n_alts[i + 1])
        for (int word = 0; word < g.hw; ++word)
            c += __popcll(allele_word(g, i, a, word) & allele_word(g, i + 1, b, word));
    pair_count[id] = c;
}

// thread per (site i, allele triple): ... and c at site i + 2
__global__ void __launch_bounds__(256)
graph_triple_count_kernel(GraphDev g, int *__restrict__ triple_count /* [n_sites][64] */)
{
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    const int i = (int)(id >> 6), a = (int)((id >> 4) & 3), b = (int)((id >> 2) & 3), c3 = (int)(id & 3);
    if (i >= g.n_sites) return;
    int c = 0;
    if (i + 2 < g.n_sites && a <= g.n_alts[i] && b <= g.n_alts[i + 1] && c3 <= g.n_alts[i + 2])
        for (int word = 0; word < g.hw; ++word)
            c += __popcll(allele_word(g, i, a, word) & allele_word(g, i + 1, b, word) & allele_word(g, i + 2, c3, word));
    triple_count[id] = c;
}

// thread per window: walk t of the plan belongs to window walk_window[t]
__global__ void __launch_bounds__(kCountThreads)
graph_map_kernel(long long n_windows, const long long *__restrict__ walk_base, int *__restrict__ walk_window)
{
    const long long w = (long long)blockIdx.x * kCountThreads + threadIdx.x;
    if (w >= n_windows) return;
    const long long b = walk_base[w], e = walk_base[w + 1];
    for (long long t = b; t < e; ++t) walk_window[t] = (int)w;
}

// Haplotypes that carry allele a_k at site s_k for every k < n (allele 0 = none of the alternates; a
// deletion is a site with one alternate: 1 = carries it).  No constraint: all; one: the popcount table;
// two or three neighbouring sites: the pair / triple tables; else the AND of the bitsets.
// `done` = false: the walk needs the bitsets (count_by_bitsets, or a deferred job: CountJobs).
template <class F>
__device__ __forceinline__ long long count_by_tables(const GraphDev &g, const int *__restrict__ allele_count, int n, F at,
                                                     bool &done)
{
    done = true;
    if (!g.alt_bits) return 0;
    if (n == 0) return g.n_hap;
    if (n == 1) {
        int site, a;
        at(0, site, a);
        return allele_count[(size_t)site * 4 + a];
    }
    if (n == 2) {               // two neighbouring sites: the pair table
        int s0, a0, s1, a1;
        at(0, s0, a0);
        at(1, s1, a1);
        if (s1 == s0 + 1) return g.pair_count[(size_t)s0 * 16 + a0 * 4 + a1];
        if (s0 == s1 + 1) return g.pair_count[(size_t)s1 * 16 + a1 * 4 + a0];
    } else if (n == 3) {        // three consecutive sites (in any order): the triple table
        int sv[3], av[3];
        at(0, sv[0], av[0]);
        at(1, sv[1], av[1]);
        at(2, sv[2], av[2]);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 2 - r; ++k)
                if (sv[k] > sv[k + 1]) {
                    const int ts = sv[k], ta = av[k];
                    sv[k] = sv[k + 1]; av[k] = av[k + 1];
                    sv[k + 1] = ts; av[k + 1] = ta;
                }
        if (sv[1] == sv[0] + 1 && sv[2] == sv[0] + 2)
            return g.triple_count[(size_t)sv[0] * 64 + av[0] * 16 + av[1] * 4 + av[2]];
    }
    done = false;
    return 0;
}

// the AND of the bitsets, sixteen words per trip so that their loads are in flight together (one word per trip made
// every such walk a chain of ~hw dependent L2 latencies)
template <class F>
__device__ inline long long count_by_bitsets(const GraphDev &g, int n, F at)
{
    constexpr int kWordsPerTrip = 16;
    long long count = 0;
    for (int w0 = 0; w0 < g.hw; w0 += kWordsPerTrip) {
        unsigned long long acc[kWordsPerTrip];
#pragma unroll
        for (int j = 0; j < kWordsPerTrip; ++j) {
            const int word = w0 + j;
            acc[j] = word < g.hw ? ~0ull : 0ull;
            if (word == g.hw - 1 && (g.n_hap & 63)) acc[j] = (1ull << (g.n_hap & 63)) - 1ull;
        }
        for (int k = 0; k < n; ++k) {
            int site, a;
            at(k, site, a);
            const unsigned long long *b = g.alt_bits + ((size_t)site * kMaxAlts) * g.hw + w0;
            const int na = g.n_alts[site];
#pragma unroll
            for (int j = 0; j < kWordsPerTrip; ++j) {
                if (w0 + j >= g.hw) continue;
                unsigned long long bits;
                if (a > 0) {
                    bits = b[(size_t)(a - 1) * g.hw + j];
                } else {
                    bits = b[j];
                    if (na > 1) bits |= b[(size_t)g.hw + j];
                    if (na > 2) bits |= b[(size_t)2 * g.hw + j];
                    bits = ~bits;
                }
                acc[j] &= bits;
            }
        }
#pragma unroll
        for (int j = 0; j < kWordsPerTrip; ++j) count += __popcll(acc[j]);
    }
    return count;
}

// Walks whose count needs the bitsets are rare among the plain walks (four or more sites in one window: 0.5 % of the
// walks of the bench's graph) and a third of the deletion walks -- but one such lane kept its whole wave in
// count_by_bitsets (a quarter of the plain kernel's waves held one: 169 us with them, 99 us without; the deletion kernel
// 173 against 118 us).  The emit kernels therefore only NOTE such a walk -- a job: where its count goes and its
// constraints, each as site * 16 + (alternates of the site) * 4 + allele -- and graph_count_jobs_kernel, sixteen lanes per
// job, counts them afterwards.  The plain walks' share of the lists is exact (graph_count_kernel adds it up per plan), so
// the plain kernel holds no bitset code at all; a deletion walk that finds its share full is counted in place as before.
struct JobHead {
    long long dst;            // t >= 0: rows 2 t, 2 t + 1 of the freq column;  ~td < 0: the staged deletion walk td
    int off, n;               // its constraints: pool[off .. off + n);  n <= 0: no job (its walk was counted in place)
    int first[4];             // the first four of them once more: most jobs have no others, and read here they cost the
                              // job kernel one dependent load less
};
struct CountJobs {
    int *counters;            // [0] plain jobs, [1] plain pool entries, [2] deletion jobs, [3] deletion pool entries
    JobHead *head;            // [plain_jobs + del_jobs]: the plain kernel's jobs first
    int *pool;                // [plain_pool + del_pool]
    int plain_jobs, plain_pool, del_jobs, del_pool;     // capacities
};
__device__ __forceinline__ int job_constraint(int site, int n_alts, int allele) { return site * 16 + n_alts * 4 + allele; }

constexpr int kJobLanes = 16;
constexpr int kJobWords = 5;     // bitset words per lane and pass: 80 words (5 120 haplotypes) in one
__global__ void __launch_bounds__(256)
graph_count_jobs_kernel(GraphDev g, CountJobs jobs, int which /* 0: the plain kernel's jobs, 1: the deletion kernel's */,
                        long long *__restrict__ freq, long long *__restrict__ stg_meta, int meta_pitch)
{
    const int lane = threadIdx.x & (kJobLanes - 1);
    const int group = (int)((blockIdx.x * blockDim.x + threadIdx.x) / kJobLanes);
    const int n_groups = (int)((gridDim.x * blockDim.x) / kJobLanes);
    const int n_jobs = which == 0 ? min(jobs.counters[0], jobs.plain_jobs) : min(jobs.counters[2], jobs.del_jobs);
    const JobHead *heads = jobs.head + (which == 0 ? 0 : jobs.plain_jobs);
    for (int jj = group; jj < n_jobs; jj += n_groups) {
        const JobHead h = heads[jj];
        if (h.n <= 0) continue;
        const int *cons = jobs.pool + h.off;
        long long count = 0;
        for (int w0 = 0; w0 < g.hw; w0 += kJobLanes * kJobWords) {
            unsigned long long acc[kJobWords];
#pragma unroll
            for (int i = 0; i < kJobWords; ++i) {
                const int word = w0 + i * kJobLanes + lane;
                acc[i] = word < g.hw ? ~0ull : 0ull;
                if (word == g.hw - 1 && (g.n_hap & 63)) acc[i] = (1ull << (g.n_hap & 63)) - 1ull;
            }
            for (int k0 = 0; k0 < h.n; k0 += 4) {
                int c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) c[u] = k0 + u < h.n ? (k0 == 0 ? h.first[u] : cons[k0 + u]) : -1;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (c[u] < 0) continue;
                    const int a = c[u] & 3, na = (c[u] >> 2) & 3;
                    const unsigned long long *b = g.alt_bits + ((size_t)(c[u] >> 4) * kMaxAlts) * g.hw + w0 + lane;
#pragma unroll
                    for (int i = 0; i < kJobWords; ++i) {
                        if (w0 + i * kJobLanes + lane >= g.hw) continue;
                        unsigned long long bits;
                        if (a > 0) {
                            bits = b[(size_t)(a - 1) * g.hw + i * kJobLanes];
                        } else {
                            bits = b[i * kJobLanes];
                            if (na > 1) bits |= b[(size_t)g.hw + i * kJobLanes];
                            if (na > 2) bits |= b[(size_t)2 * g.hw + i * kJobLanes];
                            bits = ~bits;
                        }
                        acc[i] &= bits;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < kJobWords; ++i) count += __popcll(acc[i]);
        }
#pragma unroll
        for (int off = kJobLanes / 2; off > 0; off >>= 1) count += __shfl_xor(count, off, kJobLanes);
        if (lane == 0) {
            if (h.dst >= 0) { freq[2 * h.dst] = count; freq[2 * h.dst + 1] = count; }
            else stg_meta[(size_t)(~h.dst) * meta_pitch + 3] = count;
        }
    }
}

// Thread per walk.  Neighbouring threads are walks of one window or of overlapping windows: their
// reference bytes, site records and bitset words are the same cache lines, and every per-row column
// is written fully coalesced (rows 2t and 2t+1 of thread t).  The walk index is a mixed-radix number
// over the sites of the window, last site fastest (itertools.product order); the allele digits are
// kept packed two bits each.
__device__ __forceinline__ void
emit_plain_body(const unsigned bid, const GraphDev &g, const int *__restrict__ allele_count, const int *__restrict__ walk_window,
                  const int *__restrict__ win_region, const long long *__restrict__ win_start, int W,
                  long long n_walks, const int *__restrict__ first_site, const long long *__restrict__ walk_base,
                  const int *__restrict__ win_sites,
                  uint8_t *__restrict__ kmers, long long *__restrict__ start, long long *__restrict__ stop,
                  uint8_t *__restrict__ strand, long long *__restrict__ freq, uint8_t *__restrict__ is_ref,
                  int *__restrict__ region, int *__restrict__ walk, const bool wide_rows, const CountJobs &jobs)
{
    // the block's 2 x 256 k-mer rows are contiguous in the output: they are assembled in LDS with
    // byte writes and leave with 16-byte coalesced stores (row starts are not even dword aligned)
    extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
    const long long t0 = (long long)bid * kEmitThreads;
    const long long t = t0 + threadIdx.x;
    const bool valid = t < n_walks;
    const long long tt = valid ? t : n_walks - 1;       // idle threads of the last block shadow a real walk
    const int w = walk_window[tt];
    const long long p = win_start[w];
    const int i0 = first_site[w];
    const int ns = win_sites[w];
    const int r = win_region[w];
    int q = (int)(tt - walk_base[w]);
    const int q0 = q;
    uint8_t *fwd = stage + (size_t)(2 * threadIdx.x) * W, *rev = fwd + W;
    // (windows that touch a deletion are written here as if they did not -- harmless bytes -- and
    // rewritten by graph_del_scatter_kernel, which runs next on the same stream)
    long long count = 0;
    const long long end_pos = p + W;
    bool any_alt = false;
    {
    // Everything the walk reads is requested before anything is used: the reference window as up to eight unaligned
    // 8-byte loads, the allele counts of its first eight sites as one, the positions of its first four as two.  (As
    // loops of byte loads -- W of them for the bases, one per site for the rest, each waited for -- the kernel was a
    // chain of ~25 dependent L2 latencies per wave: 180 us for 3 million walks.)
    unsigned long long rw[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) rw[c] = 8 * c < W ? load_u64(g.ref + p + 8 * c) : 0ull;
    const unsigned long long nal8 = ns > 0 ? load_u64(g.n_alts + i0) : 0ull;
    unsigned long long pos01 = 0ull, pos23 = 0ull;
    if (ns > 0) { pos01 = load_u64(g.pos + i0); pos23 = load_u64(g.pos + i0 + 2); }
    auto n_alts_of = [&](int k) { return k < 8 ? (int)((nal8 >> (8 * k)) & 0xffull) : (int)g.n_alts[i0 + k]; };
    auto pos_of = [&](int k) {
        return k < 2 ? (int)(pos01 >> (32 * k)) : (k < 4 ? (int)(pos23 >> (32 * (k - 2))) : g.pos[i0 + k]);
    };
    // allele digits, last site first
    unsigned long long dig[2] = {0ull, 0ull};     // 2 bits per site, up to 64 sites
    unsigned long long qd = (unsigned long long)(unsigned)q;
    for (int k = ns - 1; k >= 0; --k) {
        const int nall = 1 + n_alts_of(k);
        const unsigned long long a = (unsigned long long)take_digit(qd, nall);
        dig[k >> 5] |= a << (2 * (k & 31));
    }
    auto allele = [&](int k) { return (int)((dig[k >> 5] >> (2 * (k & 31))) & 3ull); };
    any_alt = (dig[0] | dig[1]) != 0ull;

    // rows 2t (forward) and 2t+1 (reverse complement): reference window first, then the alternates
#pragma unroll
    for (int j = 0; j < GFM_MAX_WIDTH; ++j) {
        if (j >= W) break;
        const uint8_t c = (uint8_t)(rw[j >> 3] >> (8 * (j & 7)));
        fwd[j] = c;
        rev[W - 1 - j] = complement(c);
    }
    for (int k = 0; k < ns; ++k) {
        const int a = allele(k);
        if (a) {
            const uint8_t c = g.alt_bases[(size_t)(i0 + k) * kMaxAlts + (a - 1)];
            const int j = (int)(pos_of(k) - p);
            fwd[j] = c;                      // same thread, same address: ordered after the reference byte
            rev[W - 1 - j] = complement(c);
        }
    }

    // haplotypes that carry every allele of the walk
    {
        auto at = [&](int k, int &site, int &a) { site = i0 + k; a = allele(k); };
        bool done;
        count = count_by_tables(g, allele_count, ns, at, done);
        if (!done && valid) {    // four or more sites: a job for graph_count_jobs_kernel (room for all of them: exact)
            const int j = atomicAdd(&jobs.counters[0], 1);
            const int off = atomicAdd(&jobs.counters[1], ns);
            if (j < jobs.plain_jobs && off + ns <= jobs.plain_pool) {
                JobHead h{t, off, ns, {0, 0, 0, 0}};
                for (int k = 0; k < ns; ++k) {
                    const int v = job_constraint(i0 + k, n_alts_of(k), allele(k));
                    jobs.pool[off + k] = v;
                    if (k < 4) h.first[k] = v;
                }
                jobs.head[j] = h;
            }
        }
    }
    }   // plain window
    __syncthreads();
    {
        const long long rows_here = 2 * ((n_walks - t0) < (long long)kEmitThreads ? (n_walks - t0) : (long long)kEmitThreads);
        const long long nbytes = rows_here * W;
        uint8_t *dst = kmers + (size_t)(2 * t0) * W;     // 512 * W * blockIdx: a multiple of 16
        for (long long o = (long long)threadIdx.x * 16; o < nbytes; o += (long long)kEmitThreads * 16) {
            if (o + 16 <= nbytes) {
                *reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(stage + o);
            } else {
                for (long long b = o; b < nbytes; ++b) dst[b] = stage[b];
            }
        }
    }
    if (!valid) return;
    const long long row = 2 * t;
    const uint8_t flag = any_alt ? 0 : 1;
    if (wide_rows) {   // the two rows of the walk as ONE store per column (gfm_graph_emit checked the alignment)
        *reinterpret_cast<longlong2 *>(start + row) = longlong2{p, end_pos};
        *reinterpret_cast<longlong2 *>(stop + row) = longlong2{end_pos, p};
        *reinterpret_cast<longlong2 *>(freq + row) = longlong2{count, count};
        *reinterpret_cast<int2 *>(region + row) = int2{r, r};
        *reinterpret_cast<int2 *>(walk + row) = int2{q0, q0};
        *reinterpret_cast<unsigned short *>(strand + row) = (unsigned short)('+' | ('-' << 8));
        *reinterpret_cast<unsigned short *>(is_ref + row) = (unsigned short)(flag | (flag << 8));
    } else {
        start[row] = p;          start[row + 1] = end_pos;
        stop[row] = end_pos;     stop[row + 1] = p;
        strand[row] = '+';       strand[row + 1] = '-';
        freq[row] = count;       freq[row + 1] = count;
        is_ref[row] = is_ref[row + 1] = flag;
        region[row] = region[row + 1] = r;
        walk[row] = walk[row + 1] = q0;
    }
}

// visitor of simulate() that writes the bases of a walk and collects what its haplotypes must carry
struct DelEmit {
    // the graph's arrays it reads, BY VALUE: a reference to the GraphDev kernel argument, held by an object that lives in
    // scratch memory, made every thread copy the whole argument struct (144 bytes) into scratch and read its fields
    // back from there
    const uint8_t *alt_bases, *ins_bases;
    const int *ins_off;
    uint8_t *fwd, *rev;
    int *src;                 // [W] reference position of base j, fetched after the walk (-1: written already)
    int W;
    // constraints as job_constraint() packs them (allele: SNP 0..3; deletion / insertion 1 = taken, 0 = passed by).  The
    // first four in registers -- the tables need no more --, the rest in scratch memory, which nothing reads before the
    // walk is over (with all of them there, every load inside the walk waited for the scratch stores before it).
    int n_cons = 0;
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    int *more;                                  // [kMaxConstraints - 4], an array of the caller's: inside this object it
                                                // kept ALL of the object in scratch memory, the four "registers" too
    bool alt = false;
    static constexpr bool kWantsBases = true;
    __device__ DelEmit(const GraphDev &g, uint8_t *fwd_, uint8_t *rev_, int *src_, int W_, int *more_)
        : alt_bases(g.alt_bases), ins_bases(g.ins_bases), ins_off(g.ins_off), fwd(fwd_), rev(rev_), src(src_), W(W_), more(more_) {}
    __device__ void add(int site, int code, int n_alts = 1)
    {
        const int v = job_constraint(site, n_alts, code);
        if (n_cons == 0) c0 = v;
        else if (n_cons == 1) c1 = v;
        else if (n_cons == 2) c2 = v;
        else if (n_cons == 3) c3 = v;
        else if (n_cons < kMaxConstraints) more[n_cons - 4] = v;
        if (n_cons < kMaxConstraints) ++n_cons;
    }
    __device__ int get(int k) const { return k == 0 ? c0 : (k == 1 ? c1 : (k == 2 ? c2 : (k == 3 ? c3 : more[k - 4]))); }
    // A reference base is only NOTED here and fetched once the walk is known (emit_del_body: eight loads in flight at a
    // time); fetched here, every base was an L2 round trip in the middle of the walk.
    __device__ void base(int j, long long x, int snp, int a, int nall)
    {
        src[j] = (int)x;
        if (snp >= 0) {
            if (a) {
                const uint8_t c = alt_bases[(size_t)snp * kMaxAlts + (a - 1)];
                fwd[j] = c;
                rev[W - 1 - j] = complement(c);
                src[j] = -1;
                alt = true;
            }
            add(snp, a, nall - 1);
        }
    }
    __device__ void ins_base(int j, int site, int t)
    {
        const uint8_t c = ins_bases[ins_off[site] + t];
        fwd[j] = c;
        rev[W - 1 - j] = complement(c);
        src[j] = -1;
        alt = true;
    }
    __device__ void took(int site) { add(site, 1); }
    __device__ void passed(int site) { add(site, 0); }
};


// Windows that touch a deletion or an insertion: one thread per walk of those windows (compacted: a kernel over all
// walks spent its time in waves with one or two such lanes).  The thread finds its layout (the ones
// graph_count_del_kernel kept, else the window's odometer up to its own rank), replays that walk assembling its two
// rows in LDS, and counts the haplotypes.  These are few, long, latency-bound threads (188 000 walks of 3 million in the
// bench's graph: less than one resident set of workgroups), so their kernel runs on a side stream BESIDE the kernel of
// the plain walks and writes to a staging area; graph_del_scatter_kernel then puts the rows in place (the plain kernel
// has written placeholder bytes there).  Run one after the other on one stream the two kernels took 180 us each; as
// two bodies of ONE kernel (first workgroups: deletion walks) 436 us -- the plain body then runs with the registers
// and scratch of the deletion body.
__device__ __forceinline__ void
emit_del_body(const unsigned bid, const GraphDev &g, const int *__restrict__ allele_count,
              const int *__restrict__ del_entry, const long long *__restrict__ del_base,
              const DelRec *__restrict__ del_rec, int W, long long n_del_walks,
              const long long *__restrict__ walk_base,
              const uint8_t *kmers, const LayoutRec *__restrict__ layouts, const int *__restrict__ n_layouts,
              uint8_t *__restrict__ stg_kmers, long long *__restrict__ stg_meta, int pitch, const CountJobs &jobs)
{
    const long long td = (long long)bid * kDelThreads + threadIdx.x;
    if (td >= n_del_walks) return;
    // LDS of the workgroup: [threads][kSiteCache] site records | [threads][W] source positions | [threads][pitch] rows
    extern __shared__ __attribute__((aligned(16))) uint8_t del_stage[];
    SiteRec *cache = reinterpret_cast<SiteRec *>(del_stage) + threadIdx.x;      // [kSiteCache][threads]: see CachedSites
    int *src = reinterpret_cast<int *>(del_stage + (size_t)kDelThreads * kSiteCache * sizeof(SiteRec)) + (size_t)threadIdx.x * W;
    uint8_t *slot = del_stage + (size_t)kDelThreads * (kSiteCache * sizeof(SiteRec) + (size_t)W * sizeof(int)) +
                    (size_t)threadIdx.x * pitch;
    // everything the walk is found from, requested together: the window, its kept layouts, its first site records
    const int m = del_entry[td];
    const DelRec dr = del_rec[m];
    const int q0 = (int)(td - del_base[m]);
    const int nl = min(n_layouts[m], kLayoutCache);
    const long long p = dr.p, limit = dr.limit;
    const int i0 = dr.i0;
#pragma unroll
    for (int k = 0; k < kSiteCache; ++k) cache[k * kDelThreads] = g.site_rec[i0 + k];
    const CachedSites sites{g.site_rec, cache, i0, kDelThreads};
    const long long t = walk_base[dr.w] + q0;          // its place among all walks
    WalkState st;
    WalkStart ws;
    long long q = q0, prod = 0;
    bool found = false;
    {   // the layouts graph_count_del_kernel kept for this window: all eight records requested at once (slots behind
        // nl hold stale bytes and are not looked at), the first whose cumulative count exceeds q0 is the walk's.  (Kept in
        // a local array they went through scratch memory: 128 bytes stored and read back by every thread.)
        int base = 0;
#pragma unroll
        for (int k = 0; k < kLayoutCache; ++k) {
            const LayoutRec rec = layouts[(size_t)m * kLayoutCache + k];
            if (k < nl && !found) {
                if (q0 < rec.cum_end) {
                    found = true;
                    q = q0 - base;
                    prod = rec.cum_end - base;
                    st.nd = (int)(rec.choice >> 24);
                    st.choice = rec.choice & 0xffffffu;
                    ws.site = rec.site;
                    ws.t = rec.t;
                }
                base = rec.cum_end;
            }
        }
    }
    if (!found) {   // beyond the kept layouts: the odometer from the start
        NoVisitor nv;
        while (!found) {                             // starts in order, inside a start the layouts in order
            int prefix = 0;
            for (;;) {                               // skip the layouts that lie before walk q0
                const int rc = simulate(g, sites, p, W, i0, ws, prefix, st, nv, 0, 0, prod, limit);
                if (rc == WALK_OK) {
                    if (q < prod) { found = true; break; }
                    q -= prod;
                }
                prefix = next_walk(st);
                if (prefix < 0) break;
            }
            if (!found && !next_start(g, p, i0, ws)) return;   // cannot happen: q0 < walks of the window
        }
    }
    // the walk's two rows (2 W contiguous bytes at 2 t W of the k-mer matrix, no alignment) are assembled in this
    // thread's LDS slot with the byte phase of that place, and go to the staging slot as dwords
    const int phase = (int)(reinterpret_cast<uintptr_t>(kmers + (size_t)(2 * t) * W) & 3u);
    uint8_t *fwd = slot + phase;
    int more_cons[kMaxConstraints - 4];          // NOT initialised (368 bytes of scratch stores per thread otherwise)
    DelEmit em(g, fwd, fwd + W, src, W, more_cons);
    long long again = 0;
    simulate(g, sites, p, W, i0, ws, st.nd, st, em, q, prod, again, limit);
    for (int j0 = 0; j0 < W; j0 += 8) {              // the reference bases it noted, eight loads in flight
        int sx[8];
        uint8_t c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) sx[u] = j0 + u < W ? src[j0 + u] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = sx[u] >= 0 ? g.ref[sx[u]] : (uint8_t)0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (sx[u] >= 0) {
                fwd[j0 + u] = c[u];
                fwd[2 * W - 1 - (j0 + u)] = complement(c[u]);
            }
    }
    {
        unsigned *out = reinterpret_cast<unsigned *>(stg_kmers + (size_t)td * pitch);
        for (int o = 0; o < pitch / 4; ++o) out[o] = reinterpret_cast<const unsigned *>(slot)[o];
    }
    // the window starts on deleted bases: carriers lack them (not for a walk that never leaves the insertion it starts in)
    if (!(ws.site >= 0 && st.last == p - 1)) for_covering_deletions(g, p, i0, [&](int dsite) { em.add(dsite, 0); });
    auto at = [&](int k, int &site, int &a) { const int v = em.get(k); site = v >> 4; a = v & 3; };
    bool done;
    long long count = count_by_tables(g, allele_count, em.n_cons, at, done);
    bool deferred = false;
    if (!done && jobs.counters) {       // a job for graph_count_jobs_kernel, which writes meta[3]
        const int j = atomicAdd(&jobs.counters[2], 1);
        if (j < jobs.del_jobs) {
            const int off = atomicAdd(&jobs.counters[3], em.n_cons);
            deferred = off + em.n_cons <= jobs.del_pool;
            if (deferred)
                for (int k = 0; k < em.n_cons; ++k) jobs.pool[jobs.plain_pool + off + k] = em.get(k);
            jobs.head[jobs.plain_jobs + j] = JobHead{~td, jobs.plain_pool + off, deferred ? em.n_cons : 0, {em.c0, em.c1, em.c2, em.c3}};
        }
    }
    if (!done && !deferred) count = count_by_bitsets(g, em.n_cons, at);
    long long *meta = stg_meta + (size_t)td * kDelMeta;
    meta[0] = t;
    meta[1] = p;
    meta[2] = st.last + 1;
    if (!deferred) meta[3] = count;
    meta[4] = em.alt ? 0 : 1;
}

__global__ void __launch_bounds__(kEmitThreads)
graph_emit_kernel(GraphDev g, const int *__restrict__ allele_count, const int *__restrict__ walk_window,
                  const int *__restrict__ win_region, const long long *__restrict__ win_start, int W,
                  long long n_walks, const int *__restrict__ first_site, const long long *__restrict__ walk_base,
                  const int *__restrict__ win_sites,
                  uint8_t *__restrict__ kmers, long long *__restrict__ start, long long *__restrict__ stop,
                  uint8_t *__restrict__ strand, long long *__restrict__ freq, uint8_t *__restrict__ is_ref,
                  int *__restrict__ region, int *__restrict__ walk, int wide_rows, CountJobs jobs)
{
    emit_plain_body(blockIdx.x, g, allele_count, walk_window, win_region, win_start, W, n_walks, first_site, walk_base,
                    win_sites, kmers, start, stop, strand, freq, is_ref, region, walk, wide_rows != 0, jobs);
}

__global__ void __launch_bounds__(kDelThreads)
graph_emit_del_kernel(GraphDev g, const int *__restrict__ allele_count,
                      const int *__restrict__ del_entry, const long long *__restrict__ del_base,
                      const DelRec *__restrict__ del_rec, int W, long long n_del_walks,
                      const long long *__restrict__ walk_base,
                      const uint8_t *kmers, const LayoutRec *__restrict__ layouts, const int *__restrict__ n_layouts,
                      uint8_t *__restrict__ stg_kmers, long long *__restrict__ stg_meta, int pitch, CountJobs jobs)
{
    emit_del_body(blockIdx.x, g, allele_count, del_entry, del_base, del_rec, W, n_del_walks, walk_base, kmers, layouts,
                  n_layouts, stg_kmers, stg_meta, pitch, jobs);
}

// thread per deletion walk: staging -> rows 2 t, 2 t + 1 (k-mer bytes as dwords where the addresses allow)
__global__ void __launch_bounds__(kEmitThreads)
graph_del_scatter_kernel(long long n_del_walks, int W, int pitch, const uint8_t *__restrict__ stg_kmers,
                         const long long *__restrict__ stg_meta, uint8_t *__restrict__ kmers,
                         long long *__restrict__ start, long long *__restrict__ stop, long long *__restrict__ freq,
                         uint8_t *__restrict__ is_ref)
{
    const long long td = (long long)blockIdx.x * kEmitThreads + threadIdx.x;
    if (td >= n_del_walks) return;
    const long long *meta = stg_meta + (size_t)td * kDelMeta;
    const long long t = meta[0], p = meta[1], end_pos = meta[2], count = meta[3];
    const uint8_t flag = (uint8_t)meta[4];
    uint8_t *dst = kmers + (size_t)(2 * t) * W;
    const int phase = (int)(reinterpret_cast<uintptr_t>(dst) & 3u);
    const uint8_t *src = stg_kmers + (size_t)td * pitch + phase;
    const int nbytes = 2 * W;
    int o = 0;
    for (; o < nbytes && ((phase + o) & 3); ++o) dst[o] = src[o];
    for (; o + 4 <= nbytes; o += 4) *reinterpret_cast<unsigned *>(dst + o) = *reinterpret_cast<const unsigned *>(src + o);
    for (; o < nbytes; ++o) dst[o] = src[o];
    const long long row = 2 * t;
    start[row] = p;          start[row + 1] = end_pos;
    stop[row] = end_pos;     stop[row + 1] = p;
    freq[row] = count;       freq[row + 1] = count;
    is_ref[row] = is_ref[row + 1] = flag;
}

#include "gfm_graph_fused.hpp"

// `pad` bytes are allocated (and zeroed) behind the array: see load_u64
template <typename T> hipError_t upload(T **dst, const T *src, size_t count, size_t pad = 0)
{
    *dst = nullptr;
    if (count == 0 && pad == 0) return hipSuccess;
    hipError_t e = hipMalloc(dst, sizeof(T) * count + pad);
    if (e != hipSuccess) return e;
    if (pad) {
        e = hipMemset(reinterpret_cast<uint8_t *>(*dst) + sizeof(T) * count, 0, pad);
        if (e != hipSuccess) return e;
    }
    if (count == 0) return hipSuccess;
    return hipMemcpy(*dst, src, sizeof(T) * count, hipMemcpyHostToDevice);
}

// grow-only device buffer: plans are made over and over (one per motif width); hipMalloc / hipFree per
// plan cost more than the kernels
template <typename T> struct Buf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t count)
    {
        if (count <= cap) return hipSuccess;
        (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, sizeof(T) * count);
        if (e == hipSuccess) cap = count;
        return e;
    }
    void release() { (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

// one PLAN of the fused path (see gfm_graph::plans)
struct FusedPlan {
    std::vector<long long> f_starts, f_stops;   // the regions the tile table was built for
    int f_width = 0, f_n_tiles = 0, f_n_general = 0;     // tiles [0, f_n_general): may hold insertions / deletions; the rest: pure
    long long f_n_windows = 0, f_general_windows = 0;
    Buf<Tile> f_tiles;
    Buf<DelWin> f_del_wins;
    Buf<HeavyWin> f_heavy;                 // the plan's heavy windows (graph_heavy_kernel); f_flags[8..9]: their count << 32 | items
    unsigned long long *h_heavy_ctl = nullptr;   // pinned [3]: that word, copied back once per plan (no heavy window: no launch); [1]: the
                                                 // number of work items of graph_del_score_kernel (its grid, once the host knows it);
                                                 // [2]: some item holds more than one round of walks (f_flags[6])
    hipEvent_t ev_heavy = nullptr, ev_items = nullptr;
    bool heavy_known = false, heavy_asked = false, items_known = false, items_asked = false;
    Buf<DelBatchRec> f_del_recs;         // per listed window: what graph_del_count_kernel found
    Buf<DelItem> f_del_items;            // work items of graph_del_score_kernel
    // the listed windows' walks themselves (LwMeta, gfm_graph_fused.hpp): filled by graph_del_score_kernel on the first call that
    // knows the number of work items, read by graph_score_kernel's wavefronts from then on
    Buf<unsigned char> f_lw_kmers;
    Buf<LwMeta> f_lw_meta;
    int lw_state = 0, lw_pitch = 0, lw_items = 0;      // 0: not decided, 1: filled (stream order), -1: this plan does without
    // [0] unused, [1] listed windows, [2] overflow of the call, [3] work items of the deletion kernels, [4] overflow among the
    // listed windows.  [1], [3], [4] belong to the PLAN -- the list of windows that touch an indel, their layouts and the
    // work items cut from them depend on the graph, the regions and the width, not on the motif
    Buf<int> f_flags;
    bool f_plan_ready = false;
    hipStream_t f_plan_stream = nullptr;   // the stream the plan was made on; a call on another one waits for ev_plan
    hipEvent_t ev_plan = nullptr;
    hipError_t init()
    {
        hipError_t e = hipEventCreateWithFlags(&ev_plan, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_heavy, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_items, hipEventDisableTiming);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&h_heavy_ctl), 3 * sizeof(unsigned long long), hipHostMallocDefault);
        if (e == hipSuccess) h_heavy_ctl[0] = h_heavy_ctl[1] = h_heavy_ctl[2] = 0ull;
        return e;
    }
    ~FusedPlan()
    {
        if (ev_plan) (void)hipEventDestroy(ev_plan);
        if (ev_heavy) (void)hipEventDestroy(ev_heavy);
        if (ev_items) (void)hipEventDestroy(ev_items);
        if (h_heavy_ctl) (void)hipHostFree(h_heavy_ctl);
        f_tiles.release(); f_del_wins.release(); f_heavy.release(); f_del_recs.release(); f_del_items.release(); f_flags.release();
        f_lw_kmers.release(); f_lw_meta.release();
    }
};

struct gfm_graph {
    GraphDev dev{};
    uint8_t *d_ref = nullptr;
    int *d_pos = nullptr;
    SiteRec *d_site_rec = nullptr, *d_site_pk = nullptr;
    uint8_t *d_n_alts = nullptr, *d_alt_bases = nullptr;
    unsigned long long *d_alt_bits = nullptr;
    int *d_allele_count = nullptr;   // [n_sites][4] haplotypes per allele (0 = reference)
    int *d_pair_count = nullptr, *d_triple_count = nullptr;   // GraphDev::pair_count / triple_count
    int *d_del_len = nullptr, *d_prev_del = nullptr;
    long long *d_max_reach = nullptr;
    int *d_ins_len = nullptr, *d_ins_off = nullptr;
    uint8_t *d_ins_bases = nullptr;
    // last plan (buffers are kept between plans)
    int n_regions = 0, width = 0;
    long long n_windows = 0, n_walks = 0;
    Buf<long long> region_off, first_start, region_stop, walk_base, win_start, walks;
    Buf<int> first_site, win_region, walk_window, flag, win_sites;
    Buf<unsigned char> scan_tmp;
    Buf<int> del_list, del_entry;
    Buf<long long> del_walks, del_base;
    Buf<uint8_t> stg_kmers;              // staging of the deletion walks' rows (emit_del_body -> graph_del_scatter_kernel)
    Buf<long long> stg_meta;
    Buf<LayoutRec> del_layouts;          // [listed windows][kLayoutCache]
    Buf<DelRec> del_rec;                 // [listed windows]
    Buf<int> del_layout_n;
    Buf<JobHead> job_head;               // CountJobs of the last plan's emits
    Buf<int> job_pool;
    CountJobs jobs{};
    long long n_del_walks = 0;
    // plan runs on the NULL stream, emit on the caller's: ordered through these events (a non-blocking
    // caller stream is not ordered against the NULL stream by itself), both ways -- emit waits for the
    // plan's last kernels, the next plan waits for the emit that still reads the plan buffers
    hipEvent_t ev_planned = nullptr, ev_emitted = nullptr;
    long long *h_back = nullptr;         // pinned: what a plan reads back (gfm_graph_plan)
    hipStream_t side = nullptr;          // the plain walks' kernels run here when there are deletion walks, beside the plain walks' (gfm_graph_emit)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool emit_pending = false;
    // ---- the TSV writer (gfm_graph_write_tsvs): the site arrays on the host, two pinned staging buffers, a copy stream
    gfm_host::HostGraph host;
    void *h_stage[2] = {nullptr, nullptr};
    size_t h_stage_cap = 0;
    hipStream_t copy_st = nullptr;
    hipEvent_t ev_copy[2] = {nullptr, nullptr};
    // ---- fused extraction -> scoring (gfm_graph_score / gfm_graph_annotate)
    std::vector<int> h_pos;              // host copy of the site positions: the tiles' first sites are found here
    // measurement aid (gfm_graph_profile_enable): event pairs around graph_score_kernel's launches
    static constexpr int kProfSlots = 64;
    hipEvent_t prof_ev[2 * kProfSlots] = {};
    int prof_on = 0, prof_n = 0;
    std::vector<int> h_indel_prefix;     // [n_sites + 1] insertion / deletion records among sites [0, i) (pure tiles; made on first use)
    // ---- the fused path's PLANS: what depends on (graph, regions, width) only -- the tile table, the list of the windows that
    // touch an insertion / deletion with their layouts and work items, the list of the heavy windows -- is made by the first call
    // for a (regions, width) and kept; a handle keeps several (kMaxPlans, least recently used goes): GRAFIMO scans motif after
    // motif over one BED file (grafimo.py:177-183) and the widths of a motif set alternate (building one for 50 000 regions
    // took 12 ms of host time a call when only the last was kept)
    static constexpr int kMaxPlans = 32;
    std::vector<FusedPlan *> plans;      // most recently used first
    FusedPlan *plan = nullptr;           // of the last gfm_graph_score[_multi] call (what gfm_graph_annotate refers to)
    Tile *h_tiles = nullptr;             // pinned staging of a tile table on its way to the device
    size_t h_tiles_cap = 0;
    hipEvent_t ev_tiles = nullptr;       // the staging has been copied
    bool tiles_pending = false;
    long long max_del_len = 0;             // the graph's longest deletion (how far behind a window a one-deletion scan looks)
    Buf<unsigned> f_slabs;
#ifdef GFM_LAB
    Buf<unsigned long long> f_dbg;
#endif
    // The fused calls of ONE handle share its scratch (overflow word, slabs, heavy list, tile table): they are serialised --
    // a gfm_graph_score / gfm_graph_annotate on another stream than the handle's last call waits for that call's work.
    hipStream_t f_last_stream = nullptr;
    hipEvent_t ev_call = nullptr;
    bool call_pending = false;
    int serialise(hipStream_t st)
    {
        if (call_pending && st != f_last_stream) {
            const hipError_t e = hipStreamWaitEvent(st, ev_call, 0);
            if (e != hipSuccess) return gfail(GFM_ERR_HIP, std::string("hipStreamWaitEvent failed: ") + hipGetErrorString(e));
        }
        return GFM_OK;
    }
    int called(hipStream_t st)
    {
        const hipError_t e = hipEventRecord(ev_call, st);
        if (e != hipSuccess) return gfail(GFM_ERR_HIP, std::string("hipEventRecord failed: ") + hipGetErrorString(e));
        f_last_stream = st;
        call_pending = true;
        return GFM_OK;
    }
    void drop_plan()
    {
        region_off.release(); first_start.release(); region_stop.release(); walk_base.release(); win_start.release(); walks.release();
        first_site.release(); win_region.release(); walk_window.release(); flag.release(); scan_tmp.release(); win_sites.release();
        del_list.release(); del_entry.release(); del_walks.release(); del_base.release();
        del_layouts.release(); del_layout_n.release(); del_rec.release(); stg_kmers.release(); stg_meta.release();
        job_head.release(); job_pool.release();
        jobs = CountJobs{};
        n_del_walks = 0;
        n_regions = 0;
        n_windows = n_walks = 0;
    }
};

GFM_API int gfm_graph_create(const uint8_t *h_ref, int64_t ref_len, int32_t n_sites, const int32_t *h_pos,
                             const uint8_t *h_n_alts, const uint8_t *h_alt_bases, const int32_t *h_del_len,
                             const int32_t *h_ins_len, const int32_t *h_ins_off, const uint8_t *h_ins_bases,
                             int64_t ins_bytes, const uint64_t *h_alt_bits, int32_t n_haplotypes, gfm_graph_t *out)
{
    if (!out) return gfail(GFM_ERR_INVALID, "NULL output handle");
    *out = nullptr;
    if (!h_ref || ref_len <= 0 || n_sites < 0 || n_haplotypes < 0)
        return gfail(GFM_ERR_INVALID, "bad reference / site count");
    if (n_sites && (!h_pos || !h_n_alts || !h_alt_bases)) return gfail(GFM_ERR_INVALID, "NULL site arrays");
    std::vector<int> del_len((size_t)n_sites, 0), prev_del((size_t)n_sites + 1, -1), ins_len((size_t)n_sites, 0),
        ins_off((size_t)n_sites, 0);
    std::vector<long long> max_reach((size_t)n_sites + 1, -1);
    int n_dels = 0, n_ins = 0;
    long long max_del_len = 0;
    long long deleted_until = -1;        // last reference position removed by an earlier deletion
    auto kind_of = [&](int i) { return del_len[(size_t)i] > 0 ? 2 : (ins_len[(size_t)i] > 0 ? 1 : 0); };
    for (int i = 0; i < n_sites; ++i) {
        const int dl = h_del_len ? h_del_len[i] : 0;
        const int il = h_ins_len ? h_ins_len[i] : 0;
        if (dl < 0 || il < 0 || (dl > 0 && il > 0))
            return gfail(GFM_ERR_INVALID, "a site is a substitution, an insertion or a deletion (site " + std::to_string(i) + ")");
        del_len[(size_t)i] = dl;
        ins_len[(size_t)i] = il;
        if (il > 0) {
            const long long off = h_ins_off ? h_ins_off[i] : -1;
            if (!h_ins_bases || off < 0 || off + il > ins_bytes)
                return gfail(GFM_ERR_INVALID, "inserted bases outside the pool (site " + std::to_string(i) + ")");
            ins_off[(size_t)i] = (int)off;
            ++n_ins;
        }
        // same position: the substitution site first, then insertions, then the deletions
        const bool tie_ok = i && h_pos[i] == h_pos[i - 1] &&
                            (kind_of(i) > kind_of(i - 1) || (kind_of(i) == kind_of(i - 1) && kind_of(i) >= 1));
        if (h_pos[i] < 0 || h_pos[i] >= ref_len || (i && h_pos[i] <= h_pos[i - 1] && !tie_ok))
            return gfail(GFM_ERR_INVALID, "site positions must be ascending inside the reference; at one position the "
                                          "substitution site comes first, then insertions, then the deletion (site " +
                                              std::to_string(i) + ")");
        if (h_n_alts[i] < 1 || h_n_alts[i] > kMaxAlts)
            return gfail(GFM_ERR_INVALID, "a site needs 1..3 alternate alleles (site " + std::to_string(i) + ")");
        if ((dl > 0 || il > 0) && h_n_alts[i] != 1)
            return gfail(GFM_ERR_INVALID, "an insertion / a deletion has one alternate allele (site " +
                                              std::to_string(i) + ")");
        if (dl > 0) {
            deleted_until = std::max(deleted_until, (long long)h_pos[i] + dl);
            ++n_dels;
            max_del_len = std::max(max_del_len, (long long)dl);
        }
        prev_del[(size_t)i + 1] = dl > 0 ? i : prev_del[(size_t)i];
        max_reach[(size_t)i + 1] = deleted_until;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        (void)hipGetLastError();
        return gfail(GFM_ERR_NODEVICE, "no HIP device available");
    }
    gfm_graph *g = new (std::nothrow) gfm_graph();
    if (!g) return gfail(GFM_ERR_NOMEM, "out of host memory");
    const int hw = (n_haplotypes + 63) / 64;
    const bool bits = h_alt_bits && n_haplotypes > 0 && n_sites > 0;
    hipError_t e = upload(&g->d_ref, h_ref, (size_t)ref_len, kReadPad);
    if (e == hipSuccess) e = upload(&g->d_pos, h_pos, (size_t)n_sites, kReadPad);
    if (e == hipSuccess) e = upload(&g->d_n_alts, h_n_alts, (size_t)n_sites, kReadPad);
    if (e == hipSuccess) e = upload(&g->d_alt_bases, h_alt_bases, (size_t)n_sites * kMaxAlts);
    if (e == hipSuccess) e = upload(&g->d_del_len, del_len.data(), del_len.size());
    if (e == hipSuccess) e = upload(&g->d_prev_del, prev_del.data(), prev_del.size());
    if (e == hipSuccess) e = upload(&g->d_max_reach, max_reach.data(), max_reach.size());
    if (e == hipSuccess) e = upload(&g->d_ins_len, ins_len.data(), ins_len.size());
    if (e == hipSuccess) e = upload(&g->d_ins_off, ins_off.data(), ins_off.size());
    if (e == hipSuccess && n_ins) e = upload(&g->d_ins_bases, h_ins_bases, (size_t)ins_bytes);
    if (e == hipSuccess) {
        std::vector<SiteRec> recs((size_t)n_sites + kSitePad, SiteRec{kNoSitePos, 0, 0, 0});
        for (int i = 0; i < n_sites; ++i) recs[(size_t)i] = SiteRec{h_pos[i], del_len[(size_t)i], ins_len[(size_t)i], h_n_alts[i]};
        e = upload(&g->d_site_rec, recs.data(), recs.size());
        for (int i = 0; i < n_sites && e == hipSuccess; ++i) {
            const uint8_t *ab = h_alt_bases + (size_t)i * kMaxAlts;
            recs[(size_t)i].n_alts |= ((int)ab[0] << 8) | ((int)ab[1] << 16) | ((int)ab[2] << 24);
        }
        if (e == hipSuccess) e = upload(&g->d_site_pk, recs.data(), recs.size());
    }
    if (e == hipSuccess && bits)
        e = upload(&g->d_alt_bits, reinterpret_cast<const unsigned long long *>(h_alt_bits),
                   (size_t)n_sites * kMaxAlts * hw);
    if (e != hipSuccess) {
        gfm_graph_destroy(g);
        return gfail(GFM_ERR_HIP, std::string("graph upload failed: ") + hipGetErrorString(e));
    }
    g->h_pos.assign(h_pos, h_pos + n_sites);
    g->max_del_len = max_del_len;
    g->host.ref_len = ref_len;
    g->host.pos.assign(h_pos, h_pos + n_sites);
    g->host.del_len = del_len;
    g->host.ins_len = ins_len;
    g->host.n_alts.assign(h_n_alts, h_n_alts + n_sites);
    g->host.max_reach = max_reach;
    g->host.has_ins = n_ins > 0;
    g->dev = GraphDev{g->d_site_rec, g->d_ref, (long long)ref_len, n_sites, g->d_pos, g->d_n_alts, g->d_alt_bases,
                      bits ? g->d_alt_bits : nullptr, bits ? n_haplotypes : 0, bits ? hw : 0,
                      g->d_del_len, n_dels, g->d_prev_del, g->d_max_reach, g->d_ins_len, g->d_ins_off, g->d_ins_bases,
                      n_ins, nullptr, nullptr, g->d_site_pk};
    if (e == hipSuccess)     // the widest windows need a little more than 64 KB of LDS per workgroup of deletion walks
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(graph_emit_del_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(kDelThreads * (kSiteCache * sizeof(SiteRec) + GFM_MAX_WIDTH * sizeof(int) + 2 * GFM_MAX_WIDTH + 8)));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&g->h_back), 8 * sizeof(long long), hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_planned, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_emitted, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_tiles, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_call, hipEventDisableTiming);
    if (e != hipSuccess) {
        gfm_graph_destroy(g);
        return gfail(GFM_ERR_HIP, std::string("event creation failed: ") + hipGetErrorString(e));
    }
    if (bits) {
        e = hipMalloc(&g->d_allele_count, sizeof(int) * 4 * (size_t)n_sites);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(graph_allele_count_kernel, dim3((n_sites + 255) / 256), dim3(256), 0, nullptr, g->dev,
                               g->d_allele_count);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMalloc(&g->d_pair_count, sizeof(int) * 16 * (size_t)n_sites);
        if (e == hipSuccess) e = hipMalloc(&g->d_triple_count, sizeof(int) * 64 * (size_t)n_sites);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(graph_pair_count_kernel, dim3((unsigned)(((size_t)n_sites * 16 + 255) / 256)), dim3(256), 0,
                               nullptr, g->dev, g->d_pair_count);
            hipLaunchKernelGGL(graph_triple_count_kernel, dim3((unsigned)(((size_t)n_sites * 64 + 255) / 256)), dim3(256), 0,
                               nullptr, g->dev, g->d_triple_count);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            gfm_graph_destroy(g);
            return gfail(GFM_ERR_HIP, std::string("allele counts failed: ") + hipGetErrorString(e));
        }
        g->dev.pair_count = g->d_pair_count;
        g->dev.triple_count = g->d_triple_count;
    }
    *out = g;
    return GFM_OK;
}

GFM_API void gfm_graph_destroy(gfm_graph_t g)
{
    if (!g) return;
    g->drop_plan();
    (void)hipFree(g->d_ref); (void)hipFree(g->d_pos); (void)hipFree(g->d_n_alts); (void)hipFree(g->d_site_rec); (void)hipFree(g->d_site_pk);
    (void)hipFree(g->d_alt_bases); (void)hipFree(g->d_alt_bits); (void)hipFree(g->d_allele_count);
    (void)hipFree(g->d_pair_count); (void)hipFree(g->d_triple_count);
    (void)hipFree(g->d_del_len); (void)hipFree(g->d_prev_del); (void)hipFree(g->d_max_reach);
    (void)hipFree(g->d_ins_len); (void)hipFree(g->d_ins_off); (void)hipFree(g->d_ins_bases);
    if (g->ev_planned) (void)hipEventDestroy(g->ev_planned);
    if (g->ev_emitted) (void)hipEventDestroy(g->ev_emitted);
    if (g->ev_fork) (void)hipEventDestroy(g->ev_fork);
    if (g->ev_join) (void)hipEventDestroy(g->ev_join);
    if (g->side) (void)hipStreamDestroy(g->side);
    if (g->h_back) (void)hipHostFree(g->h_back);
    if (g->ev_tiles) (void)hipEventDestroy(g->ev_tiles);
    if (g->ev_call) (void)hipEventDestroy(g->ev_call);
    if (g->h_tiles) (void)hipHostFree(g->h_tiles);
    for (int k = 0; k < 2; ++k) {
        if (g->h_stage[k]) (void)hipHostFree(g->h_stage[k]);
        if (g->ev_copy[k]) (void)hipEventDestroy(g->ev_copy[k]);
    }
    if (g->copy_st) (void)hipStreamDestroy(g->copy_st);
    for (hipEvent_t e : g->prof_ev)
        if (e) (void)hipEventDestroy(e);
    for (FusedPlan *pl : g->plans) delete pl;
    g->plans.clear();
    g->plan = nullptr;
    g->f_slabs.release();
    delete g;
}

GFM_API int gfm_graph_plan(gfm_graph_t g, int32_t n_regions, const int64_t *h_starts, const int64_t *h_stops,
                           int32_t width, int64_t *n_windows, int64_t *n_rows)
{
    if (!g || n_regions < 0 || (n_regions && (!h_starts || !h_stops)))
        return gfail(GFM_ERR_INVALID, "bad argument");
    if (width < 1 || width > GFM_MAX_WIDTH) return gfail(GFM_ERR_INVALID, "width outside [1, 64]");
    // windows of region r: starts p in [max(S,0), min(E, ref_len) - W]  (vg find -p S-E -K W, pinned by
    // expected_seqs.tsv: x:0-20, W=19 -> p in {0, 1}); with insertions in the graph a walk that reads inserted bases
    // may start later: p up to E - 1
    std::vector<int64_t> first((size_t)n_regions), last((size_t)n_regions), limit((size_t)n_regions);
    for (int r = 0; r < n_regions; ++r) {
        const long long s = std::max<long long>(h_starts[r], 0);
        const long long e = std::min<long long>(h_stops[r], g->dev.ref_len);
        first[(size_t)r] = s;
        limit[(size_t)r] = e;
        last[(size_t)r] = e - (g->dev.n_ins > 0 ? 1 : width);
    }
    return gfm_graph_plan_windows(g, n_regions, first.data(), last.data(), limit.data(), width, n_windows, n_rows);
}

GFM_API int gfm_graph_plan_windows(gfm_graph_t g, int32_t n_ranges, const int64_t *h_first, const int64_t *h_last,
                                   const int64_t *h_limit, int32_t width, int64_t *n_windows, int64_t *n_rows)
{
    const int32_t n_regions = n_ranges;
    if (!g || n_regions < 0 || (n_regions && (!h_first || !h_last || !h_limit)))
        return gfail(GFM_ERR_INVALID, "bad argument");
    if (width < 1 || width > GFM_MAX_WIDTH) return gfail(GFM_ERR_INVALID, "width outside [1, 64]");
    if (g->emit_pending) {   // the last emit may still read the plan buffers this call rewrites / frees
        GX_TRY(hipEventSynchronize(g->ev_emitted));
        g->emit_pending = false;
    }
    g->n_windows = g->n_walks = 0;
    g->width = width;
    g->n_regions = n_regions;
    std::vector<long long> off(n_regions + 1, 0), first(n_regions, 0), rstop(n_regions, 0);
    for (int r = 0; r < n_regions; ++r) {
        const long long s = std::max<long long>(h_first[r], 0);
        const long long e = std::min<long long>(h_limit[r], g->dev.ref_len);
        first[r] = s;
        rstop[r] = e;
        // (a plain window must end inside its region; with insertions in the graph the kernels check that per walk)
        off[r + 1] = off[r] + std::max<long long>(0, std::min<long long>(h_last[r], e - (g->dev.n_ins > 0 ? 1 : width)) - s + 1);
    }
    if (n_windows) *n_windows = off[n_regions];
    if (n_rows) *n_rows = 0;
    if (off[n_regions] == 0) return GFM_OK;
    if (off[n_regions] > 0x7fffffffll) return gfail(GFM_ERR_INVALID, "too many windows in one plan (split the regions)");
    const size_t nw = (size_t)off[n_regions];
    GX_TRY(g->region_off.reserve(off.size()));
    GX_TRY(g->first_start.reserve(first.size()));
    GX_TRY(g->region_stop.reserve(rstop.size()));
    GX_TRY(g->first_site.reserve(nw));
    GX_TRY(g->walks.reserve(nw));
    // [0] overflow flag, [1] number of windows that touch a deletion, [2..5] two 64-bit totals (plain walks that need the
    // bitsets, their constraints), [6..9] CountJobs::counters of an emit
    GX_TRY(g->flag.reserve(10));
    GX_TRY(g->win_region.reserve(nw));
    GX_TRY(g->win_start.reserve(nw));
    GX_TRY(g->win_sites.reserve(nw));
    GX_TRY(g->walk_base.reserve(nw + 1));
    const bool dels = g->dev.n_dels > 0 || g->dev.n_ins > 0;   // windows that need the layout enumeration
    GX_TRY(g->del_list.reserve(dels ? nw : 1));
    GX_TRY(hipMemcpyAsync(g->region_off.p, off.data(), sizeof(long long) * off.size(), hipMemcpyHostToDevice, nullptr));
    GX_TRY(hipMemcpyAsync(g->first_start.p, first.data(), sizeof(long long) * first.size(), hipMemcpyHostToDevice, nullptr));
    GX_TRY(hipMemcpyAsync(g->region_stop.p, rstop.data(), sizeof(long long) * rstop.size(), hipMemcpyHostToDevice, nullptr));
    GX_TRY(hipMemsetAsync(g->flag.p, 0, 10 * sizeof(int), nullptr));
    GX_TRY(hipMemsetAsync(g->walk_base.p, 0, sizeof(long long), nullptr));
    // walks per window -> inclusive prefix (row base of every window) on the device: only the totals and the
    // overflow flag come back
    const unsigned blocks = (unsigned)((nw + kCountThreads - 1) / kCountThreads);
    hipLaunchKernelGGL(graph_count_kernel, dim3(blocks), dim3(kCountThreads), 0, nullptr, g->dev, n_regions,
                       g->region_off.p, g->first_start.p, g->region_stop.p, width, (long long)nw, g->first_site.p, g->walks.p,
                       g->win_region.p, g->win_start.p, g->flag.p, g->del_list.p, g->flag.p + 1, g->win_sites.p,
                       reinterpret_cast<unsigned long long *>(g->flag.p + 2));
    GX_TRY(hipGetLastError());
    int n_listed = 0;
    if (dels) {
        // how many windows need the layout enumeration: sizes their buffers and launches exactly (a grid over ALL
        // windows whose threads mostly return at once took as long as the counting itself)
        GX_TRY(hipMemcpy(&n_listed, g->flag.p + 1, sizeof n_listed, hipMemcpyDeviceToHost));
    }
    const unsigned lblocks = (unsigned)((n_listed + kCountThreads - 1) / kCountThreads);
    if (n_listed > 0) {
        GX_TRY(g->del_walks.reserve((size_t)n_listed));
        GX_TRY(g->del_base.reserve((size_t)n_listed + 1));
        GX_TRY(g->del_layouts.reserve((size_t)n_listed * kLayoutCache));
        GX_TRY(g->del_layout_n.reserve((size_t)n_listed));
        GX_TRY(g->del_rec.reserve((size_t)n_listed));
        GX_TRY(hipMemsetAsync(g->del_base.p, 0, sizeof(long long), nullptr));
        hipLaunchKernelGGL(graph_count_del_kernel, dim3(lblocks), dim3(kCountThreads), 0, nullptr, g->dev, g->del_list.p,
                           g->flag.p + 1, g->win_start.p, g->win_region.p, g->region_stop.p, width, g->first_site.p,
                           g->walks.p, g->del_walks.p, g->flag.p, g->del_layouts.p, g->del_layout_n.p, g->del_rec.p);
        GX_TRY(hipGetLastError());
    }
    size_t tmp_bytes = 0;
    GX_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, g->walks.p, g->walk_base.p + 1, (int)nw, nullptr));
    GX_TRY(g->scan_tmp.reserve(tmp_bytes));
    GX_TRY(hipcub::DeviceScan::InclusiveSum(g->scan_tmp.p, tmp_bytes, g->walks.p, g->walk_base.p + 1, (int)nw, nullptr));
    long long total_del = 0;
    if (n_listed > 0) {   // same scan over the listed windows (the temporary storage of the larger scan is enough)
        size_t tmp2 = 0;
        GX_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, tmp2, g->del_walks.p, g->del_base.p + 1, n_listed, nullptr));
        GX_TRY(g->scan_tmp.reserve(std::max(tmp_bytes, tmp2)));
        tmp2 = std::max(tmp_bytes, tmp2);
        GX_TRY(hipcub::DeviceScan::InclusiveSum(g->scan_tmp.p, tmp2, g->del_walks.p, g->del_base.p + 1, n_listed, nullptr));
    }
    // the plan's totals come back together: four copies into pinned memory, ONE wait (each blocking copy was a
    // synchronisation of its own, ~30 us); the wait also orders the host vectors above
    long long *back = g->h_back;          // [0] walks, [1] deletion walks, [2..3] slow totals, [4] overflow flag
    back[0] = back[1] = back[2] = back[3] = back[4] = 0;
    GX_TRY(hipMemcpyAsync(&back[0], g->walk_base.p + nw, sizeof(long long), hipMemcpyDeviceToHost, nullptr));
    if (n_listed > 0)
        GX_TRY(hipMemcpyAsync(&back[1], g->del_base.p + n_listed, sizeof(long long), hipMemcpyDeviceToHost, nullptr));
    GX_TRY(hipMemcpyAsync(&back[2], g->flag.p + 2, 2 * sizeof(long long), hipMemcpyDeviceToHost, nullptr));
    GX_TRY(hipMemcpyAsync(&back[4], g->flag.p, sizeof(int), hipMemcpyDeviceToHost, nullptr));
    GX_TRY(hipStreamSynchronize(nullptr));
    const long long total = back[0];
    total_del = back[1];
    const int overflow = (int)back[4];
    if (overflow) return gfail(GFM_ERR_OVERFLOW, "a window holds more than 2^30 walks through its sites: its rows do not fit one plan");
    // (GRAFIMO_PLAN_MAX_WALKS: test aid -- a small cap makes callers cut their regions into several plans)
    static const long long plan_cap = [] { const char *e = std::getenv("GRAFIMO_PLAN_MAX_WALKS"); return e ? std::min(atoll(e), 0x3fffffffll) : 0x3fffffffll; }();
    if (total > plan_cap) return gfail(GFM_ERR_OVERFLOW, "more than 2^31 rows in one plan (split the regions)");
    if (total > 0) {
        GX_TRY(g->walk_window.reserve((size_t)total));
        hipLaunchKernelGGL(graph_map_kernel, dim3(blocks), dim3(kCountThreads), 0, nullptr, (long long)nw,
                           g->walk_base.p, g->walk_window.p);
        GX_TRY(hipGetLastError());
    }
    if (total_del > 0) {
        GX_TRY(g->del_entry.reserve((size_t)total_del));
        GX_TRY(g->stg_kmers.reserve((size_t)total_del * (size_t)((2 * width + 6) & ~3)));
        GX_TRY(g->stg_meta.reserve((size_t)total_del * kDelMeta));
        hipLaunchKernelGGL(graph_del_map_kernel, dim3(lblocks), dim3(kCountThreads), 0, nullptr, g->flag.p + 1,
                           g->del_base.p, g->del_entry.p);
        GX_TRY(hipGetLastError());
    }
    g->jobs = CountJobs{};
    if (g->dev.alt_bits && total > 0) {   // room for the walks whose haplotype count is deferred (see CountJobs)
        const unsigned long long slow[2] = {(unsigned long long)back[2], (unsigned long long)back[3]};
        unsigned long long del_pool = 8ull * (unsigned long long)total_del + (total_del ? 4096ull : 0ull);
        if (const char *e = std::getenv("GRAFIMO_EXTRACT_DEL_POOL"))   // test aid: a small pool makes deletion walks count in place
            del_pool = std::min<unsigned long long>(del_pool, strtoull(e, nullptr, 10));
        if (slow[0] + (unsigned long long)total_del > 0x7fffffffull || slow[1] + del_pool > 0x7fffffffull)
            return gfail(GFM_ERR_OVERFLOW, "too many multi-site walks in one plan (split the regions)");
        GX_TRY(g->job_head.reserve((size_t)(slow[0] + (unsigned long long)total_del) + 1));
        GX_TRY(g->job_pool.reserve((size_t)(slow[1] + del_pool) + 1));
        g->jobs = CountJobs{g->flag.p + 6, g->job_head.p, g->job_pool.p, (int)slow[0], (int)slow[1], (int)total_del, (int)del_pool};
    }
    GX_TRY(hipEventRecord(g->ev_planned, nullptr));   // behind the map kernels
    g->n_windows = (long long)nw;
    g->n_walks = total;
    g->n_del_walks = total_del;
    if (n_rows) *n_rows = 2 * g->n_walks;
    return GFM_OK;
}

GFM_API int gfm_graph_emit(gfm_graph_t g, uint8_t *d_kmers, int64_t *d_start, int64_t *d_stop,
                           uint8_t *d_strand, int64_t *d_freq, uint8_t *d_is_ref, int32_t *d_region,
                           int32_t *d_walk, void *stream)
{
    if (!g) return gfail(GFM_ERR_INVALID, "graph is NULL");
    if (g->n_walks == 0) return GFM_OK;
    if (!d_kmers || !d_start || !d_stop || !d_strand || !d_freq || !d_is_ref || !d_region || !d_walk)
        return gfail(GFM_ERR_INVALID, "NULL output buffer");
    const unsigned blocks = (unsigned)((g->n_walks + kEmitThreads - 1) / kEmitThreads);
    GX_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), g->ev_planned, 0));
    // an earlier emit of this handle (possibly on another stream) still owns the staging area and the job lists
    if (g->emit_pending) GX_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), g->ev_emitted, 0));
    const unsigned dblocks = (unsigned)((g->n_del_walks + kDelThreads - 1) / kDelThreads);
    const unsigned sblocks = (unsigned)((g->n_del_walks + kEmitThreads - 1) / kEmitThreads);
    const int pitch = (2 * g->width + 6) & ~3;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // rows 2t, 2t+1 of a walk leave as one store per column where the columns' addresses allow it
    const int wide_rows = ((reinterpret_cast<uintptr_t>(d_start) | reinterpret_cast<uintptr_t>(d_stop) |
                            reinterpret_cast<uintptr_t>(d_freq)) & 15u) == 0 &&
                          ((reinterpret_cast<uintptr_t>(d_region) | reinterpret_cast<uintptr_t>(d_walk)) & 7u) == 0 &&
                          ((reinterpret_cast<uintptr_t>(d_strand) | reinterpret_cast<uintptr_t>(d_is_ref)) & 1u) == 0;
    // measurement aid: GRAFIMO_EXTRACT_SERIAL=1 runs the two emit kernels one after the other on the caller's stream,
    // so that a kernel trace shows what each takes alone
#ifdef GFM_LAB
    static const bool serial = [] { const char *e = std::getenv("GRAFIMO_EXTRACT_SERIAL"); return e && *e == '1'; }();
#else
    constexpr bool serial = false;
#endif
    const CountJobs jobs = g->jobs;
    if (jobs.counters) GX_TRY(hipMemsetAsync(jobs.counters, 0, 4 * sizeof(int), st));
    // The deletion walks' chain -- kernel, its count jobs, the scatter of its staged rows -- stays on the caller's stream;
    // the plain walks and their count jobs go to the side stream and are done before the scatter needs their placeholder
    // rows.  Which of the two gets the side stream does not matter for the total (scripts/extract_timeline.sh: 175 us from
    // the first kernel's start to the scatter's end either way): the kernels share the chip's wave slots -- whichever starts
    // first runs near its stand-alone time (deletions 55 -> 84 us, plain 95 -> 105 us), the other one stretches (130 us) --
    // and what counts is the sum of their work.
    hipStream_t plain_st = (dblocks && !serial) ? g->side : st;
    if (dblocks) {
        if (!serial) {
            GX_TRY(hipEventRecord(g->ev_fork, st));
            GX_TRY(hipStreamWaitEvent(g->side, g->ev_fork, 0));
        }
        const size_t del_lds = (size_t)kDelThreads * (kSiteCache * sizeof(SiteRec) + (size_t)g->width * sizeof(int) + (size_t)pitch);
        hipLaunchKernelGGL(graph_emit_del_kernel, dim3(dblocks), dim3(kDelThreads), del_lds, st,
                           g->dev, g->d_allele_count, g->del_entry.p, g->del_base.p, g->del_rec.p, g->width, g->n_del_walks,
                           g->walk_base.p, d_kmers, g->del_layouts.p, g->del_layout_n.p, g->stg_kmers.p, g->stg_meta.p,
                           pitch, jobs);
        if (jobs.counters && jobs.del_jobs > 0)      // the counts it left to graph_count_jobs_kernel
            hipLaunchKernelGGL(graph_count_jobs_kernel, dim3(4096), dim3(256), 0, st, g->dev, jobs, 1,
                               reinterpret_cast<long long *>(d_freq), g->stg_meta.p, kDelMeta);
    }
    hipLaunchKernelGGL(graph_emit_kernel, dim3(blocks), dim3(kEmitThreads), (size_t)2 * kEmitThreads * g->width, plain_st,
                       g->dev, g->d_allele_count, g->walk_window.p, g->win_region.p, g->win_start.p, g->width, g->n_walks,
                       g->first_site.p, g->walk_base.p, g->win_sites.p, d_kmers, reinterpret_cast<long long *>(d_start),
                       reinterpret_cast<long long *>(d_stop), d_strand, reinterpret_cast<long long *>(d_freq), d_is_ref,
                       d_region, d_walk, wide_rows, jobs);
    if (jobs.counters && jobs.plain_jobs > 0)    // the counts the plain kernel left to it
        hipLaunchKernelGGL(graph_count_jobs_kernel, dim3(2048), dim3(256), 0, plain_st, g->dev, jobs, 0,
                           reinterpret_cast<long long *>(d_freq), g->stg_meta.p, kDelMeta);
    if (dblocks) {   // join: the staged rows over the placeholders the plain kernel wrote
        if (!serial) {
            GX_TRY(hipEventRecord(g->ev_join, g->side));
            GX_TRY(hipStreamWaitEvent(st, g->ev_join, 0));
        }
        hipLaunchKernelGGL(graph_del_scatter_kernel, dim3(sblocks), dim3(kEmitThreads), 0, st,
                           g->n_del_walks, g->width, pitch, g->stg_kmers.p, g->stg_meta.p, d_kmers,
                           reinterpret_cast<long long *>(d_start), reinterpret_cast<long long *>(d_stop),
                           reinterpret_cast<long long *>(d_freq), d_is_ref);
    }
    GX_TRY(hipGetLastError());
    GX_TRY(hipEventRecord(g->ev_emitted, static_cast<hipStream_t>(stream)));
    g->emit_pending = true;
    return GFM_OK;
}

// ------------------------------------------------------------------------------------------- the TSV files of scan_graph
// Rows travel in chunks: device -> one of two pinned staging buffers (copy stream) while host threads format the chunk before
// (graph_tsv_writer.cpp); a chunk is [kmers | start | stop | freq | region | walk | strand | is_ref] of <= kTsvChunkRows rows.
namespace {
constexpr long long kTsvChunkRows = 1ll << 21;
struct StageView { unsigned char *kmers; long long *start, *stop, *freq; int *region, *walk; unsigned char *strand, *is_ref; };
size_t stage_bytes(long long n, int W) { return (((size_t)n * (size_t)W + 15) & ~(size_t)15) + (size_t)n * (3 * 8 + 2 * 4 + 2) + 64; }
StageView stage_view(void *base, long long n, int W)
{
    StageView v{};
    unsigned char *p = static_cast<unsigned char *>(base);
    v.kmers = p; p += ((size_t)n * (size_t)W + 15) & ~(size_t)15;
    v.start = reinterpret_cast<long long *>(p); p += (size_t)n * 8;
    v.stop = reinterpret_cast<long long *>(p); p += (size_t)n * 8;
    v.freq = reinterpret_cast<long long *>(p); p += (size_t)n * 8;
    v.region = reinterpret_cast<int *>(p); p += (size_t)n * 4;
    v.walk = reinterpret_cast<int *>(p); p += (size_t)n * 4;
    v.strand = p; p += (size_t)n;
    v.is_ref = p;
    return v;
}
}  // namespace

GFM_API int gfm_graph_write_tsvs(gfm_graph_t g, const uint8_t *d_kmers, const int64_t *d_start, const int64_t *d_stop,
                                 const uint8_t *d_strand, const int64_t *d_freq, const uint8_t *d_is_ref, const int32_t *d_region,
                                 const int32_t *d_walk, int64_t n_rows, int32_t width, int32_t n_regions,
                                 const int64_t *h_region_stops, const char *const *h_labels, const char *const *h_paths,
                                 const char *chrom_name, uint32_t flags, int n_threads, uint8_t *h_seen, void *stream,
                                 gfm_tsv_write_stats_t *stats)
{
    if (!g || n_rows < 0 || n_regions < 0 || width < 1 || width > GFM_MAX_WIDTH) return gfail(GFM_ERR_INVALID, "bad argument");
    if (n_regions && (!h_region_stops || !h_labels || !h_paths || !h_seen)) return gfail(GFM_ERR_INVALID, "NULL region arrays");
    if (!chrom_name) return gfail(GFM_ERR_INVALID, "NULL chromosome name");
    if (n_rows && (!d_kmers || !d_start || !d_stop || !d_strand || !d_freq || !d_is_ref || !d_region || !d_walk))
        return gfail(GFM_ERR_INVALID, "NULL row buffer");
    static_assert(sizeof(gfm_tsv_write_stats_t) == sizeof(gfm_host::WriteStats), "stats layout of the C ABI");
    gfm_host::WriteStats ws{};
    const auto t0 = std::chrono::steady_clock::now();
    if (n_threads <= 0) n_threads = (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    gfm_host::WriteJob job{};
    job.W = width;
    job.n_regions = n_regions;
    job.region_stop = reinterpret_cast<const long long *>(h_region_stops);
    job.labels = h_labels;
    job.paths = h_paths;
    job.chrom = chrom_name;
    job.node_paths = !(flags & GFM_TSV_NO_NODEPATH);
    job.seen = h_seen;
    job.threads = n_threads;
    if (n_rows > 0) {
        const long long chunk = std::min<long long>(n_rows, kTsvChunkRows);
        const size_t need = stage_bytes(chunk, width);
        if (need > g->h_stage_cap) {
            for (int k = 0; k < 2; ++k) {
                if (g->h_stage[k]) (void)hipHostFree(g->h_stage[k]);
                g->h_stage[k] = nullptr;
            }
            g->h_stage_cap = 0;
            for (int k = 0; k < 2; ++k) GX_TRY(hipHostMalloc(&g->h_stage[k], need, hipHostMallocDefault));
            g->h_stage_cap = need;
        }
        if (!g->copy_st) GX_TRY(hipStreamCreateWithFlags(&g->copy_st, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k)
            if (!g->ev_copy[k]) GX_TRY(hipEventCreateWithFlags(&g->ev_copy[k], hipEventDisableTiming));
        GX_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));       // the rows are complete
        const long long n_chunks = (n_rows + chunk - 1) / chunk;
        auto issue = [&](long long c) -> hipError_t {
            const long long r0 = c * chunk, n = std::min(chunk, n_rows - r0);
            const StageView v = stage_view(g->h_stage[c & 1], chunk, width);
            hipError_t e = hipMemcpyAsync(v.kmers, d_kmers + (size_t)r0 * (size_t)width, (size_t)n * (size_t)width, hipMemcpyDeviceToHost, g->copy_st);
            if (e == hipSuccess) e = hipMemcpyAsync(v.start, d_start + r0, (size_t)n * 8, hipMemcpyDeviceToHost, g->copy_st);
            if (e == hipSuccess) e = hipMemcpyAsync(v.stop, d_stop + r0, (size_t)n * 8, hipMemcpyDeviceToHost, g->copy_st);
            if (e == hipSuccess) e = hipMemcpyAsync(v.freq, d_freq + r0, (size_t)n * 8, hipMemcpyDeviceToHost, g->copy_st);
            if (e == hipSuccess) e = hipMemcpyAsync(v.region, d_region + r0, (size_t)n * 4, hipMemcpyDeviceToHost, g->copy_st);
            if (e == hipSuccess) e = hipMemcpyAsync(v.walk, d_walk + r0, (size_t)n * 4, hipMemcpyDeviceToHost, g->copy_st);
            if (e == hipSuccess) e = hipMemcpyAsync(v.strand, d_strand + r0, (size_t)n, hipMemcpyDeviceToHost, g->copy_st);
            if (e == hipSuccess) e = hipMemcpyAsync(v.is_ref, d_is_ref + r0, (size_t)n, hipMemcpyDeviceToHost, g->copy_st);
            if (e == hipSuccess) e = hipEventRecord(g->ev_copy[c & 1], g->copy_st);
            return e;
        };
        GX_TRY(issue(0));
        for (long long c = 0; c < n_chunks; ++c) {
            const auto tc = std::chrono::steady_clock::now();
            GX_TRY(hipEventSynchronize(g->ev_copy[c & 1]));
            ws.copy_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count();
            if (c + 1 < n_chunks) GX_TRY(issue(c + 1));         // (its buffer's last chunk, c - 1, has been written)
            const long long r0 = c * chunk, n = std::min(chunk, n_rows - r0);
            const StageView v = stage_view(g->h_stage[c & 1], chunk, width);
            const gfm_host::RowChunk rc{v.kmers, v.start, v.stop, v.freq, v.strand, v.is_ref, v.region, v.walk, n};
            std::string err;
            const int rc_ = gfm_host::write_chunk(g->host, job, rc, ws, err);
            if (rc_) {
                (void)hipStreamSynchronize(g->copy_st);
                return gfail(rc_, err);
            }
        }
    }
    ws.total_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats) std::memcpy(stats, &ws, sizeof ws);
    return GFM_OK;
}

// ------------------------------------------------------------------------------------------- fused path
extern "C" int gfm_motif_view_(gfm_motif_t m, int max_bins, int small_bins, const int64_t **sm, int *W, int *min_val, int *L,
                               int *win_lo, int *win_nb, int *device, int *n_cu, const unsigned **d_ftab);

namespace {
// LDS histogram windows of the fused kernels, per number of motifs that share the enumeration: at most `kFusedMaxBins` bins
// (scores outside a window spill to global atomics) -- but `kFusedSmallBins` whenever that many hold 90 % of a motif's
// background mass: a window of 8 000 bins lets TWO workgroups of graph_score_kernel share a CU (synthetic W = 19..64 motifs with
// 12 000+ reachable scores: gfm_graph_score 0.15 / 0.25 / 0.36 / 0.57 ms at W = 19 / 30 / 40 / 64 with the wide window and one
// workgroup per CU, 0.105 / 0.18 / 0.25 / 0.44 ms with the small one, round 4).  Two and three motifs: windows that leave
// the workgroup its wavefronts.
constexpr int kFusedMaxBins[kMaxMM] = {16384, 8000, 5400};
#ifndef GFM_GRAPH_SMALL_BINS1         // (lab builds vary it)
#define GFM_GRAPH_SMALL_BINS1 8000
#endif
constexpr int kFusedSmallBins[kMaxMM] = {GFM_GRAPH_SMALL_BINS1, 6000, 5000};
constexpr size_t kCuLdsBytes = 160 * 1024;

// first site at or behind `target`, searched from a hint (tiles come in ascending order: a step or two)
int site_lower_bound(const std::vector<int> &pos, int hint, long long target)
{
    const int n = (int)pos.size();
    if (hint > n) hint = n;
    if (hint > 0 && (long long)pos[(size_t)hint - 1] >= target) hint = 0;        // the hint lies behind: start over
    int step = 1, lo = hint, hi = hint;
    while (hi < n && (long long)pos[(size_t)hi] < target) { lo = hi + 1; hi += step; step <<= 1; }
    if (hi > n) hi = n;
    return (int)(std::lower_bound(pos.begin() + lo, pos.begin() + hi, target, [](int a, long long b) { return (long long)a < b; }) -
                 pos.begin());
}

// One instantiation of a fused kernel: its register count (asked once) and the dynamic-LDS attribute (set once).
struct KernelInfo { const void *fn = nullptr; int vgprs = 128; bool ready = false; };
int kernel_prepare(KernelInfo &k, const void *fn)
{
    if (k.ready) return GFM_OK;
    hipFuncAttributes at{};
    GX_TRY(hipFuncGetAttributes(&at, fn));
    GX_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCuLdsBytes));
    k.fn = fn;
    k.vgprs = std::max(8, at.numRegs);
    k.ready = true;
    return GFM_OK;
}
// wavefronts per workgroup and workgroups per CU: as many wavefronts on a CU as its LDS and the kernel's registers allow
// (512 VGPRs per SIMD lane, four SIMDs; two workgroups rather than one where both give the same)
struct Shape { int waves = 8, per_cu = 1; };
Shape pick_shape(int vgprs, size_t lds_fixed, size_t lds_per_wave)
{
    const int alloc = (vgprs + 7) & ~7;
    const int by_regs = std::max(1, std::min(8, 512 / std::max(8, alloc))) * 4;      // wavefronts per CU the registers allow
    Shape best{};
    int best_waves = 0;
#ifdef GFM_LAB
    static const int force = [] { const char *e = std::getenv("GRAFIMO_FUSED_WAVES"); return e ? atoi(e) : 0; }();
#else
    constexpr int force = 0;
#endif
    for (int per_cu = 2; per_cu >= 1; --per_cu)
        for (int waves = kFusedMaxWaves; waves >= 4; waves -= 4) {     // whole rounds of the four SIMDs: a workgroup of ten wavefronts
                                                                        // loads them 3, 3, 2, 2 and two such do not fit where 2 x 10 / 4 would (82.9 against 60.8 us)
            if (force && waves != force) continue;
            if ((size_t)per_cu * (lds_fixed + (size_t)waves * lds_per_wave) > kCuLdsBytes) continue;
            if (per_cu * waves > by_regs) continue;
            if (per_cu * waves > best_waves) { best_waves = per_cu * waves; best = Shape{waves, per_cu}; }
        }
    if (best_waves == 0) best = Shape{4, 1};
    return best;
}

template <int MM> struct FusedKernels {
    static KernelInfo score[2][2], heavy, del_score;       // [listing][general]
};
template <int MM> KernelInfo FusedKernels<MM>::score[2][2];
template <int MM> KernelInfo FusedKernels<MM>::heavy;
template <int MM> KernelInfo FusedKernels<MM>::del_score;

struct FusedLaunch {
    gfm_graph *g;
    FusedPlan *P;
    FusedArgs a;
    hipStream_t st;
    int n_cu, n_motifs;
    bool listing, indels, with_hist;
    unsigned long long *heavy_ctl;
    int *overflow;                // the call's overflow word: the caller's d_overflow (zeroed by the caller with its counters), or f_flags[2]
};

// the launches of one gfm_graph_score[_multi] call for MM motifs
template <int MM> int launch_fused(FusedLaunch &L)
{
    gfm_graph *g = L.g;
    FusedPlan *P = L.P;
    FusedArgs &a = L.a;
    hipStream_t st = L.st;
    const int W = a.W, n_cu = L.n_cu;
    const size_t tab_bytes = sizeof(unsigned) * (size_t)fused_tab_dwords(MM, W), hist_bytes = sizeof(unsigned) * (size_t)a.slab_stride;
    using FK = FusedKernels<MM>;
    // ---- graph_score_kernel, twice: over the tiles that may hold insertions / deletions, then -- its lean instantiation -- over
    // the pure ones (the host sorted the table into the two ranges)
    struct Part { int begin, end, grid; Shape sh; size_t lds; } part[2];
#ifdef GFM_LAB
    static const int split = [] { const char *e = std::getenv("GRAFIMO_FUSED_SPLIT"); return e ? atoi(e) : 0; }();
#else
    constexpr int split = 0;
#endif
    for (int general = 1; general >= 0; --general) {
        KernelInfo &ks = FK::score[L.listing ? 1 : 0][general];
        const void *fn = L.listing ? (general ? reinterpret_cast<const void *>(graph_score_kernel<MM, true, true>)
                                              : reinterpret_cast<const void *>(graph_score_kernel<MM, true, false>))
                                   : (general ? reinterpret_cast<const void *>(graph_score_kernel<MM, false, true>)
                                              : reinterpret_cast<const void *>(graph_score_kernel<MM, false, false>));
        if (const int rc = kernel_prepare(ks, fn)) return rc;
        const size_t wave_bytes = L.listing ? (general ? sizeof(WaveLdsT<MM, true, true>) : sizeof(WaveLdsT<MM, true, false>))
                                            : (general ? sizeof(WaveLdsT<MM, false, true>) : sizeof(WaveLdsT<MM, false, false>));
        Part &pt = part[general];
        // One launch of the GENERAL instantiation over all tiles is the default: it classifies pure tiles with the same lean
        // code path, and the tiles that may hold insertions / deletions (a fifth of them, two to three times the cost each)
        // fill a chip badly on their own -- split: 36.5 + 25.8 us against 49.7 us for the one launch (profiles/r05_fused_ab.txt).
        // GRAFIMO_FUSED_SPLIT=1 (measurement aid) runs the two instantiations one after the other, =2 side by side on two streams.
        pt.begin = general ? 0 : (split ? P->f_n_general : P->f_n_tiles);
        pt.end = general ? (split ? P->f_n_general : P->f_n_tiles) : P->f_n_tiles;
        pt.sh = pick_shape(ks.vgprs, tab_bytes + hist_bytes + 16, wave_bytes + sizeof(long long) + sizeof(int));
        const int n_t = pt.end - pt.begin;
        pt.grid = n_t > 0 ? std::max(1, std::min((n_t + pt.sh.waves - 1) / pt.sh.waves, pt.sh.per_cu * n_cu)) : 0;
        pt.lds = tab_bytes + (size_t)pt.sh.waves * wave_bytes + sizeof(long long) * (size_t)pt.sh.waves + sizeof(int) * (size_t)(pt.sh.waves + 4) + hist_bytes;
    }
    const int g1 = part[0].grid + part[1].grid;          // slab rows of the two launches: the general one's first
    // ---- graph_heavy_kernel: a grid that fills the chip whatever the number of tiles
    if (const int rc = kernel_prepare(FK::heavy, reinterpret_cast<const void *>(graph_heavy_kernel<MM>))) return rc;
    const Shape shh = pick_shape(FK::heavy.vgprs, tab_bytes + hist_bytes + 16, sizeof(HeavyLdsT<MM>) + sizeof(long long));
    const int g_heavy = shh.per_cu * n_cu;
    const size_t lds_h = tab_bytes + (size_t)shh.waves * sizeof(HeavyLdsT<MM>) + sizeof(long long) * (size_t)shh.waves + hist_bytes;
    GX_TRY(g->f_slabs.reserve((size_t)std::max(g1, g_heavy) * (size_t)std::max(1, a.slab_stride) + 1));
    a.slabs = g->f_slabs.p;
    // the listed windows' walks (graph_del_score_kernel: one wavefront per work item of the plan).  On every call but a plan's
    // first the list, the layouts and the work items exist already and the kernel depends on nothing graph_score_kernel does:
    // it COULD run beside it on the handle's side stream.  Measured twice (GRAFIMO_FUSED_BESIDE=1, profiles/r05_fused_ab.txt):
    // with the grid that fills every CU's LDS whatever the number of items the call was 0.094 against 0.088 ms; with the grid
    // sized to the plan's items (the host learns their number on the plan's first call: at most one workgroup per CU, launched
    // behind the score kernel) 0.106 against 0.089 ms -- graph_score_kernel stretches by more than graph_del_score_kernel
    // takes alone (its share of the tiles per wavefront is fixed before it starts: whatever delays some of its wavefronts
    // delays its end).  One after the other it is; the item count still sizes the grid.
    auto launch_del_score = [&](hipStream_t on, int grid_cap, bool fill_cache) -> int {
        if (const int rc = kernel_prepare(FK::del_score, reinterpret_cast<const void *>(graph_del_score_kernel<MM>))) return rc;
        int pitch = ((W + 3) / 4) * 4;
        if ((pitch / 4) % 2 == 0) pitch += 4;         // an odd dword pitch: the lanes' slots fall on all LDS banks
        const size_t lds_b = tab_bytes + sizeof(SiteRec) * kSiteCache * kFusedDelThreads +
                             sizeof(LayoutRec) * kFusedDelThreads * kFusedLayouts + 3 * sizeof(long long) * kFusedDelThreads +
                             3 * sizeof(int) * kFusedDelThreads + sizeof(int) * (size_t)kFusedDelThreads * W +
                             (size_t)kFusedDelThreads * pitch;
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (size_t)(150 * 1024) / lds_b));
        const int grid = grid_cap > 0 ? std::min(grid_cap, per_cu * n_cu) : per_cu * n_cu;
        hipLaunchKernelGGL((graph_del_score_kernel<MM>), dim3((unsigned)grid), dim3(kFusedDelThreads), lds_b, on, g->dev, a,
                           P->f_tiles.p, P->f_del_wins.p, P->f_flags.p + 1, P->f_del_recs.p, P->f_del_items.p, P->f_flags.p + 3,
                           pitch, fill_cache ? P->f_lw_kmers.p : nullptr, fill_cache ? P->f_lw_meta.p : nullptr, P->lw_pitch);
        return GFM_OK;
    };
#ifdef GFM_LAB
    static const bool serial = [] { const char *e = std::getenv("GRAFIMO_FUSED_BESIDE"); return !(e && *e == '1'); }();   // '1' = beside
#else
    constexpr bool serial = true;
#endif
    if (!P->items_known && P->items_asked) {
        if (hipEventQuery(P->ev_items) == hipSuccess) P->items_known = true;
        else (void)hipGetLastError();          // ("not ready" is no error of this call)
    }
    const int n_items = P->items_known ? (int)(P->h_heavy_ctl[1] & 0xffffffffull) : -1;
    // The plan's cache of the listed windows' walks (LwMeta): decided once the item count is back.  This call FILLS it -- the
    // deletion kernel replays the walks as on every earlier call and stores them beside scoring them; the calls that follow
    // (stream order, or the handle's call event) hand the cache to graph_score_kernel and launch no deletion kernel at all.
    // A plan whose items hold several rounds of walks (a listed window of > 4 096 walks with the pool used up) or whose walks
    // would take more than kLwCacheBytes does without.
    bool lw_fill = false;
    if (!L.listing && L.indels && P->lw_state == 0 && P->items_known) {
        // (GRAFIMO_FUSED_WALK_CACHE_BYTES: test aid -- 0 makes every plan do without the cache, as a plan too large for it does)
        static const size_t kLwCacheBytes = [] {
            const char *e = std::getenv("GRAFIMO_FUSED_WALK_CACHE_BYTES");
            return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)256 << 20;
        }();
        const int pitch16 = ((W + 15) / 16) * 16;
        const size_t rows = (size_t)std::max(n_items, 0) * kFusedDelThreads;
        if (n_items > 0 && (int)(P->h_heavy_ctl[2] & 0xffffffffull) == 0 && rows * ((size_t)pitch16 + sizeof(LwMeta)) <= kLwCacheBytes) {
            GX_TRY(P->f_lw_kmers.reserve(rows * (size_t)pitch16 + 16));
            GX_TRY(P->f_lw_meta.reserve(rows + 1));
            GX_TRY(hipMemsetAsync(P->f_lw_meta.p, 0xff, rows * sizeof(LwMeta), st));        // tile_k = -1: no walk in the lane
            P->lw_pitch = pitch16;
            P->lw_items = n_items;
            lw_fill = true;
        } else {
            P->lw_state = -1;
        }
    }
    const bool lw_cached = !L.listing && P->lw_state == 1;
    if (lw_cached) {
        a.lw_kmers = reinterpret_cast<const uint4 *>(P->f_lw_kmers.p);
        a.lw_meta = P->f_lw_meta.p;
        a.lw_items = P->lw_items;
        a.lw_pitch = P->lw_pitch;
    }
    const bool beside = split == 2 && part[0].grid > 0 && part[1].grid > 0;
    const bool del_beside = !L.listing && L.indels && !serial && n_items > 0;
    const bool del_none = !L.listing && (n_items == 0 || lw_cached);   // the plan lists no window, or graph_score_kernel scores their walks: nothing to launch
    if (L.indels) {
        const size_t n_batches = ((size_t)P->f_general_windows + kFusedDelThreads - 1) / kFusedDelThreads;      // (listed windows live in the general tiles)
        GX_TRY(P->f_del_recs.reserve(n_batches * kFusedDelThreads + 1));
        GX_TRY(P->f_del_items.reserve(n_batches * kDelMaxItems + (size_t)kDelExtraItems + 1));
    }
    if (beside || del_beside) {
        GX_TRY(hipEventRecord(g->ev_fork, st));
        GX_TRY(hipStreamWaitEvent(g->side, g->ev_fork, 0));
    }
    // gfm_graph_profile_enable: the score kernel between two events.  With ONE launch (the default) the events ride on the
    // dispatch itself (hipExtLaunchKernelGGL: they hold the kernel's own begin and end, what rocprofv3 reports); events recorded
    // around it on the stream take in ~3 us of dispatch, 8 % of this kernel.
    const bool timed = g->prof_on && g->prof_n < gfm_graph::kProfSlots;
    const bool timed_on_dispatch = timed && part[0].grid == 0 && !beside;
    if (timed && !timed_on_dispatch) GX_TRY(hipEventRecord(g->prof_ev[2 * g->prof_n], st));
    for (int general = 1; general >= 0; --general) {
        const Part &pt = part[general];
        if (pt.grid == 0) continue;
        hipStream_t st = (beside && general) ? g->side : L.st;      // (shadows: the general tiles' launch goes to the side stream)
        FusedArgs ap = a;
        ap.slabs = a.slabs + (general ? (size_t)0 : (size_t)part[1].grid * (size_t)a.slab_stride);
        if (!general && part[1].grid > 0) ap.lw_items = 0;       // (two launches -- a lab split: the cached walks are the general one's)
        const dim3 grid((unsigned)pt.grid), block((unsigned)pt.sh.waves * 64);
#define GFM_LAUNCH_SCORE(LST, GEN)                                                                                              \
        do {                                                                                                                    \
            if (timed_on_dispatch)                                                                                              \
                hipExtLaunchKernelGGL((graph_score_kernel<MM, LST, GEN>), grid, block, pt.lds, st, g->prof_ev[2 * g->prof_n],   \
                                      g->prof_ev[2 * g->prof_n + 1], 0, g->dev, ap, P->f_tiles.p, pt.begin, pt.end,             \
                                      P->f_del_wins.p, P->f_flags.p + 1, L.overflow, P->f_heavy.p, L.heavy_ctl,                 \
                                      P->f_flags.p + 4);                                                                        \
            else                                                                                                                \
                hipLaunchKernelGGL((graph_score_kernel<MM, LST, GEN>), grid, block, pt.lds, st, g->dev, ap, P->f_tiles.p,       \
                                   pt.begin, pt.end, P->f_del_wins.p, P->f_flags.p + 1, L.overflow, P->f_heavy.p,               \
                                   L.heavy_ctl, P->f_flags.p + 4);                                                              \
        } while (0)
        if (L.listing) { if (general) GFM_LAUNCH_SCORE(true, true); else GFM_LAUNCH_SCORE(true, false); }
        else { if (general) GFM_LAUNCH_SCORE(false, true); else GFM_LAUNCH_SCORE(false, false); }
#undef GFM_LAUNCH_SCORE
    }
    GX_TRY(hipGetLastError());      // (before the event query below, whose "not ready" answer is cleared: a launch failure must not go with it)
    if (timed) {
        if (!timed_on_dispatch) GX_TRY(hipEventRecord(g->prof_ev[2 * g->prof_n + 1], st));
        ++g->prof_n;
    }
    if (del_beside && !lw_cached)   // behind the score kernel's launch: its workgroups have the CUs' LDS first
        if (const int rc = launch_del_score(g->side, std::min(n_items, n_cu), false)) return rc;
    int n_slabs = g1;
    {
        // the heavy windows: launched until the plan's count has come back and says there is none
        if (!P->heavy_known && P->heavy_asked) {
            if (hipEventQuery(P->ev_heavy) == hipSuccess) P->heavy_known = true;
            else (void)hipGetLastError();          // ("not ready" is no error of this call: it must not surface in the check below)
        }
        if (!P->heavy_known || (*P->h_heavy_ctl & 0xffffffffull) != 0ull) {
            hipLaunchKernelGGL((graph_heavy_kernel<MM>), dim3((unsigned)g_heavy), dim3((unsigned)shh.waves * 64), lds_h, st, g->dev, a,
                               P->f_tiles.p, P->f_heavy.p, L.heavy_ctl, g1);
            n_slabs = std::max(g1, g_heavy);
        }
        if (L.listing) {
            GX_TRY(hipMemcpyAsync(P->h_heavy_ctl, L.heavy_ctl, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            GX_TRY(hipEventRecord(P->ev_heavy, st));
            P->heavy_asked = true;
        }
    }
    if (L.listing && !L.indels) {
        P->f_plan_ready = true;
        P->f_plan_stream = st;
        GX_TRY(hipEventRecord(P->ev_plan, st));
    }
    if (L.indels && L.listing) {
        // a plan's first call: the listed windows' layouts are counted and cut into work items (gfm_graph_fused.hpp), behind the
        // score kernel that lists them
        const size_t lds_a = sizeof(SiteRec) * kSiteCache * kFusedDelThreads;
        hipLaunchKernelGGL(graph_del_count_kernel, dim3((unsigned)(12 * n_cu)), dim3(kFusedDelThreads), lds_a, st, g->dev, W,
                           P->f_tiles.p, P->f_del_wins.p, P->f_flags.p + 1, L.overflow, P->f_flags.p + 4, P->f_del_recs.p,
                           P->f_del_items.p, P->f_flags.p + 3, P->f_flags.p + 5, P->f_flags.p + 6);
        P->f_plan_ready = true;       // (stream order: the calls that follow on this stream find the plan complete)
        P->f_plan_stream = st;
        GX_TRY(hipEventRecord(P->ev_plan, st));
        // how many work items the plan holds: the grid of graph_del_score_kernel on the calls that follow
        GX_TRY(hipMemcpyAsync(&P->h_heavy_ctl[1], P->f_flags.p + 3, sizeof(int), hipMemcpyDeviceToHost, st));
        GX_TRY(hipMemcpyAsync(&P->h_heavy_ctl[2], P->f_flags.p + 6, sizeof(int), hipMemcpyDeviceToHost, st));
        GX_TRY(hipEventRecord(P->ev_items, st));
        P->items_asked = true;
    }
    if (L.indels && !del_beside && !del_none) {
        if (const int rc = launch_del_score(st, n_items, lw_fill)) return rc;
        if (lw_fill) P->lw_state = 1;
    }
    if (beside || del_beside) {       // join: the histogram reduction and everything the caller enqueues next see the side stream's work
        GX_TRY(hipEventRecord(g->ev_join, g->side));
        GX_TRY(hipStreamWaitEvent(st, g->ev_join, 0));
    }
    for (int m = 0; m < MM; ++m)
        if (a.hnb[m] > 0)
            hipLaunchKernelGGL(graph_hist_reduce_kernel, dim3((unsigned)((a.hnb[m] + 1 + 255) / 256), (unsigned)((n_slabs + kSlabGroup - 1) / kSlabGroup)),
                               dim3(256), 0, st, g->f_slabs.p, n_slabs, a.slab_stride, a.hoff[m], a.hlo[m], a.hnb[m], a.min_val[m], a.hist[m]);
    GX_TRY(hipGetLastError());
    return GFM_OK;
}
}  // namespace

GFM_API int gfm_graph_score_multi(gfm_graph_t g, const gfm_motif_t *motifs, int32_t n_motifs, int32_t n_regions,
                                  const int64_t *h_starts, const int64_t *h_stops, uint32_t flags, const int32_t *select_cutoffs,
                                  uint64_t *const *d_hist, void *const *d_hits, const int64_t *hit_capacity,
                                  uint64_t *const *d_hit_count, uint64_t *d_n_rows, int32_t *d_overflow, int64_t *n_windows,
                                  void *stream)
{
    if (!g || !motifs || n_motifs < 1 || n_motifs > kMaxMM || n_regions < 0 || (n_regions && (!h_starts || !h_stops)))
        return gfail(GFM_ERR_INVALID, n_motifs > kMaxMM ? "at most three motifs share one enumeration" : "bad argument");
    if (!d_hit_count || !d_n_rows || !hit_capacity || !select_cutoffs) return gfail(GFM_ERR_INVALID, "NULL argument array");
    FusedArgs a{};
    int W = 0, n_cu = 256, mdev = -1;
    bool with_hist = false;
    for (int m = 0; m < n_motifs; ++m) {
        if (!motifs[m] || !d_hit_count[m] || hit_capacity[m] < 0 || (hit_capacity[m] && (!d_hits || !d_hits[m])))
            return gfail(GFM_ERR_INVALID, "NULL motif / device buffer");
        const int64_t *sm = nullptr;
        int Wm = 0, L = 0, dev_m = 0;
        const bool hist_m = d_hist && d_hist[m];
        const int rc = gfm_motif_view_(motifs[m], kFusedMaxBins[n_motifs - 1], kFusedSmallBins[n_motifs - 1], &sm, &Wm, &a.min_val[m], &L,
                                       &a.hlo[m], &a.hnb[m], &dev_m, &n_cu, &a.tab[m]);
        if (rc) return rc;
        if (m == 0) { W = Wm; mdev = dev_m; }
        else if (Wm != W || dev_m != mdev) return gfail(GFM_ERR_INVALID, "the motifs of one call have one width and live on one device");
        if (!hist_m) a.hnb[m] = 0;
        with_hist = with_hist || hist_m;
        a.hist[m] = hist_m ? reinterpret_cast<unsigned long long *>(d_hist[m]) : nullptr;
        a.cutoff[m] = select_cutoffs[m];
        a.hits[m] = static_cast<GraphHit *>(d_hits ? d_hits[m] : nullptr);
        a.hit_cap[m] = hit_capacity[m];
        a.hit_count[m] = reinterpret_cast<unsigned long long *>(d_hit_count[m]);
        a.hoff[m] = a.slab_stride;
        if (a.hnb[m] > 0) a.slab_stride += a.hnb[m] + 1;
    }
    {
        int dev = -1;
        GX_TRY(hipGetDevice(&dev));
        if (dev != mdev) return gfail(GFM_ERR_INVALID, "the motif lives on another device than the current one");
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (const int rc = g->serialise(st)) return rc;
    // ---- the plan: looked up among the handle's (GRAFIMO scans the same BED regions motif after motif, grafimo.py:177-183;
    // the widths of a motif set alternate), built and uploaded when this (regions, width) is new
    const size_t nr = (size_t)n_regions;
    FusedPlan *P = nullptr;
    for (size_t k = 0; k < g->plans.size(); ++k) {
        FusedPlan *c = g->plans[k];
        if (c->f_width == W && c->f_starts.size() == nr &&
            (nr == 0 || (std::memcmp(c->f_starts.data(), h_starts, nr * sizeof(long long)) == 0 &&
                         std::memcmp(c->f_stops.data(), h_stops, nr * sizeof(long long)) == 0))) {
            P = c;
            g->plans.erase(g->plans.begin() + (long)k);
            g->plans.insert(g->plans.begin(), P);          // most recently used first
            break;
        }
    }
    if (!P) {
        if ((int)g->plans.size() >= gfm_graph::kMaxPlans) {
            // the least recently used plan goes; kernels of earlier calls may still read its buffers: hipFree waits for the device
            delete g->plans.back();
            g->plans.pop_back();
        }
        P = new (std::nothrow) FusedPlan();
        if (!P) return gfail(GFM_ERR_NOMEM, "out of host memory");
        g->plans.insert(g->plans.begin(), P);
        GX_TRY(P->init());
        if (g->tiles_pending) {          // the staging buffer may still be read by the last copy
            GX_TRY(hipEventSynchronize(g->ev_tiles));
            g->tiles_pending = false;
        }
        const long long tail = g->dev.n_ins > 0 ? 1 : W;
        size_t n_tiles = 0;
        for (int r = 0; r < n_regions; ++r) {
            const long long s = std::max<long long>(h_starts[r], 0), e = std::min<long long>(h_stops[r], g->dev.ref_len);
            const long long nw = std::max<long long>(0, e - tail - s + 1);
            n_tiles += (size_t)((nw + kTileWin - 1) / kTileWin);
        }
        // DelWin / HeavyWin / the hit entries pack (tile | window of the tile << kDelTileBits) into 32 bits: many short regions
        // give partly filled tiles, so the tile count is the limit that binds first
        static_assert(kDelTileBits + 6 <= 31 && kTileWin == 64, "tile index and window of the tile share one int");
        if (n_tiles >= (1ull << kDelTileBits))
            return gfail(GFM_ERR_INVALID, "more than 2^25 tiles of 64 window starts in one call (split the regions)");
        if (n_tiles > g->h_tiles_cap) {
            if (g->h_tiles) (void)hipHostFree(g->h_tiles);
            g->h_tiles = nullptr;
            g->h_tiles_cap = 0;
            GX_TRY(hipHostMalloc(reinterpret_cast<void **>(&g->h_tiles), sizeof(Tile) * (n_tiles + 1), hipHostMallocDefault));
            g->h_tiles_cap = n_tiles;
        }
        // sites that are an insertion or a deletion, counted from the left: a tile is PURE (kTilePure) when none lies under it
        if (g->h_indel_prefix.empty()) {
            g->h_indel_prefix.assign(g->host.pos.size() + 1, 0);
            for (size_t i = 0; i < g->host.pos.size(); ++i)
                g->h_indel_prefix[i + 1] = g->h_indel_prefix[i] + ((g->host.del_len[i] > 0 || g->host.ins_len[i] > 0) ? 1 : 0);
        }
        // The tiles in genome order first (a host vector), with their cost keys ...
        std::vector<Tile> built(n_tiles);
        std::vector<unsigned> key(n_tiles);
        constexpr unsigned kKeyMax = 1u << 14;
        long long w_base = 0, general_windows = 0;
        size_t ti = 0, n_general = 0;
        int hint_lo = 0, hint_hi = 0, hint_far = 0;
        for (int r = 0; r < n_regions; ++r) {
            const long long s = std::max<long long>(h_starts[r], 0), e = std::min<long long>(h_stops[r], g->dev.ref_len);
            const long long nw = std::max<long long>(0, e - tail - s + 1);
            for (long long off = 0; off < nw; off += kTileWin) {
                Tile &t = built[ti];
                const int n_win = (int)std::min<long long>(kTileWin, nw - off);
                t.p0 = s + off;
                t.limit = e;
                t.region = r;
                t.i_lo = hint_lo = site_lower_bound(g->h_pos, hint_lo, t.p0 - 1);
                t.i_hi = hint_hi = site_lower_bound(g->h_pos, std::max(hint_hi, hint_lo), t.p0 + n_win - 1 + W);
                t.w_base = (int)w_base;
                t.i_far = hint_far = site_lower_bound(g->h_pos, std::max(hint_far, hint_hi), t.p0 + n_win - 1 + W + g->max_del_len);
                const bool pure = t.i_hi - t.i_lo + 1 <= kWaveSites && g->h_indel_prefix[(size_t)t.i_hi] == g->h_indel_prefix[(size_t)t.i_lo] &&
                                  g->host.max_reach[(size_t)t.i_lo] < t.p0;
                t.n_win = n_win | (pure ? kTilePure : 0);
                // cost: the site records under the tile (each costs every window that holds it a phase-2 walk), then its windows;
                // the tiles that may hold insertions / deletions count as the dearest (they are the general code path's)
                const unsigned cost = (unsigned)std::min(t.i_hi - t.i_lo, 120) * 64u + (unsigned)n_win;
                key[ti] = pure ? cost : (kKeyMax / 2 + cost);
                if (!pure) { ++n_general; general_windows += n_win; }
                ++ti;
                w_base += n_win;
                if (w_base > 0x7fffffffll) return gfail(GFM_ERR_INVALID, "too many windows in one call (split the regions)");
            }
        }
        // ... then into the pinned staging in order of DESCENDING cost (a counting sort: the keys are small; std::sort of
        // 150 000 tiles was 25 ms of the 35 ms a plan for 50 000 regions took to build).  The persistent grid deals tiles
        // round-robin (no ticket: one global word sustains 88 atomics a microsecond), so a wavefront's share is fixed before
        // the kernel starts -- and in genome order it is luck: the SQ counters showed wavefronts busy 42 us on average in a
        // kernel of 61 us, the rest is waiting for the unluckiest one.  Longest first, every wavefront gets one tile of every
        // cost stratum, the cheap ones last.  Nothing refers to a tile's place in the table but through its index: entries,
        // listed and heavy windows carry it.
        {
            std::vector<unsigned> first(kKeyMax + 1, 0);
            for (size_t i = 0; i < n_tiles; ++i) ++first[kKeyMax - 1 - key[i]];              // bucket 0 = the dearest
            unsigned run = 0;
            for (unsigned b = 0; b <= kKeyMax; ++b) { const unsigned c = first[b]; first[b] = run; run += c; }
            for (size_t i = 0; i < n_tiles; ++i) g->h_tiles[first[kKeyMax - 1 - key[i]]++] = built[i];
        }
        P->f_n_general = (int)n_general;
        P->f_general_windows = general_windows;
        GX_TRY(P->f_tiles.reserve(n_tiles + 1));
        if (n_tiles) {
            GX_TRY(hipMemcpyAsync(P->f_tiles.p, g->h_tiles, sizeof(Tile) * n_tiles, hipMemcpyHostToDevice, st));
            GX_TRY(hipEventRecord(g->ev_tiles, st));
            g->tiles_pending = true;
        }
        P->f_starts.assign(h_starts, h_starts + nr);
        P->f_stops.assign(h_stops, h_stops + nr);
        P->f_width = W;
        P->f_n_tiles = (int)n_tiles;
        P->f_n_windows = w_base;
    }
    g->plan = P;
    if (n_windows) *n_windows = P->f_n_windows;
    if (P->f_n_tiles == 0) return GFM_OK;
    const bool indels = g->dev.n_dels > 0 || g->dev.n_ins > 0;
    GX_TRY(P->f_del_wins.reserve((size_t)P->f_general_windows + 1));      // (listed windows live in the general tiles)
    GX_TRY(P->f_flags.reserve(16));
    GX_TRY(P->f_heavy.reserve((size_t)std::min<long long>(P->f_n_windows, kHeavyCap) + 1));
    // What depends on (graph, regions, width) only is made by the first call of a tile table and kept: the list of the windows
    // that touch an insertion / deletion with their layouts and work items, and the list of the heavy windows.
    const bool listing = !P->f_plan_ready;
    if (!listing && st != P->f_plan_stream) GX_TRY(hipStreamWaitEvent(st, P->ev_plan, 0));
    if (listing) {
        GX_TRY(hipMemsetAsync(P->f_flags.p, 0, 16 * sizeof(int), st));
        P->heavy_known = P->heavy_asked = false;
        P->items_known = P->items_asked = false;
        P->lw_state = 0;
    } else if (!d_overflow) {
        GX_TRY(hipMemsetAsync(P->f_flags.p + 2, 0, sizeof(int), st));
    }
    // The call's overflow word: the caller's own when it gives one -- zeroed by the caller together with its counters (one
    // fill for all of them) and written by the kernels directly.  (Round 5 kept a word of the plan's: a 4-byte memset in front
    // of the score kernel and a 4-byte copy behind the reduction, ~5 us each of a 60 us call.)
    int *const overflow_word = d_overflow ? d_overflow : P->f_flags.p + 2;
    a.W = W;
    a.n_motifs = n_motifs;
    a.forward_only = (flags & GFM_GRAPH_FORWARD_ONLY) ? 1 : 0;
    a.n_rows = reinterpret_cast<unsigned long long *>(d_n_rows);
    a.listing = listing ? 1 : 0;
    a.plan_overflow = P->f_flags.p + 4;
#ifdef GFM_LAB      // lab builds only (scripts/lab_build.sh -DGFM_LAB): the product reads neither variable
    static const int timer_mode = [] { const char *e = std::getenv("GRAFIMO_FUSED_TIMERS"); return e ? atoi(e) : 0; }();
    const bool timers = timer_mode == 1, tile_log = timer_mode == 2;
    a.dbg = nullptr;
    a.tile_log = nullptr;
    if (timers || tile_log) {
        GX_TRY(g->f_dbg.reserve(64 + 2 * (size_t)P->f_n_tiles));
        GX_TRY(hipMemsetAsync(g->f_dbg.p, 0, (64 + 2 * (size_t)P->f_n_tiles) * sizeof(unsigned long long), st));
        if (timers) a.dbg = g->f_dbg.p;
        else a.tile_log = g->f_dbg.p + 64;
    }
    static const int lab = [] { const char *e = std::getenv("GRAFIMO_FUSED_LAB"); return e ? atoi(e) : 0; }();
    a.lab = lab;
#endif
    FusedLaunch L{g, P, a, st, n_cu, n_motifs, listing, indels, with_hist, reinterpret_cast<unsigned long long *>(P->f_flags.p + 8), overflow_word};
    int rc = GFM_OK;
    if (n_motifs == 1) rc = launch_fused<1>(L);
    else if (n_motifs == 2) rc = launch_fused<2>(L);
    else rc = launch_fused<3>(L);
    if (rc) return rc;
#ifdef GFM_LAB
    if (timers) {      // measurement aid: what the wavefronts of graph_del_score_kernel spent where (10-ns ticks)
        unsigned long long h[48];
        int fl[4];
        GX_TRY(hipStreamSynchronize(st));
        GX_TRY(hipMemcpy(h, g->f_dbg.p, sizeof h, hipMemcpyDeviceToHost));
        GX_TRY(hipMemcpy(fl, P->f_flags.p, sizeof fl, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[fused] listed windows %d, work items %d\n", fl[1], fl[3]);
        for (int k = 0; k < 16; ++k)
            if (h[32 + k]) std::fprintf(stderr, "[fused] phase %d: n %llu, mean %.2f us, max %.2f us\n", k, h[32 + k], 0.01 * (double)h[k] / (double)h[32 + k], 0.01 * (double)h[16 + k]);
    }
    if (tile_log) {
        GX_TRY(hipStreamSynchronize(st));
        if (!listing && P->f_n_tiles > 0) {       // per tile: duration and start (relative to its wavefront's first tile), with the tile's record
            std::vector<unsigned long long> tt((size_t)P->f_n_tiles), tp((size_t)P->f_n_tiles);
            std::vector<Tile> tl((size_t)P->f_n_tiles);
            GX_TRY(hipMemcpy(tt.data(), g->f_dbg.p + 64, tt.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            GX_TRY(hipMemcpy(tp.data(), g->f_dbg.p + 64 + tt.size(), tp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            GX_TRY(hipMemcpy(tl.data(), P->f_tiles.p, tl.size() * sizeof(Tile), hipMemcpyDeviceToHost));
            std::vector<int> order(tt.size());
            for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
            auto dur = [&](size_t i) { return 0.01 * (double)(tt[i] >> 40); };
            auto beg = [&](size_t i) { return 0.01 * (double)((tt[i] >> 16) & 0xffffffull); };
            auto wg = [&](size_t i) { return (size_t)(tt[i] & 0xffffull); };
            std::sort(order.begin(), order.end(), [&](int x, int y) { return (tt[(size_t)x] >> 40) > (tt[(size_t)y] >> 40); });
            std::fprintf(stderr, "[fused] tiles %zu: duration max %.2f, p99 %.2f, p90 %.2f, median %.2f, min %.2f us\n", order.size(), dur((size_t)order[0]),
                         dur((size_t)order[order.size() / 100]), dur((size_t)order[order.size() / 10]), dur((size_t)order[order.size() / 2]), dur((size_t)order.back()));
            double sum_p = 0, sum_g = 0;
            size_t n_p = 0, n_g = 0, G = 0;
            std::vector<double> ends(tt.size());
            for (size_t i = 0; i < tt.size(); ++i) {
                ends[i] = beg(i) + dur(i);
                G = std::max(G, wg(i) + 1);
                if (tl[i].n_win & kTilePure) { sum_p += dur(i); ++n_p; } else { sum_g += dur(i); ++n_g; }
            }
            std::vector<double> wg_end(G, 0.0), wg_busy(G, 0.0);
            std::vector<int> wg_tiles(G, 0);
            for (size_t i = 0; i < tt.size(); ++i) {
                wg_end[wg(i)] = std::max(wg_end[wg(i)], ends[i]);
                wg_busy[wg(i)] += dur(i);
                ++wg_tiles[wg(i)];
            }
            std::sort(ends.begin(), ends.end());
            std::fprintf(stderr, "[fused] pure tiles %zu: mean %.2f us; general tiles %zu: mean %.2f us; wave-time in tiles %.0f us\n", n_p,
                         n_p ? sum_p / (double)n_p : 0.0, n_g, n_g ? sum_g / (double)n_g : 0.0, sum_p + sum_g);
            {   // phases per kind of tile: staging -> LDS | classify | base scores | reference walks + scan | phase 2
                double ps[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
                for (size_t i = 0; i < tp.size(); ++i)
                    for (int k = 0; k < 5; ++k) ps[(tl[i].n_win & kTilePure) ? 0 : 1][k] += 0.01 * (double)((tp[i] >> (12 * k)) & 0xfffull);
                for (int kind = 0; kind < 2; ++kind) {
                    const double n = (double)(kind ? n_g : n_p);
                    if (n > 0)
                        std::fprintf(stderr, "[fused] %s tiles, mean us per phase: commit %.2f, classify %.2f, base scores %.2f, reference walks + scan %.2f, phase 2 %.2f\n",
                                     kind ? "general" : "pure", ps[kind][0] / n, ps[kind][1] / n, ps[kind][2] / n, ps[kind][3] / n, ps[kind][4] / n);
                }
            }
            std::fprintf(stderr, "[fused] tile END times: p10 %.1f, p50 %.1f, p90 %.1f, p99 %.1f, max %.1f us\n", ends[ends.size() / 10],
                         ends[ends.size() / 2], ends[ends.size() * 9 / 10], ends[ends.size() * 99 / 100], ends.back());
            {
                double e0 = 0, e1 = 0, b0 = 0, b1 = 0, t0 = 0, t1 = 0;
                for (size_t b = 0; b < G; ++b) {
                    (b < G / 2 ? e0 : e1) += wg_end[b];
                    (b < G / 2 ? b0 : b1) += wg_busy[b];
                    (b < G / 2 ? t0 : t1) += wg_tiles[b];
                }
                const double h = (double)(G / 2 ? G / 2 : 1);
                std::fprintf(stderr, "[fused] workgroups %zu: lower half / upper half of the grid: done at %.1f / %.1f us, tiles %.1f / %.1f, wavefront-time %.0f / %.0f us\n",
                             G, e0 / h, e1 / h, t0 / h, t1 / h, b0 / h, b1 / h);
                std::vector<double> we = wg_end;
                std::sort(we.begin(), we.end());
                std::fprintf(stderr, "[fused] workgroups done at: min %.1f, p10 %.1f, median %.1f, p90 %.1f, max %.1f us\n", we.front(), we[G / 10], we[G / 2],
                             we[G * 9 / 10], we.back());
            }
            for (int k = 0; k < 8 && k < (int)order.size(); ++k) {
                const size_t i = (size_t)order[(size_t)k];
                const Tile &t = tl[i];
                std::fprintf(stderr, "[fused]   tile %zu: %.2f us (began at %.2f, workgroup %zu), windows %d%s, sites %d (far %d)\n", i, dur(i), beg(i), wg(i),
                             t.n_win & 0xff, (t.n_win & kTilePure) ? " pure" : "", t.i_hi - t.i_lo, t.i_far - t.i_lo);
            }
        }
    }
#endif
    return g->called(st);
}

GFM_API int gfm_graph_profile_enable(gfm_graph_t g, int on)
{
    if (!g) return gfail(GFM_ERR_INVALID, "graph is NULL");
    if (on)
        for (hipEvent_t &e : g->prof_ev)
            if (!e) GX_TRY(hipEventCreate(&e));
    g->prof_on = on ? 1 : 0;
    g->prof_n = 0;
    return GFM_OK;
}

GFM_API int gfm_graph_profile_read(gfm_graph_t g, float *h_ms_out, int capacity, int *n_out)
{
    if (!g || !h_ms_out || !n_out || capacity < 0) return gfail(GFM_ERR_INVALID, "bad argument");
    const int n = std::min(capacity, g->prof_n);
    for (int k = 0; k < n; ++k) {
        GX_TRY(hipEventSynchronize(g->prof_ev[2 * k + 1]));
        GX_TRY(hipEventElapsedTime(&h_ms_out[k], g->prof_ev[2 * k], g->prof_ev[2 * k + 1]));
    }
    *n_out = n;
    g->prof_n = 0;
    return GFM_OK;
}

GFM_API int gfm_graph_score(gfm_graph_t g, gfm_motif_t m, int32_t n_regions, const int64_t *h_starts, const int64_t *h_stops,
                            uint32_t flags, int32_t select_cutoff, uint64_t *d_hist, void *d_hits, int64_t hit_capacity,
                            uint64_t *d_hit_count, uint64_t *d_n_rows, int32_t *d_overflow, int64_t *n_windows, void *stream)
{
    if (!m) return gfail(GFM_ERR_INVALID, "bad argument");
    return gfm_graph_score_multi(g, &m, 1, n_regions, h_starts, h_stops, flags, &select_cutoff, &d_hist, &d_hits, &hit_capacity,
                                 &d_hit_count, d_n_rows, d_overflow, n_windows, stream);
}

GFM_API int gfm_graph_annotate(gfm_graph_t g, const void *d_hits, const uint64_t *d_hit_count, int64_t hit_capacity,
                               const int32_t *d_cutoff, const double *d_qtable, void *d_records, void *stream)
{
    if (!g) return gfail(GFM_ERR_INVALID, "graph is NULL");
    FusedPlan *P = g->plan;
    if (hit_capacity <= 0 || !P || P->f_n_tiles == 0) return GFM_OK;
    if (!d_hits || !d_hit_count || !d_records) return gfail(GFM_ERR_INVALID, "NULL device buffer");
    static_assert(sizeof(HitRec) == sizeof(gfm_graph_hit_t) && sizeof(GraphHit) == 16, "record layouts of the C ABI");
    if (hit_capacity > 0x7fffffffll) return gfail(GFM_ERR_INVALID, "hit capacity beyond 2^31");
    const unsigned blocks = (unsigned)std::min<int64_t>(hit_capacity, 8192);   // a wavefront per entry, entries dealt over the grid
    if (const int rc = g->serialise(static_cast<hipStream_t>(stream))) return rc;
    hipLaunchKernelGGL(graph_annotate_kernel, dim3(blocks), dim3(64), 0, static_cast<hipStream_t>(stream), g->dev,
                       g->d_allele_count, P->f_width, P->f_tiles.p, P->f_n_tiles, static_cast<const GraphHit *>(d_hits),
                       reinterpret_cast<const unsigned long long *>(d_hit_count), (long long)hit_capacity, d_cutoff, d_qtable,
                       static_cast<HitRec *>(d_records));
    GX_TRY(hipGetLastError());
    return g->called(static_cast<hipStream_t>(stream));
}
