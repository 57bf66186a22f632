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
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <new>
#include <string>
#include <vector>

#include "grafimo_hip.h"

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

constexpr int kWave = 64;
constexpr int kMaxAlts = 3;
constexpr int kCountThreads = 256;
constexpr int kEmitThreads = 256;                    // 4 waves, one window per wave at a time
constexpr long long kMaxWalksPerWindow = 1ll << 20;  // refuse pathological windows (2^20 walks)

struct GraphDev {
    const uint8_t *ref;         // [ref_len]
    long long ref_len;
    int n_sites;
    const int *pos;             // [n_sites] ascending, 0-based
    const uint8_t *n_alts;      // [n_sites] 1..3
    const uint8_t *alt_bases;   // [n_sites][3]
    const unsigned long long *alt_bits;   // [n_sites][3][hw] or nullptr
    int n_hap, hw;
};

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

// one thread per window: first site inside it and the number of walks (product of allele counts)
__global__ void __launch_bounds__(kCountThreads)
graph_count_kernel(GraphDev g, int n_regions, const long long *__restrict__ region_off,
                   const long long *__restrict__ first_start, int W, long long n_windows,
                   int *__restrict__ first_site, long long *__restrict__ n_walks, int *__restrict__ win_region,
                   long long *__restrict__ win_start, int *__restrict__ overflow)
{
    const long long w = (long long)blockIdx.x * kCountThreads + threadIdx.x;
    if (w >= n_windows) return;
    const int r = region_of(region_off, n_regions, w);
    const long long p = first_start[r] + (w - region_off[r]);
    const int i0 = lower_bound_pos(g.pos, g.n_sites, p);
    long long walks = 1;
    for (int i = i0; i < g.n_sites && g.pos[i] < p + W; ++i) {
        walks *= 1 + g.n_alts[i];
        if (walks > kMaxWalksPerWindow) { walks = 0; atomicMax(overflow, 1); break; }
    }
    first_site[w] = i0;
    n_walks[w] = walks;
    win_region[w] = r;      // the emit kernel reads these instead of repeating the search: its waves
    win_start[w] = p;       // are latency-bound, and a binary search is a chain of dependent loads
}

__device__ inline uint8_t complement(uint8_t c)
{
    switch (c) {
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
    case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
    default: return c;   // N stays N
    }
}

// One wave per window, walks of the window one after the other.  Lane k < n_sites_in_window owns
// site k (its allele of the current walk, last site varying fastest: itertools.product order);
// lane j < W owns base j of the k-mer; lanes own words of the haplotype bitsets.
__global__ void __launch_bounds__(kEmitThreads)
graph_emit_kernel(GraphDev g, const int *__restrict__ win_region, const long long *__restrict__ win_start,
                  int W, long long n_windows, const int *__restrict__ first_site, const long long *__restrict__ walk_base,
                  uint8_t *__restrict__ kmers, long long *__restrict__ start, long long *__restrict__ stop,
                  uint8_t *__restrict__ strand, long long *__restrict__ freq, uint8_t *__restrict__ is_ref,
                  int *__restrict__ region, int *__restrict__ walk)
{
    __shared__ uint8_t kbuf[kEmitThreads / kWave][kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = threadIdx.x >> 6;
    const long long stride = (long long)gridDim.x * (kEmitThreads / kWave);
    for (long long w = (long long)blockIdx.x * (kEmitThreads / kWave) + wv; w < n_windows; w += stride) {
        const int r = win_region[w];
        const long long p = win_start[w];
        const int i0 = first_site[w];
        const long long base = walk_base[w];
        const int walks = (int)(walk_base[w + 1] - base);
        // sites of the window: lane k owns site i0 + k (at most W <= 64 of them)
        int my_pos = -1, my_nall = 1;
        if (i0 + lane < g.n_sites && g.pos[i0 + lane] < p + W && lane < W) {
            my_pos = g.pos[i0 + lane];
            my_nall = 1 + g.n_alts[i0 + lane];
        }
        const int ns = __popcll(__builtin_amdgcn_ballot_w64(my_pos >= 0));
        // suffix products of the allele counts: stride of site k in the mixed-radix walk index
        int my_stride = 1;
        for (int k = ns - 1; k >= 0; --k) {
            const int nk = __shfl(my_nall, k);
            if (lane < k) my_stride *= nk;
        }
        const uint8_t ref_base = lane < W ? g.ref[p + lane] : (uint8_t)0;
        for (int q = 0; q < walks; ++q) {
            const int my_allele = my_pos >= 0 ? (q / my_stride) % my_nall : 0;
            // k-mer bytes: reference window, then the alternate bases of this walk
            kbuf[wv][lane] = ref_base;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (my_allele > 0)
                kbuf[wv][my_pos - p] = g.alt_bases[(size_t)(i0 + lane) * kMaxAlts + (my_allele - 1)];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const long long row = 2 * (base + q);
            if (lane < W) {
                kmers[row * W + lane] = kbuf[wv][lane];
                kmers[(row + 1) * W + lane] = complement(kbuf[wv][W - 1 - lane]);
            }
            // haplotypes that carry every allele of the walk
            long long count = ns == 0 ? g.n_hap : 0;     // no site in the window: every haplotype carries it
            if (g.alt_bits && ns > 0) {
                for (int w0 = 0; w0 < g.hw; w0 += kWave) {     // every lane takes part in the shuffles
                    const int word = w0 + lane;
                    const bool live = word < g.hw;
                    unsigned long long acc = ~0ull;
                    if (word == g.hw - 1 && (g.n_hap & 63)) acc = (1ull << (g.n_hap & 63)) - 1ull;
                    for (int k = 0; k < ns; ++k) {
                        const int a = __shfl(my_allele, k);
                        const int na = __shfl(my_nall, k) - 1;
                        if (!live) continue;
                        const unsigned long long *b = g.alt_bits + ((size_t)(i0 + k) * kMaxAlts) * g.hw + word;
                        unsigned long long bits;
                        if (a > 0) {
                            bits = b[(size_t)(a - 1) * g.hw];
                        } else {
                            bits = b[0];
                            if (na > 1) bits |= b[(size_t)g.hw];
                            if (na > 2) bits |= b[(size_t)2 * g.hw];
                            bits = ~bits;
                        }
                        acc &= bits;
                    }
                    if (live) count += __popcll(acc);
                }
                for (int d = 32; d > 0; d >>= 1) count += __shfl_xor(count, d);
            }
            const bool any_alt = __builtin_amdgcn_ballot_w64(my_allele > 0) != 0ull;
            if (lane < 2) {      // lane 0: forward row, lane 1: its reverse complement
                start[row + lane] = lane ? p + W : p;
                stop[row + lane] = lane ? p : p + W;
                strand[row + lane] = lane ? '-' : '+';
                freq[row + lane] = count;
                is_ref[row + lane] = any_alt ? 0 : 1;
                region[row + lane] = r;
                walk[row + lane] = q;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // kbuf is rewritten by the next walk
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <typename T> hipError_t upload(T **dst, const T *src, size_t count)
{
    *dst = nullptr;
    if (count == 0) return hipSuccess;
    hipError_t e = hipMalloc(dst, sizeof(T) * count);
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, src, sizeof(T) * count, hipMemcpyHostToDevice);
}

}  // namespace

struct gfm_graph {
    GraphDev dev{};
    uint8_t *d_ref = nullptr;
    int *d_pos = nullptr;
    uint8_t *d_n_alts = nullptr, *d_alt_bases = nullptr;
    unsigned long long *d_alt_bits = nullptr;
    // last plan
    int n_regions = 0, width = 0;
    long long n_windows = 0, n_walks = 0;
    long long *d_region_off = nullptr, *d_first_start = nullptr, *d_walk_base = nullptr;
    int *d_first_site = nullptr, *d_win_region = nullptr;
    long long *d_win_start = nullptr;
    void drop_plan()
    {
        (void)hipFree(d_region_off); (void)hipFree(d_first_start); (void)hipFree(d_walk_base);
        (void)hipFree(d_first_site); (void)hipFree(d_win_region); (void)hipFree(d_win_start);
        d_region_off = d_first_start = d_walk_base = d_win_start = nullptr;
        d_first_site = d_win_region = nullptr;
        n_regions = 0; n_windows = n_walks = 0;
    }
};

GFM_API int gfm_graph_create(const uint8_t *h_ref, int64_t ref_len, int32_t n_sites, const int32_t *h_pos,
                             const uint8_t *h_n_alts, const uint8_t *h_alt_bases,
                             const uint64_t *h_alt_bits, int32_t n_haplotypes, gfm_graph_t *out)
{
    if (!out) return gfail(GFM_ERR_INVALID, "NULL output handle");
    *out = nullptr;
    if (!h_ref || ref_len <= 0 || n_sites < 0 || n_haplotypes < 0)
        return gfail(GFM_ERR_INVALID, "bad reference / site count");
    if (n_sites && (!h_pos || !h_n_alts || !h_alt_bases)) return gfail(GFM_ERR_INVALID, "NULL site arrays");
    for (int i = 0; i < n_sites; ++i) {
        if (h_pos[i] < 0 || h_pos[i] >= ref_len || (i && h_pos[i] <= h_pos[i - 1]))
            return gfail(GFM_ERR_INVALID, "site positions must be strictly ascending inside the reference (site " +
                                              std::to_string(i) + ")");
        if (h_n_alts[i] < 1 || h_n_alts[i] > kMaxAlts)
            return gfail(GFM_ERR_INVALID, "a site needs 1..3 alternate alleles (site " + std::to_string(i) + ")");
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
    hipError_t e = upload(&g->d_ref, h_ref, (size_t)ref_len);
    if (e == hipSuccess) e = upload(&g->d_pos, h_pos, (size_t)n_sites);
    if (e == hipSuccess) e = upload(&g->d_n_alts, h_n_alts, (size_t)n_sites);
    if (e == hipSuccess) e = upload(&g->d_alt_bases, h_alt_bases, (size_t)n_sites * kMaxAlts);
    if (e == hipSuccess && bits)
        e = upload(&g->d_alt_bits, reinterpret_cast<const unsigned long long *>(h_alt_bits),
                   (size_t)n_sites * kMaxAlts * hw);
    if (e != hipSuccess) {
        gfm_graph_destroy(g);
        return gfail(GFM_ERR_HIP, std::string("graph upload failed: ") + hipGetErrorString(e));
    }
    g->dev = GraphDev{g->d_ref, (long long)ref_len, n_sites, g->d_pos, g->d_n_alts, g->d_alt_bases,
                      bits ? g->d_alt_bits : nullptr, bits ? n_haplotypes : 0, bits ? hw : 0};
    *out = g;
    return GFM_OK;
}

GFM_API void gfm_graph_destroy(gfm_graph_t g)
{
    if (!g) return;
    g->drop_plan();
    (void)hipFree(g->d_ref); (void)hipFree(g->d_pos); (void)hipFree(g->d_n_alts);
    (void)hipFree(g->d_alt_bases); (void)hipFree(g->d_alt_bits);
    delete g;
}

GFM_API int gfm_graph_plan(gfm_graph_t g, int32_t n_regions, const int64_t *h_starts, const int64_t *h_stops,
                           int32_t width, int64_t *n_windows, int64_t *n_rows)
{
    if (!g || n_regions < 0 || (n_regions && (!h_starts || !h_stops)))
        return gfail(GFM_ERR_INVALID, "bad argument");
    if (width < 1 || width > GFM_MAX_WIDTH) return gfail(GFM_ERR_INVALID, "width outside [1, 64]");
    g->drop_plan();
    g->width = width;
    g->n_regions = n_regions;
    // windows of region r: starts p in [max(S,0), min(E, ref_len) - W]  (vg find -p S-E -K W, pinned by
    // expected_seqs.tsv: x:0-20, W=19 -> p in {0, 1})
    std::vector<long long> off(n_regions + 1, 0), first(n_regions, 0);
    for (int r = 0; r < n_regions; ++r) {
        const long long s = std::max<long long>(h_starts[r], 0);
        const long long e = std::min<long long>(h_stops[r], g->dev.ref_len);
        first[r] = s;
        off[r + 1] = off[r] + std::max<long long>(0, e - width - s + 1);
    }
    g->n_windows = off[n_regions];
    if (n_windows) *n_windows = g->n_windows;
    if (n_rows) *n_rows = 0;
    if (g->n_windows == 0) return GFM_OK;
    if (g->n_windows > 0x7fffffffll) return gfail(GFM_ERR_INVALID, "too many windows in one plan (split the regions)");
    GX_TRY(upload(&g->d_region_off, off.data(), off.size()));
    GX_TRY(upload(&g->d_first_start, first.data(), first.size()));
    // walks per window -> exclusive prefix (row base of every window), all on the device: only the
    // total and the overflow flag come back
    struct Scratch {
        long long *walks = nullptr;
        int *flag = nullptr;
        void *tmp = nullptr;
        ~Scratch() { (void)hipFree(walks); (void)hipFree(flag); (void)hipFree(tmp); }
    } sc;
    const size_t nw = (size_t)g->n_windows;
    GX_TRY(hipMalloc(&g->d_first_site, sizeof(int) * nw));
    GX_TRY(hipMalloc(&sc.walks, sizeof(long long) * nw));
    GX_TRY(hipMalloc(&sc.flag, sizeof(int)));
    GX_TRY(hipMemsetAsync(sc.flag, 0, sizeof(int), nullptr));
    GX_TRY(hipMalloc(&g->d_win_region, sizeof(int) * nw));
    GX_TRY(hipMalloc(&g->d_win_start, sizeof(long long) * nw));
    GX_TRY(hipMalloc(&g->d_walk_base, sizeof(long long) * (nw + 1)));
    GX_TRY(hipMemsetAsync(g->d_walk_base, 0, sizeof(long long), nullptr));
    const unsigned blocks = (unsigned)((g->n_windows + kCountThreads - 1) / kCountThreads);
    hipLaunchKernelGGL(graph_count_kernel, dim3(blocks), dim3(kCountThreads), 0, nullptr, g->dev, n_regions,
                       g->d_region_off, g->d_first_start, width, g->n_windows, g->d_first_site, sc.walks,
                       g->d_win_region, g->d_win_start, sc.flag);
    GX_TRY(hipGetLastError());
    size_t tmp_bytes = 0;
    GX_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, sc.walks, g->d_walk_base + 1, (int)nw, nullptr));
    GX_TRY(hipMalloc(&sc.tmp, tmp_bytes));
    GX_TRY(hipcub::DeviceScan::InclusiveSum(sc.tmp, tmp_bytes, sc.walks, g->d_walk_base + 1, (int)nw, nullptr));
    long long total = 0;
    int overflow = 0;
    GX_TRY(hipMemcpy(&total, g->d_walk_base + nw, sizeof total, hipMemcpyDeviceToHost));
    GX_TRY(hipMemcpy(&overflow, sc.flag, sizeof overflow, hipMemcpyDeviceToHost));
    if (overflow) return gfail(GFM_ERR_OVERFLOW, "a window holds more than 2^20 walks through its sites");
    g->n_walks = total;
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
    const long long waves = g->n_windows;
    const unsigned blocks = (unsigned)std::min<long long>((waves + 3) / 4, 256 * 32);
    hipLaunchKernelGGL(graph_emit_kernel, dim3(blocks), dim3(kEmitThreads), 0, static_cast<hipStream_t>(stream),
                       g->dev, g->d_win_region, g->d_win_start, g->width, g->n_windows,
                       g->d_first_site, g->d_walk_base, d_kmers, reinterpret_cast<long long *>(d_start),
                       reinterpret_cast<long long *>(d_stop), d_strand, reinterpret_cast<long long *>(d_freq),
                       d_is_ref, d_region, d_walk);
    GX_TRY(hipGetLastError());
    return GFM_OK;
}
