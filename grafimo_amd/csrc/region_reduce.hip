// region_reduce.hip -- the two reductions BASELINE.json's north_star names and the reference does not have:
//   * per REGION its best (score, row)                 -> gfm_region_best
//   * per LOCUS (region, {start, stop}) the best score of any row on either strand ("forward/reverse-strand max")
//                                                        -> gfm_locus_max
// The reference keeps every strand as a row of its own (score_sequences.py:279-321) and reports rows, not maxima; its
// one consumer of "the best hit of a region" is --top-graphs: the first N distinct sequence_names of the table sorted
// by p-value (res_writer.py:153-157), i.e. the regions ranked by their best hit.  Nothing here changes the reported
// rows; these are extra outputs, and at N > 1 GPUs they let rank 0 receive n_regions entries per rank
// instead of every hit when only the top regions are asked for (grafimo_amd/top_hits.py).
//
// Both kernels are byte/integer work on 4-24 bytes per row: HBM/L2-bound, no MFMA.  The wave-level part: a row's key
// is combined with its neighbours' inside the 64-lane wavefront by shuffles (DPP) before anything touches memory --
// a segmented max-scan over the lanes' region ids for the region best (rows of a region are contiguous in every input
// of this library: TSV files, extraction rows, synthetic batches -- one atomic per wave and region instead of one per
// row), the lane ^ 1 exchange for the two strands of a walk that sit in neighbouring rows.

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "grafimo_hip.h"

#define GFM_API extern "C" __attribute__((visibility("default")))
extern "C" void gfm_set_error_(const char *msg);   // thread-local slot of grafimo_hip.hip

namespace {

int rfail(int code, const std::string &msg)
{
    gfm_set_error_(msg.c_str());
    return code;
}

#define RR_TRY(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return rfail(GFM_ERR_HIP, std::string(#expr " failed: ") + hipGetErrorString(e_));    \
    } while (0)

constexpr int kThreads = 256;
constexpr unsigned long long kRowMask = (1ull << GFM_BEST_ROW_BITS) - 1ull;

__device__ __forceinline__ long long shfl_up_ll(long long v, int d)
{
    const int lo = __shfl_up((int)(v & 0xffffffffll), d), hi = __shfl_up((int)(v >> 32), d);
    return ((long long)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ long long shfl_xor_ll(long long v, int d)
{
    const int lo = __shfl_xor((int)(v & 0xffffffffll), d), hi = __shfl_xor((int)(v >> 32), d);
    return ((long long)hi << 32) | (unsigned)lo;
}

// the rows a reduction looks at: score >= cutoff, a region id inside the table, and -- when the caller passes the
// haplotype counts -- carried by a haplotype (what ResultTmp.to_df keeps without --recomb, resultsTmp.py:309-310)
struct RowFilter {
    const int *scores;
    const int *region;
    const long long *freq;
    long long n;
    int n_regions, cutoff;
    __device__ bool take(long long i, int &s, int &r) const
    {
        if (i >= n) return false;
        s = scores[i];
        r = region[i];
        if (s < cutoff || (unsigned)r >= (unsigned)n_regions) return false;
        return !freq || freq[i] > 0;
    }
};

// Wave per 64 consecutive rows.  key = score << 44 | (2^44 - 1 - row): the maximum is the best score, the lowest row
// among equals.  Inclusive max-scan over the lanes, segmented by region id: six shuffle steps, after which the last
// lane of every run of equal ids holds the run's maximum and is the only one that goes to memory.
__global__ void __launch_bounds__(kThreads)
region_best_kernel(RowFilter f, const int *__restrict__ d_cutoff, long long row_base,
                   unsigned long long *__restrict__ best)
{
    if (d_cutoff) f.cutoff = max(f.cutoff, *d_cutoff);
    const int lane = threadIdx.x & 63;
    const long long waves = ((long long)gridDim.x * kThreads) >> 6;
    const long long strips = (f.n + 63) >> 6;
    for (long long strip = ((long long)blockIdx.x * kThreads + threadIdx.x) >> 6; strip < strips; strip += waves) {
        const long long i = (strip << 6) + lane;
        int s = 0, r = -1;
        const bool ok = f.take(i, s, r);
        if (!ok) r = -1;
        long long key = ok ? (long long)(((unsigned long long)(unsigned)s << GFM_BEST_ROW_BITS) |
                                         (kRowMask - (unsigned long long)(row_base + i))) : 0ll;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long k2 = shfl_up_ll(key, d);
            const int r2 = __shfl_up(r, d);
            if (lane >= d && r2 == r && k2 > key) key = k2;
        }
        const int r_next = __shfl_down(r, 1);
        if (r >= 0 && (lane == 63 || r_next != r))
            atomicMax(&best[r], (unsigned long long)key);
    }
}

// thread block per region: region ids of the rows [off[r], off[r + 1])
__global__ void __launch_bounds__(kThreads)
region_ids_kernel(const long long *__restrict__ off, int n_regions, long long n, int *__restrict__ region)
{
    for (int r = blockIdx.x; r < n_regions; r += gridDim.x) {
        const long long a = max(0ll, off[r]), b = min(n, off[r + 1]);
        for (long long i = a + threadIdx.x; i < b; i += kThreads) region[i] = r;
    }
}

// ---- locus max.  A locus is (region, min(start, stop), max(start, stop)): the reference span a k-mer occupies, whatever
// its strand ('-' rows carry start > stop, score_sequences.py:288-291).  Open-addressing table in global memory whose
// slots hold a REPRESENTATIVE ROW (64-bit CAS from -1), so keys are compared exactly -- through the rows' own columns --
// although a slot is one word; the value array beside it takes atomicMax of the scores.
__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

__global__ void __launch_bounds__(kThreads)
locus_insert_kernel(RowFilter f, const int *__restrict__ d_cutoff, const long long *__restrict__ start,
                    const long long *__restrict__ stop, long long *__restrict__ rep, int *__restrict__ val,
                    unsigned long long cap_mask, int *__restrict__ slot_out, unsigned long long *__restrict__ failed)
{
    if (d_cutoff) f.cutoff = max(f.cutoff, *d_cutoff);
    const int lane = threadIdx.x & 63;
    const long long waves = ((long long)gridDim.x * kThreads) >> 6;
    const long long strips = (f.n + 63) >> 6;
    for (long long strip = ((long long)blockIdx.x * kThreads + threadIdx.x) >> 6; strip < strips; strip += waves) {
        const long long i = (strip << 6) + lane;
        int s = 0, r = -1;
        const bool ok = f.take(i, s, r);
        long long lo = 0, hi = 0;
        if (ok) {
            const long long a = start[i], b = stop[i];
            lo = a < b ? a : b;
            hi = a < b ? b : a;
        } else {
            r = -1 - lane;               // never equal to a neighbour's
        }
        // the two strands of one walk in neighbouring rows (the extraction's layout: rows 2t, 2t + 1): one exchange
        // with lane ^ 1 gives their maximum, and only the even lane goes to the table
        const int ps = __shfl_xor(s, 1), pr = __shfl_xor(r, 1);
        const long long plo = shfl_xor_ll(lo, 1), phi = shfl_xor_ll(hi, 1);
        const bool paired = ok && pr == r && plo == lo && phi == hi;
        const bool leader = ok && (!paired || (lane & 1) == 0);
        const int s_ins = paired && ps > s ? ps : s;
        int h_found = -1;
        if (leader) {
            unsigned long long h = mix64(((unsigned long long)(unsigned)r * 0x9e3779b97f4a7c15ull) ^
                                         mix64((unsigned long long)lo) ^ ((unsigned long long)hi * 0xc2b2ae3d27d4eb4full)) & cap_mask;
            for (unsigned long long probes = 0; probes <= cap_mask; ++probes, h = (h + 1) & cap_mask) {
                long long cur = rep[h];
                if (cur == -1) {
                    cur = (long long)atomicCAS(reinterpret_cast<unsigned long long *>(&rep[h]), ~0ull, (unsigned long long)i);
                    if (cur == -1) { h_found = (int)h; break; }
                }
                if (f.region[cur] == r) {
                    const long long a = start[cur], b = stop[cur];
                    if ((a < b ? a : b) == lo && (a < b ? b : a) == hi) { h_found = (int)h; break; }
                }
            }
            if (h_found >= 0) atomicMax(&val[h_found], s_ins);
            else atomicAdd(failed, 1ull);
        }
        const int h_mate = __shfl_xor(h_found, 1);
        if (ok && !leader) h_found = h_mate;
        if (i < f.n) slot_out[i] = ok ? h_found : -1;
    }
}

__global__ void __launch_bounds__(kThreads)
locus_read_kernel(long long n, const int *__restrict__ val, int *__restrict__ slot_then_max)
{
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const int h = slot_then_max[i];
        slot_then_max[i] = h >= 0 ? val[h] : -1;
    }
}

int grid_for(long long n)
{
    const long long want = (n + kThreads - 1) / kThreads;
    return (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
}

}  // namespace

GFM_API int gfm_region_ids(const int64_t *d_region_off, int32_t n_regions, int64_t n, int32_t *d_region_out, void *stream)
{
    if (n_regions < 0 || n < 0) return rfail(GFM_ERR_INVALID, "negative count");
    if (n_regions == 0 || n == 0) return GFM_OK;
    if (!d_region_off || !d_region_out) return rfail(GFM_ERR_INVALID, "NULL device buffer");
    hipLaunchKernelGGL(region_ids_kernel, dim3((unsigned)(n_regions < 65535 ? n_regions : 65535)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), reinterpret_cast<const long long *>(d_region_off), n_regions,
                       (long long)n, d_region_out);
    RR_TRY(hipGetLastError());
    return GFM_OK;
}

GFM_API int gfm_region_best(const int32_t *d_scores, int64_t n, const int32_t *d_region, int32_t n_regions,
                            const int64_t *d_freq, int32_t min_score, const int32_t *d_cutoff, int64_t row_base,
                            uint64_t *d_best, void *stream)
{
    if (n < 0 || n_regions < 0 || row_base < 0) return rfail(GFM_ERR_INVALID, "negative count");
    if ((unsigned long long)(row_base + n) >= kRowMask) return rfail(GFM_ERR_INVALID, "row ids beyond 2^44");
    if (n == 0 || n_regions == 0) return GFM_OK;
    if (!d_scores || !d_region || !d_best) return rfail(GFM_ERR_INVALID, "NULL device buffer");
    const RowFilter f{d_scores, d_region, reinterpret_cast<const long long *>(d_freq), (long long)n, n_regions,
                      min_score < 0 ? 0 : min_score};
    hipLaunchKernelGGL(region_best_kernel, dim3((unsigned)grid_for(n)), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       f, d_cutoff, (long long)row_base, reinterpret_cast<unsigned long long *>(d_best));
    RR_TRY(hipGetLastError());
    return GFM_OK;
}

GFM_API int64_t gfm_locus_max_workspace(int64_t n_rows)
{
    if (n_rows < 0) return 0;
    unsigned long long cap = 1024;
    while (cap < 2ull * (unsigned long long)n_rows) cap <<= 1;
    return (int64_t)(cap * (sizeof(long long) + sizeof(int)) + 64);
}

GFM_API int gfm_locus_max(const int32_t *d_scores, int64_t n, const int32_t *d_region, int32_t n_regions,
                          const int64_t *d_start, const int64_t *d_stop, const int64_t *d_freq, int32_t min_score,
                          const int32_t *d_cutoff, void *d_work, int64_t work_bytes, int32_t *d_locus_max, void *stream)
{
    if (n < 0 || n_regions < 0) return rfail(GFM_ERR_INVALID, "negative count");
    if (n == 0) return GFM_OK;
    if (!d_scores || !d_region || !d_start || !d_stop || !d_work || !d_locus_max)
        return rfail(GFM_ERR_INVALID, "NULL device buffer");
    if ((reinterpret_cast<uintptr_t>(d_work) & 7u) != 0) return rfail(GFM_ERR_INVALID, "d_work must be 8-byte aligned");
    // the largest power-of-two table the workspace holds: [failure counter | rep int64[cap] | val int32[cap]]
    unsigned long long cap = 1024;
    if (work_bytes < (int64_t)(cap * 12 + 64)) return rfail(GFM_ERR_INVALID, "workspace too small (gfm_locus_max_workspace)");
    while ((cap << 1) * 12 + 64 <= (unsigned long long)work_bytes && (cap << 1) <= (1ull << 30)) cap <<= 1;
    if (cap < (unsigned long long)n + (unsigned long long)n / 4 && cap < (1ull << 30))
        // (every row may be a locus of its own; the probe loop gives up on a full table and counts it, but a table
        // that can fill is a workspace that was not sized with gfm_locus_max_workspace)
        return rfail(GFM_ERR_INVALID, "workspace too small for this many rows (gfm_locus_max_workspace)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned long long *failed = static_cast<unsigned long long *>(d_work);
    long long *rep = reinterpret_cast<long long *>(static_cast<unsigned char *>(d_work) + 64);
    int *val = reinterpret_cast<int *>(rep + cap);
    RR_TRY(hipMemsetAsync(failed, 0, 64, st));
    RR_TRY(hipMemsetAsync(rep, 0xff, cap * 12, st));
    const RowFilter f{d_scores, d_region, reinterpret_cast<const long long *>(d_freq), (long long)n, n_regions,
                      min_score < 0 ? 0 : min_score};
    hipLaunchKernelGGL(locus_insert_kernel, dim3((unsigned)grid_for(n)), dim3(kThreads), 0, st, f, d_cutoff,
                       reinterpret_cast<const long long *>(d_start), reinterpret_cast<const long long *>(d_stop), rep, val,
                       cap - 1, d_locus_max, failed);
    hipLaunchKernelGGL(locus_read_kernel, dim3((unsigned)grid_for(n)), dim3(kThreads), 0, st, (long long)n, val, d_locus_max);
    RR_TRY(hipGetLastError());
    return GFM_OK;
}
