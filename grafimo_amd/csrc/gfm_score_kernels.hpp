// gfm_score_kernels.hpp -- what the score kernel (gfm_score_quad.hpp) shares with its host side: launch geometry,
// argument blocks, the hit-queue plumbing; post_kernel and select_hits_kernel.
// Part of libgrafimo_hip.so (included by grafimo_hip.hip and score_quad_tu.hip).
// Reference lines cited as file:line are relative to /root/reference/src/grafimo/.
#pragma once

#include "gfm_common.hpp"

namespace {

// ---------------------------------------------------------------------------------------
// score kernel geometry
constexpr int kWave = 64;
constexpr int kThreads = 1024;                 // up to 16 waves per workgroup, one workgroup per CU
constexpr int kWavesPerWG = kThreads / kWave;
constexpr unsigned kPoison = 0xFFFFu;          // table entry of a byte that is not A,C,G,T (> 1000*64)
constexpr int kMaxLdsBytes = 160 * 1024;
constexpr int kWGsPerCU = 1;                   // target residency of the score kernel
constexpr int kReserveCUs = 4;                 // CUs left to tail-stream kernels when one is given
constexpr int kHitQueue = 64;                  // per-wave LDS hit queue (entries, >= 64)

// ---------------------------------------------------------------------------------------
// Hit list plumbing shared by the fused and the separate selection.
// Every wave queues its hits in LDS (kHitQueue entries).  A full queue is flushed to the dense
// list with ONE returning global atomic (rare, spread over the run).  What is still queued at
// the end goes to the workgroup's slab of a staging area with plain stores; post_kernel
// appends the slabs.  No returning atomic sits on the kernel's tail: per wave or per
// workgroup, 512..4096 same-word atomics there cost +30..45 us on a 100 us kernel (measured;
// one word sustains ~88 returning atomics/us, MI355X_MICROARCH "dequeue").
// HitCtl rotates its mid-run counters so that the kernel that zeroes one never races a kernel
// that uses it: call k uses slot k % kCtlSlots, its post_kernel zeroes slot (k + kCtlAhead) % kCtlSlots.
// That slot was last used by call k - kWorkspaces, whose post kernel ran before this one, and is next used by
// call k + kWorkspaces, which starts after this post kernel (the scoring workspace it shares is a ring of
// kWorkspaces = GFM_WORKSPACE_RING = 4: see gfm_score_kmers).  (Round 6 tried a ring of eight -- up to eight batches in
// flight, the host seven steps ahead of a tail that runs long: the step's gap over the score kernel stayed where it was,
// 0.2-2.7 us by run, profiles/r06_step_gap3.txt -- and went back to four: a workspace set is ~11 MB per motif handle.)
constexpr int kWorkspaces = GFM_WORKSPACE_RING;
constexpr int kCtlSlots = 2 * GFM_WORKSPACE_RING;
constexpr int kCtlAhead = GFM_WORKSPACE_RING;
struct HitCtl {
    unsigned long long mid[kCtlSlots];   // entries flushed mid-run by the current call
    unsigned long long snap[kCtlSlots];  // *hit_count as the call found it
};
constexpr int kResidPerWG = kWavesPerWG * kHitQueue;  // staging slab entries per workgroup

__device__ inline void hitq_push(long long *hitq, int &qn, bool hit, long long entry, int lane,
                                 const unsigned long long *hit_count, unsigned long long *mid,
                                 long long *hit_rows, long long hit_cap)
{
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);   // (HIP's __ballot(int) would
                                                                          // materialise the bool in a VGPR)
    if (!mask) return;
    const int nh = __popcll(mask);
    if (qn + nh > kHitQueue) {
        unsigned long long base = 0;
        if (lane == 0) base = (hit_count ? *hit_count : 0ull) + atomicAdd(mid, (unsigned long long)qn);
        base = __shfl(base, 0);
        for (int i = lane; i < qn; i += kWave)
            if ((long long)(base + i) < hit_cap) hit_rows[base + i] = hitq[i];
        qn = 0;
    }
    if (hit) hitq[qn + __popcll(mask & ((1ull << lane) - 1ull))] = entry;
    qn += nh;
}

// all waves of the workgroup call this once, after their last push
template <int WAVES>
__device__ inline void hitq_finish(const long long *hitq, int qn, int *wq_n /* shared [WAVES] */,
                                   int wave, int lane, int tid, long long *resid, int *resid_n)
{
    if (lane == 0) wq_n[wave] = qn;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        if (w < wave) base += wq_n[w];
        tot += wq_n[w];
    }
    long long *slab = resid + (size_t)blockIdx.x * kResidPerWG;
    for (int i = lane; i < qn; i += kWave) slab[base + i] = hitq[i];
    if (tid == 0) resid_n[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------------------
// Rows outside a partial LDS window are counted by no-return global atomics.  Their scores are the few bins next to the
// window's edges, and ONE address takes ~88 atomics per microsecond on this chip: 1e5 such rows per launch (0.1 % of
// 1e8) on a handful of bins cost a three-motif launch 100-170 us of its 430 (profiles/r04_config5_shapes.txt).  So the
// counters exist kSpillCopies times and a workgroup uses the copy of its index; post_kernel adds the copies up.
constexpr int kSpillCopies = 16;

// Argument block of one motif in a score launch (score_quad_kernel<W, MM> takes MM of them).
struct MotifArgs {
    const uint16_t *tab;      // [2*NDW][64] pair tables (global)
    int lo, nb, min_val;      // LDS histogram window [lo, lo+nb) (+ the N bin at nb)
    int use_hist;             // 0 none, 1 LDS window -> slab (+ spill outside the window)
    int spill_lo;             // first score of the motif's full range
    int spill_n;              // bins of the full range
    unsigned *spill;          // [kSpillCopies][full range] counters of rows outside the window (post re-zeroes)
    int cutoff;               // rows with score >= cutoff are hits; GFM_NO_SELECT: none
    int slot;                 // HitCtl slot of this call
    int *scores;
    unsigned *partials;
    long long *hit_rows;
    long long hit_cap;
    const unsigned long long *hit_count;   // nullptr: the list restarts at 0 (GFM_FLAG_RESET_HITS)
    HitCtl *ctl;
    long long *resid;
    int *resid_n;
};
template <int MM> struct ScoreArgs {
    MotifArgs m[MM];
    int store_through;   // 1: score stores write through (sc0 sc1), 0: streaming (nt); see score_store_through()
};

// LDS reads of the inner loop go through address-space-3 pointers built from 32-bit offsets: the
// dynamic LDS of this kernel starts at offset 0 (it has no static LDS; checked at entry), so a table
// index IS the address and constant offsets fold into the instruction instead of costing a v_add each
typedef __attribute__((address_space(3))) const unsigned lds_cu32;
typedef __attribute__((address_space(3))) const uint16_t lds_cu16;
typedef __attribute__((address_space(3))) const unsigned long long lds_cu64;
__device__ inline unsigned lds_offset(const void *p)
{
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char *)p;
}

// post_kernel: everything that follows a scoring / selection kernel, in one launch.
//  blocks [0, hist_blocks): sum the per-workgroup histogram slabs into the caller's uint64
//    histogram (bin lo+b; the extra slab bin counts N rows, which score min_val).  A block owns
//    256 bins x kSlabsPerGroup slabs: enough blocks to pull the slabs at L2/HBM rate instead of
//    one latency-bound column walk per bin.
//  blocks [hist_blocks, hist_blocks + spill_blocks): add the spill counters (rows outside a partial
//    LDS window) and hand them back zeroed.
//  the remaining hit_slabs blocks: append residual hit slab g to the dense list at
//    snap + mid + (counts of slabs < g); the first of them publishes the new *hit_count and
//    zeroes the mid-run counter two calls ahead.
constexpr int kSlabsPerGroup = 16;   // 4 / 8 / 16 measure alike; 64 and 256 make the tail the critical path
// One motif's share of a post launch.  A batched score launch (up to kPostJobs = 3 motifs) posts all of them in ONE
// launch, blockIdx.y = motif: in stream order a launch costs ~5 us whatever it does.
constexpr int kPostJobs = 3;
struct PostJob {
    const unsigned *partials;
    unsigned long long *hist64;
    unsigned *spill;
    const long long *resid;
    const int *resid_n;
    HitCtl *ctl;
    long long *hit_rows;
    unsigned long long *hit_count;
    long long hit_cap;
    int nslabs, nb, lo, min_val, bin_blocks, hist_blocks, spill_lo, spill_n, spill_blocks, hit_slabs, par, total;
};
struct PostJobs { PostJob j[kPostJobs]; };

__device__ inline void
post_body(const unsigned *__restrict__ partials, int nslabs, int nb, int lo, int min_val,
          unsigned long long *__restrict__ hist64, int bin_blocks, int hist_blocks,
          unsigned *__restrict__ spill, int spill_lo, int spill_n, int spill_blocks,
          const long long *__restrict__ resid, const int *__restrict__ resid_n, int hit_slabs,
          HitCtl *__restrict__ ctl, int par, long long *__restrict__ hit_rows, long long hit_cap,
          unsigned long long *__restrict__ hit_count)
{
    const int tid = threadIdx.x;
    // A call that selects nothing has no hit-slab block to re-zero the mid-run counter two calls ahead
    // (below); without this a flush of call k would still be counted by call k+3.
    if (blockIdx.x == 0 && tid == 0 && hit_slabs == 0 && ctl) ctl->mid[(par + kCtlAhead) % kCtlSlots] = 0ull;
    if ((int)blockIdx.x < hist_blocks) {
        const int bx = blockIdx.x % bin_blocks, by = blockIdx.x / bin_blocks;
        const int b = bx * 256 + tid;
        if (b > nb) return;
        const int g0 = by * kSlabsPerGroup;
        const int g1 = min(g0 + kSlabsPerGroup, nslabs);
        unsigned long long s = 0;
        const size_t stride = (size_t)(nb + 1);
        // non-temporal: the slabs are read once, beside a score kernel that streams through the same L2 -- with plain
        // loads (lines allocated in L2) this reduction cost that kernel 4 of its 94 us (profiles/r03_tail_cost.txt)
#pragma unroll 8
        for (int g = g0; g < g1; ++g) s += __builtin_nontemporal_load(&partials[g * stride + b]);
        if (s) atomicAdd(&hist64[b == nb ? min_val : lo + b], s);
        return;
    }
    if ((int)blockIdx.x < hist_blocks + spill_blocks) {
        const int b = ((int)blockIdx.x - hist_blocks) * 256 + tid;
        if (b < spill_n) {
            unsigned long long v = 0;
#pragma unroll
            for (int c = 0; c < kSpillCopies; ++c) {
                const unsigned x = spill[(size_t)c * (size_t)spill_n + b];
                if (x) {
                    v += x;
                    spill[(size_t)c * (size_t)spill_n + b] = 0u;
                }
            }
            if (v) atomicAdd(&hist64[spill_lo + b], v);
        }
        return;
    }
    __shared__ int part[256];
    __shared__ int part_all[256];
    const int g = blockIdx.x - hist_blocks - spill_blocks;
    int s = 0, sa = 0;
    for (int k = tid; k < hit_slabs; k += 256) {
        const int v = resid_n[k];
        sa += v;
        if (k < g) s += v;
    }
    part[tid] = s;
    part_all[tid] = sa;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (tid < d) { part[tid] += part[tid + d]; part_all[tid] += part_all[tid + d]; }
        __syncthreads();
    }
    const unsigned long long start = ctl->snap[par] + ctl->mid[par];
    const unsigned long long base = start + (unsigned long long)part[0];
    const int cnt = resid_n[g];
    const long long *slab = resid + (size_t)g * kResidPerWG;
    for (int i = tid; i < cnt; i += 256)
        if ((long long)(base + i) < hit_cap) __builtin_nontemporal_store(__builtin_nontemporal_load(&slab[i]), &hit_rows[base + i]);
    if (g == 0 && tid == 0) {
        *hit_count = start + (unsigned long long)part_all[0];
        ctl->mid[(par + kCtlAhead) % kCtlSlots] = 0ull;
    }
}

__global__ void __launch_bounds__(256) post_kernel(const PostJobs jobs)
{
    const PostJob &q = jobs.j[blockIdx.y];
    if ((int)blockIdx.x >= q.total) return;
    post_body(q.partials, q.nslabs, q.nb, q.lo, q.min_val, q.hist64, q.bin_blocks, q.hist_blocks, q.spill, q.spill_lo,
              q.spill_n, q.spill_blocks, q.resid, q.resid_n, q.hit_slabs, q.ctl, q.par, q.hit_rows, q.hit_cap,
              q.hit_count);
}

// Rows with score >= *cutoff -> hit list (separate pass; used when the cutoff depends on
// the global histogram, i.e. --qvalueT).  Same queue / slab scheme as the fused selection.
constexpr int kSelThreads = 256;
// `gate` (optional): the pass runs only if the candidate list behind it overflowed (*gate_count > gate_cap); then
// it restarts the hit list whatever `hit_count` says.  Otherwise it leaves no residual hits and its post kernel
// republishes the count it found (gfm_select_hits_from: the full pass is the fallback of the candidate filter).
__global__ void __launch_bounds__(kSelThreads)
select_hits_kernel(const int *__restrict__ scores, long long n, const int *__restrict__ cutoff_ptr,
                   long long row_base, long long *__restrict__ hit_rows, long long hit_cap,
                   const unsigned long long *__restrict__ hit_count, HitCtl *__restrict__ ctl,
                   int par, long long *__restrict__ resid, int *__restrict__ resid_n,
                   const unsigned long long *__restrict__ gate_count, long long gate_cap)
{
    __shared__ long long hq[kSelThreads / kWave][kHitQueue];
    __shared__ int wq_n[kSelThreads / kWave];
    const int cutoff = *cutoff_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long *hitq = hq[wave];
    int qn = 0;
    if (gate_count) {
        if ((long long)*gate_count <= gate_cap) {      // the candidates were complete: nothing to do
            if (tid == 0) {
                resid_n[blockIdx.x] = 0;
                if (blockIdx.x == 0) ctl->snap[par] = hit_count ? *hit_count : 0ull;
            }
            return;
        }
        hit_count = nullptr;                           // overflow: the list restarts from the scores
    }
    if (blockIdx.x == 0 && tid == 0) ctl->snap[par] = hit_count ? *hit_count : 0ull;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    const long long n4 = (n + 3) >> 2;
    const long long iters = (n4 + nthreads - 1) / nthreads;  // wave-uniform trip count
    const long long t0 = (long long)blockIdx.x * blockDim.x + tid;
    for (long long it = 0; it < iters; ++it) {
        const long long i = t0 + it * nthreads;
        int sc[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
        if (i * 4 + 4 <= n) {
            const int4 v = reinterpret_cast<const int4 *>(scores)[i];
            sc[0] = v.x; sc[1] = v.y; sc[2] = v.z; sc[3] = v.w;
        } else {
            for (int j = 0; j < 4; ++j)
                if (i * 4 + j < n) sc[j] = scores[i * 4 + j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            hitq_push(hitq, qn, sc[j] >= cutoff,
                      ((row_base + i * 4 + j) << GFM_HIT_SCORE_BITS) | (long long)sc[j], lane,
                      hit_count, &ctl->mid[par], hit_rows, hit_cap);
    }
    hitq_finish<kSelThreads / kWave>(hitq, qn, wq_n, wave, lane, tid, resid, resid_n);
}

// The entries of a candidate list (hit entries of rows with score >= some LOWER cutoff, as the fused selection of
// the score kernel leaves them) whose score reaches *cutoff -> a new hit list.  A q-value threshold needs the
// global histogram before its cutoff is known, but q >= p: the rows with q < t are among those with p < t, which
// the score kernel can select on the fly -- the filter then reads a few MB of entries instead of every score
// (1e8 rows: 400 MB).  A candidate list that overflowed (count > capacity) is reported through the published
// count (= the candidates' count) and repaired by the gated select_hits_kernel that follows.
// Compaction by blocks of kFilterPerBlock entries: every block counts its survivors (block-wide scan), takes its place
// in the output with ONE returning atomic and writes them there (the order of a hit list does not matter: it is
// sorted on the host).  The wave queues of the other selection kernels would flush once per 64 survivors -- with
// nearly every candidate surviving that was 15 600 returning atomics on one word for 1e6 candidates (~180 us).
// *hit_count must be 0 when the kernel starts.
constexpr int kFilterPerThread = 16;
constexpr int kFilterPerBlock = kSelThreads * kFilterPerThread;
__global__ void __launch_bounds__(kSelThreads)
filter_hits_kernel(const long long *__restrict__ cand, const unsigned long long *__restrict__ cand_count,
                   long long cand_cap, const int *__restrict__ cutoff_ptr, long long *__restrict__ hit_rows,
                   long long hit_cap, unsigned long long *__restrict__ hit_count)
{
    __shared__ int wave_tot[kSelThreads / kWave];
    __shared__ unsigned long long block_base;
    const int cutoff = *cutoff_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long have = *cand_count;
    if ((long long)have > cand_cap) {      // overflowed candidates: say so through the count; the gated pass repairs it
        if (blockIdx.x == 0 && tid == 0) *hit_count = have;
        return;
    }
    const long long n = (long long)have;
    const long long first = (long long)blockIdx.x * kFilterPerBlock;
    if (first >= n) return;
    long long e[kFilterPerThread];
    int mine = 0;
#pragma unroll
    for (int j = 0; j < kFilterPerThread; ++j) {
        const long long i = first + (long long)j * kSelThreads + tid;      // coalesced
        e[j] = i < n ? cand[i] : -1ll;
        const bool keep = i < n && (int)(e[j] & ((1ll << GFM_HIT_SCORE_BITS) - 1)) >= cutoff;
        if (!keep) e[j] = -1ll;
        mine += keep ? 1 : 0;
    }
    // exclusive prefix of `mine` over the block
    int incl = mine;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == kWave - 1) wave_tot[wave] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kSelThreads / kWave; ++w) {
        if (w < wave) before += wave_tot[w];
        total += wave_tot[w];
    }
    if (tid == 0) block_base = total ? atomicAdd(hit_count, (unsigned long long)total) : 0ull;
    __syncthreads();
    unsigned long long at = block_base + (unsigned long long)(before + incl - mine);
#pragma unroll
    for (int j = 0; j < kFilterPerThread; ++j)
        if (e[j] >= 0) {
            if ((long long)at < hit_cap) hit_rows[at] = e[j];
            ++at;
        }
}

}  // namespace
