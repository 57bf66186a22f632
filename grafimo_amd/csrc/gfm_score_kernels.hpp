// gfm_score_kernels.hpp -- score_hist_kernel (the dominant, HBM-bound kernel), post_kernel, select_hits_kernel
// Part of libgrafimo_hip.so (one translation unit: included by grafimo_hip.hip only).
// Reference lines cited as file:line are relative to /root/reference/src/grafimo/.
#pragma once

#include "gfm_common.hpp"

namespace {

// ---------------------------------------------------------------------------------------
// score kernel geometry
constexpr int kWave = 64;
constexpr int kThreads = 1024;                 // 16 waves per workgroup, one workgroup per CU
constexpr int kWavesPerWG = kThreads / kWave;
// Widths above 44 (NDW >= 12) run 8 waves per workgroup: 16 strips of 128 x W bytes would leave the
// LDS histogram window only a few thousand bins (W=64: 4 K), and the rows outside the window pay a
// global atomic each.
__host__ __device__ constexpr int waves_for_ndw(int ndw) { return ndw >= 12 ? kWavesPerWG / 2 : kWavesPerWG; }
constexpr int kChunk = 128;                    // k-mers per wave per iteration (multiple of 64)
constexpr unsigned kPoison = 0xFFFFu;          // table entry of a byte that is not A,C,G,T (> 1000*64)
constexpr int kMaxLdsBytes = 160 * 1024;
constexpr int kWGsPerCU = 1;                   // target residency of the score kernel
constexpr int kReserveCUs = 4;                 // CUs left to tail-stream kernels when one is given
constexpr int kHitQueue = 64;                  // per-wave LDS hit queue (entries, >= 64)
constexpr int kDepth = 3;                      // chunks prefetched ahead per wave

// per-wave LDS strip: the staged chunk (+8 B slack for the last row's trailing dword),
// followed by the wave's hit queue(s), one per motif of the launch
// Rows of W % 16 == 0 bytes would start 4, 8, 12 or 16 dwords apart in the strip: the 64 lanes of a
// row-wise read then share 8, 4, 8 or 2 LDS banks (W=32: 189 us for 2e7 rows).  Those widths are
// staged one dword apart (row stride W + 4: an odd number of dwords, conflict free: 142 us).
__host__ __device__ inline int row_pad_bytes(int W) { return (W % 16 == 0) ? 4 : 0; }
__host__ __device__ inline int stage_data_bytes(int W)
{
    return ((kChunk * (W + row_pad_bytes(W)) + 15) & ~15) + 8;
}
__host__ __device__ inline int stage_stride_bytes(int W, int mm) { return stage_data_bytes(W) + mm * kHitQueue * 8; }

// ---------------------------------------------------------------------------------------
// Hit list plumbing shared by the fused and the separate selection.
// Every wave queues its hits in LDS (kHitQueue entries).  A full queue is flushed to the dense
// list with ONE returning global atomic (rare, spread over the run).  What is still queued at
// the end goes to the workgroup's slab of a staging area with plain stores; post_kernel
// appends the slabs.  No returning atomic sits on the kernel's tail: per wave or per
// workgroup, 512..4096 same-word atomics there cost +30..45 us on a 100 us kernel (measured;
// one word sustains ~88 returning atomics/us, MI355X_MICROARCH "dequeue").
// HitCtl rotates three mid-run counters so that the kernel that zeroes one never races a
// kernel that uses it: call k uses slot k%3, its post_kernel zeroes slot (k+2)%3.
struct HitCtl {
    unsigned long long mid[3];   // entries flushed mid-run by the current call
    unsigned long long snap[3];  // *hit_count as the call found it
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
// score_hist_kernel<NDW, SELECT>
//
// One wave owns a stream of 128-k-mer chunks.  A chunk is W*128 contiguous bytes of the
// row-major uint8 [n][W] matrix (16-byte aligned because 128*W % 16 == 0), fetched with
// fully coalesced 16 B/lane loads into registers one chunk ahead, parked in a wave-private
// LDS strip, and re-read row-wise: lane r takes k-mer r as NDW+1 aligned dwords that
// v_alignbit turns into NDW dwords of consecutive bases.  Bases are looked up TWO at a time:
// bits 1..3 of an ASCII code ((c>>1)&7: A/a 0, C/c 1, T/t 2, G/g 3, N/n 7) of two
// neighbouring bases form a 6-bit index into that pair's 64-entry uint16 table in LDS
// (entry = sm[b0][2p] + sm[b1][2p+1]); the 16 hot entries of a pair sit in 8 distinct banks, so
// the lookup is conflict free, and a dword of 4 bases costs 7 VALU + 2 LDS instead of 9 + 4.
// Entries with a code 4..7 hold kPoison: a k-mer that touched one scores min_val
// (score_sequences.py:376-378).  A position >= W contributes 0 whatever its byte, so the tail
// of the last dword needs no masking.  Scores go out as coalesced int32; the score histogram
// is built with LDS atomics in a per-workgroup window [lo, lo+nb) (+1 bin for N rows)
// and flushed once per workgroup as a plain-store slab (no global atomics).  Wide motifs whose
// whole score range does not fit next to the strips keep the window over the densest part of the
// background score distribution; the few rows outside it go to a global spill array.
// MM motifs of the same width can share ONE read of the k-mers (BASELINE config 5: per
// (k-mer, motif) pair the algorithmic bytes drop from W + 4 to W/MM + 4): the staged strip is
// scored against MM table sets, each motif has its own histogram window, hit queue and outputs.
struct MotifArgs {
    const uint16_t *tab;      // [2*NDW][64] pair tables (global)
    int lo, nb, min_val;      // LDS histogram window [lo, lo+nb) (+ the N bin at nb)
    int use_hist;             // 0 none, 1 LDS window -> slab (+ spill outside the window)
    int spill_lo;             // first score of the motif's full range
    unsigned *spill;          // [full range] counters of rows outside the window (post re-zeroes)
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
template <int MM> struct ScoreArgs { MotifArgs m[MM]; };

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

template <int NDW, int MM>
__global__ void __launch_bounds__(waves_for_ndw(NDW) * kWave)
score_hist_kernel(const uint8_t *__restrict__ kmers, long long n, int W, long long row_base,
                  const ScoreArgs<MM> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kWaves = waves_for_ndw(NDW);     // waves of this workgroup
    constexpr int kWgThreads = kWaves * kWave;
    constexpr int kTabBytes = 2 * NDW * 64 * 2;  // 2*NDW base pairs x (8 x 8 codes) x uint16
    constexpr int kLoads = (kChunk * 4 * NDW + 1023) / 1024;  // 16 B loads per lane per chunk

    // MM == 1: uint16 pair tables.  MM > 1: ONE table of 64-bit entries packing the motifs' partial
    // scores in 19-bit fields (a k-mer's score is <= 64000 < 2^19, so fields never carry) plus, in
    // bits 57.., a count of invalid codes: one ds_read_b64 + one 64-bit add per base pair serves
    // all motifs of the launch, so the inner loop costs the same for 1, 2 or 3 motifs.
    constexpr int kTabRegion = (MM == 1 ? 1 : 4) * kTabBytes;
    if (lds_offset(smem) != 0u) __builtin_trap();   // the lookups below use absolute LDS offsets
    unsigned char *tab = smem;
    unsigned char *stage_base = smem + kTabRegion;
    const int sstride = stage_stride_bytes(W, MM);
    unsigned *hist[MM];
    {
        unsigned *h = reinterpret_cast<unsigned *>(stage_base + kWaves * sstride);
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            hist[m] = h;
            if (a.m[m].use_hist) h += a.m[m].nb + 1;
        }
    }

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    // wave-uniform by construction; readfirstlane tells the compiler, so the chunk index, the strip
    // address and the row/score addressing of a chunk stay in scalar registers
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    if constexpr (MM == 1) {
        for (int i = tid; i < kTabBytes / 2; i += kWgThreads)
            reinterpret_cast<uint16_t *>(tab)[i] = a.m[0].tab[i];
    } else {
        for (int i = tid; i < kTabBytes / 2; i += kWgThreads) {
            unsigned long long v = 0;
            const unsigned t0 = a.m[0].tab[i];
            if (t0 == kPoison) {
                v = 1ull << 57;                       // same codes are invalid for every motif
            } else {
#pragma unroll
                for (int m = 0; m < MM; ++m) v |= (unsigned long long)a.m[m].tab[i] << (19 * m);
            }
            reinterpret_cast<unsigned long long *>(tab)[i] = v;
        }
    }
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        if (a.m[m].use_hist)
            for (int i = tid; i <= a.m[m].nb; i += kWgThreads) hist[m][i] = 0u;
    }
    __syncthreads();

    unsigned char *stage = stage_base + wave * sstride;
    const unsigned stage_off = (unsigned)(kTabRegion + wave * sstride);   // its absolute LDS offset
    const long long total_bytes = n * (long long)W;
    const long long nchunks = (n + kChunk - 1) / kChunk;
    const int chunk_bytes = kChunk * W;
    const long long cstride = (long long)gridDim.x * kWaves;

    // only the instantiations that can see a width of 16, 32, 48 or 64 carry the padded staging
    const int pad = (NDW % 4 == 0) ? row_pad_bytes(W) : 0;
    const unsigned pad_inv = pad ? (65536u + (unsigned)(W >> 4) - 1u) / (unsigned)(W >> 4) : 0u;  // ceil(2^16 / (W/16))
    uint4 pre[kDepth][kLoads];
    auto fetch = [&](uint4 (&dst)[kLoads], long long c) {
        const long long cbase = c * (long long)chunk_bytes;
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
            const int off = i * 1024 + lane * 16;
            const long long g = cbase + off;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (off < chunk_bytes) {
                if (g + 16 <= total_bytes) {
                    // once-read stream: non-temporal policy (plain loads: 5.1 TB/s, nt: 6.2 TB/s on
                    // the same byte mix, scripts/micro/stream_bw_nt.hip)
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(kmers + g));
                    v = make_uint4(t.x, t.y, t.z, t.w);
                } else if (g < total_bytes) {  // ragged end of the matrix: byte loads
                    unsigned t[4] = {0u, 0u, 0u, 0u};
                    for (int b = 0; b < 16 && g + b < total_bytes; ++b)
                        t[b >> 2] |= (unsigned)kmers[g + b] << (8 * (b & 3));
                    v = make_uint4(t[0], t[1], t[2], t[3]);
                }
            }
            dst[i] = v;
        }
    };

    long long *hitq[MM];
    int qn[MM];  // wave-uniform
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        hitq[m] = reinterpret_cast<long long *>(stage + stage_data_bytes(W)) + m * kHitQueue;
        qn[m] = 0;
        if (a.m[m].cutoff != GFM_NO_SELECT && blockIdx.x == 0 && tid == 0)
            a.m[m].ctl->snap[a.m[m].slot] = a.m[m].hit_count ? *a.m[m].hit_count : 0ull;
    }

    long long c = (long long)blockIdx.x * kWaves + wave;
#pragma unroll
    for (int d = 0; d < kDepth; ++d)
        if (c + d * cstride < nchunks) fetch(pre[d], c + d * cstride);
    while (c < nchunks) {
#pragma unroll
    for (int d = 0; d < kDepth; ++d) {
        if (c >= nchunks) break;
        const long long crow = c * kChunk;                                        // first row of the chunk
        const int rem = (int)((n - crow) < (long long)kChunk ? (n - crow) : (long long)kChunk);  // live rows
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
            const int off = i * 1024 + lane * 16;
            if (off < chunk_bytes) {
                if (pad) {   // W % 16 == 0: the 16-byte piece lies inside row off / W
                    const unsigned r = ((unsigned)(off >> 4) * pad_inv) >> 16;
                    unsigned *dst = reinterpret_cast<unsigned *>(stage + off + 4 * r);
                    dst[0] = pre[d][i].x;
                    dst[1] = pre[d][i].y;
                    dst[2] = pre[d][i].z;
                    dst[3] = pre[d][i].w;
                } else {
                    *reinterpret_cast<uint4 *>(stage + off) = pre[d][i];
                }
            }
        }
        if (c + kDepth * cstride < nchunks) fetch(pre[d], c + kDepth * cstride);
        // LDS ops of one wave execute in program order; the fence only pins the compiler.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

#pragma unroll
        for (int p = 0; p < kChunk / kWave; ++p) {
            const int k = p * kWave + lane;
            const int boff = k * (W + pad);
            const unsigned sh = (unsigned)(boff & 3) * 8u;
            lds_cu32 *src = (lds_cu32 *)(uintptr_t)(stage_off + (unsigned)(boff & ~3));
            unsigned w[NDW + 1];
#pragma unroll
            for (int d2 = 0; d2 <= NDW; ++d2) w[d2] = src[d2];
            int acc[MM];
            bool any_n;
            if constexpr (MM == 1) {
                int s1 = 0;
#pragma unroll
                for (int d2 = 0; d2 < NDW; ++d2) {
                    const unsigned x = __builtin_amdgcn_alignbit(w[d2 + 1], w[d2], sh);
                    // one bit-select: bits 1..3 of each 16-bit half from x (first base of a pair), the
                    // other bits from x >> 5 (bits 4..6: second base); then one mask per table offset
                    unsigned y;
                    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(y) : "s"(0x000E000Eu), "v"(x), "v"(x >> 5));
                    const unsigned e0 = y & 0x7Eu;
                    const unsigned e1 = (y >> 16) & 0x7Eu;
                    s1 += *(lds_cu16 *)(uintptr_t)(e0 + (unsigned)((2 * d2) * 128));
                    s1 += *(lds_cu16 *)(uintptr_t)(e1 + (unsigned)((2 * d2 + 1) * 128));
                }
                any_n = (unsigned)s1 >= kPoison;
                acc[0] = s1;
            } else {
                unsigned long long s64 = 0;
#pragma unroll
                for (int d2 = 0; d2 < NDW; ++d2) {
                    const unsigned x = __builtin_amdgcn_alignbit(w[d2 + 1], w[d2], sh);
                    unsigned y;
                    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(y) : "s"(0x000E000Eu), "v"(x), "v"(x >> 5));
                    const unsigned e0 = (y & 0x7Eu) << 2;            // 8-byte entries
                    const unsigned e1 = ((y >> 16) & 0x7Eu) << 2;
                    s64 += *(lds_cu64 *)(uintptr_t)(e0 + (unsigned)((2 * d2) * 512));
                    s64 += *(lds_cu64 *)(uintptr_t)(e1 + (unsigned)((2 * d2 + 1) * 512));
                }
                any_n = (s64 >> 57) != 0;
#pragma unroll
                for (int m = 0; m < MM; ++m) acc[m] = (int)((s64 >> (19 * m)) & 0x7FFFFull);
            }
            const bool live = k < rem;
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                const MotifArgs &ma = a.m[m];
                const bool is_n = any_n;
                const int score = is_n ? ma.min_val : acc[m];
                if (live) {
                    __builtin_nontemporal_store(score, ma.scores + crow + k);   // scalar base + lane offset
                    if (ma.use_hist) {
                        const unsigned off = (unsigned)(score - ma.lo);
                        if (is_n || off < (unsigned)ma.nb)
                            atomicAdd(&hist[m][is_n ? (unsigned)ma.nb : off], 1u);
                        else
                            atomicAdd(&ma.spill[score - ma.spill_lo], 1u);   // outside the window: rare
                    }
                }
                if (ma.cutoff != GFM_NO_SELECT)
                    hitq_push(hitq[m], qn[m], live && score >= ma.cutoff,
                              ((row_base + crow + k) << GFM_HIT_SCORE_BITS) | (long long)score, lane,
                              ma.hit_count, &ma.ctl->mid[ma.slot], ma.hit_rows, ma.hit_cap);
            }
        }
        // the strip is rewritten next iteration: keep this iteration's reads ahead of it
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        c += cstride;
    }
    }

    // the lookup tables are dead once every wave has left the loop: their LDS holds the per-wave
    // queue lengths (no static LDS)
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        const MotifArgs &ma = a.m[m];
        if (ma.cutoff != GFM_NO_SELECT)
            hitq_finish<kWaves>(hitq[m], qn[m], reinterpret_cast<int *>(tab + m * kTabBytes), wave,
                                     lane, tid, ma.resid, ma.resid_n);
        if (ma.use_hist) {
            unsigned *slab = ma.partials + (size_t)blockIdx.x * (size_t)(ma.nb + 1);
            for (int i = tid; i <= ma.nb; i += kWgThreads) slab[i] = hist[m][i];
        }
    }
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
constexpr int kSlabsPerGroup = 16;
__global__ void __launch_bounds__(256)
post_kernel(const unsigned *__restrict__ partials, int nslabs, int nb, int lo, int min_val,
            unsigned long long *__restrict__ hist64, int bin_blocks, int hist_blocks,
            unsigned *__restrict__ spill, int spill_lo, int spill_n, int spill_blocks,
            const long long *__restrict__ resid, const int *__restrict__ resid_n, int hit_slabs,
            HitCtl *__restrict__ ctl, int par, long long *__restrict__ hit_rows, long long hit_cap,
            unsigned long long *__restrict__ hit_count)
{
    const int tid = threadIdx.x;
    // A call that selects nothing has no hit-slab block to re-zero the mid-run counter two calls ahead
    // (below); without this a flush of call k would still be counted by call k+3.
    if (blockIdx.x == 0 && tid == 0 && hit_slabs == 0 && ctl) ctl->mid[(par + 2) % 3] = 0ull;
    if ((int)blockIdx.x < hist_blocks) {
        const int bx = blockIdx.x % bin_blocks, by = blockIdx.x / bin_blocks;
        const int b = bx * 256 + tid;
        if (b > nb) return;
        const int g0 = by * kSlabsPerGroup;
        const int g1 = min(g0 + kSlabsPerGroup, nslabs);
        unsigned long long s = 0;
        const size_t stride = (size_t)(nb + 1);
#pragma unroll 8
        for (int g = g0; g < g1; ++g) s += partials[g * stride + b];
        if (s) atomicAdd(&hist64[b == nb ? min_val : lo + b], s);
        return;
    }
    if ((int)blockIdx.x < hist_blocks + spill_blocks) {
        const int b = ((int)blockIdx.x - hist_blocks) * 256 + tid;
        if (b < spill_n) {
            const unsigned v = spill[b];
            if (v) {
                atomicAdd(&hist64[spill_lo + b], (unsigned long long)v);
                spill[b] = 0u;
            }
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
        if ((long long)(base + i) < hit_cap) hit_rows[base + i] = slab[i];
    if (g == 0 && tid == 0) {
        *hit_count = start + (unsigned long long)part_all[0];
        ctl->mid[(par + 2) % 3] = 0ull;
    }
}

// Rows with score >= *cutoff -> hit list (separate pass; used when the cutoff depends on
// the global histogram, i.e. --qvalueT).  Same queue / slab scheme as the fused selection.
constexpr int kSelThreads = 256;
__global__ void __launch_bounds__(kSelThreads)
select_hits_kernel(const int *__restrict__ scores, long long n, const int *__restrict__ cutoff_ptr,
                   long long row_base, long long *__restrict__ hit_rows, long long hit_cap,
                   const unsigned long long *__restrict__ hit_count, HitCtl *__restrict__ ctl,
                   int par, long long *__restrict__ resid, int *__restrict__ resid_n)
{
    __shared__ long long hq[kSelThreads / kWave][kHitQueue];
    __shared__ int wq_n[kSelThreads / kWave];
    const int cutoff = *cutoff_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long *hitq = hq[wave];
    int qn = 0;
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


}  // namespace
