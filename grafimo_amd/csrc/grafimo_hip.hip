// grafimo_hip.hip -- MI355X (gfx950 / CDNA4) kernels + C ABI for GRAFIMO's k-mer scoring path.
//
// Written for gfx950 only: 64-lane wavefronts, 160 KiB LDS per CU, 256 CUs in 8 XCDs.
// The hot op is an HBM-bound gather (W table lookups + W integer adds per k-mer,
// < 1 op per byte): no MFMA.  See DESIGN.md for the roofline and the data layout.
//
// Reference lines cited as file:line are relative to /root/reference/src/grafimo/.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "grafimo_hip.h"

#define GFM_API extern "C" __attribute__((visibility("default")))

namespace {

constexpr int kRange = 1000;             // utils.py:26
constexpr double kLogFactor = 1.44269504;  // utils.py:25 (truncated 1/ln2, verbatim)

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(GFM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                 \
    } while (0)

// device allocation that frees itself (error paths of the host-side helpers)
template <typename T> struct DevBuf {
    T *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc(&p, sizeof(T) * count); }
    operator T *() const { return p; }
};

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
    const unsigned long long mask = __ballot(hit);
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
    const int wave = tid >> 6;

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
            const long long row = c * kChunk + k;
            const int boff = k * (W + pad);
            const unsigned sh = (unsigned)(boff & 3) * 8u;
            const unsigned *src = reinterpret_cast<const unsigned *>(stage + (boff & ~3));
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
                    const unsigned xm = x & 0x0E0E0E0Eu;   // 2 * ((c >> 1) & 7) per byte
                    const unsigned y = xm | (xm >> 5);     // bytes 0 and 2: 2*(code_lo + 8*code_hi)
                    const unsigned e0 = y & 0x7Eu;
                    const unsigned e1 = (y >> 16) & 0x7Eu;
                    s1 += *reinterpret_cast<const uint16_t *>(tab + (2 * d2) * 128 + e0);
                    s1 += *reinterpret_cast<const uint16_t *>(tab + (2 * d2 + 1) * 128 + e1);
                }
                any_n = (unsigned)s1 >= kPoison;
                acc[0] = s1;
            } else {
                unsigned long long s64 = 0;
#pragma unroll
                for (int d2 = 0; d2 < NDW; ++d2) {
                    const unsigned x = __builtin_amdgcn_alignbit(w[d2 + 1], w[d2], sh);
                    const unsigned xm = x & 0x0E0E0E0Eu;
                    const unsigned y = xm | (xm >> 5);
                    const unsigned e0 = (y & 0x7Eu) << 2;            // 8-byte entries
                    const unsigned e1 = ((y >> 16) & 0x7Eu) << 2;
                    s64 += *reinterpret_cast<const unsigned long long *>(tab + (2 * d2) * 512 + e0);
                    s64 += *reinterpret_cast<const unsigned long long *>(tab + (2 * d2 + 1) * 512 + e1);
                }
                any_n = (s64 >> 57) != 0;
#pragma unroll
                for (int m = 0; m < MM; ++m) acc[m] = (int)((s64 >> (19 * m)) & 0x7FFFFull);
            }
            const bool live = row < n;
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                const MotifArgs &ma = a.m[m];
                const bool is_n = any_n;
                const int score = is_n ? ma.min_val : acc[m];
                if (live) {
                    __builtin_nontemporal_store(score, ma.scores + row);   // written once, read later
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
                              ((row_base + row) << GFM_HIT_SCORE_BITS) | (long long)score, lane,
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

// ---------------------------------------------------------------------------------------
// pvalue_dp_kernel: score-distribution DP of comp_pval_mat (motif_processing.pyx:552-603),
// one 1024-thread workgroup per motif, gather form:
//   cur[t] = sum over n in A,C,G,T of prev[t - sm[n][pos]] * bg[n]
// accumulated per target in A->C->G->T order with the product rounded before the add
// (__dmul_rn/__dadd_rn: no FMA contraction) and the reference's `> 0` support test.  The
// reference scatters, but each (n, idx) pair hits a distinct target once per n and n runs
// outermost, so per target the additions arrive in exactly this order: bit-identical.
// Rows ping-pong in global memory (they live in L2: 2*L*8 B <= 1 MB); only the reachable
// window [cum_lo[pos], cum_hi[pos]] of a row is computed or read.
constexpr int kDpThreads = 1024;

__global__ void __launch_bounds__(kDpThreads)
pvalue_dp_kernel(const int *__restrict__ sm, const double *__restrict__ bg, int W, int L,
                 const int *__restrict__ cum_lo, const int *__restrict__ cum_hi,
                 double *__restrict__ buf, double *__restrict__ pmf_out)
{
    double *cur = buf;
    double *prev = buf + L;
    const int tid = threadIdx.x;
    {   // position 0 (motif_processing.pyx:593-594)
        const int l0 = cum_lo[0], h0 = cum_hi[0];
        for (int t = l0 + tid; t <= h0; t += kDpThreads) cur[t] = 0.0;
        __syncthreads();
        if (tid == 0)
            for (int nuc = 0; nuc < 4; ++nuc) {
                const int s = sm[nuc * W];
                cur[s] = __dadd_rn(cur[s], __dmul_rn(1.0, bg[nuc]));
            }
        __syncthreads();
    }
    for (int pos = 1; pos < W; ++pos) {
        double *tmp = cur; cur = prev; prev = tmp;
        const int lp = cum_lo[pos - 1], hp = cum_hi[pos - 1];
        const int lc = cum_lo[pos], hc = cum_hi[pos];
        const int s0 = sm[0 * W + pos], s1 = sm[1 * W + pos], s2 = sm[2 * W + pos],
                  s3 = sm[3 * W + pos];
        const double b0 = bg[0], b1 = bg[1], b2 = bg[2], b3 = bg[3];
        for (int t = lc + tid; t <= hc; t += kDpThreads) {
            double acc = 0.0;
            int idx = t - s0;
            if (idx >= lp && idx <= hp) { const double v = prev[idx]; if (v > 0) acc = __dadd_rn(acc, __dmul_rn(v, b0)); }
            idx = t - s1;
            if (idx >= lp && idx <= hp) { const double v = prev[idx]; if (v > 0) acc = __dadd_rn(acc, __dmul_rn(v, b1)); }
            idx = t - s2;
            if (idx >= lp && idx <= hp) { const double v = prev[idx]; if (v > 0) acc = __dadd_rn(acc, __dmul_rn(v, b2)); }
            idx = t - s3;
            if (idx >= lp && idx <= hp) { const double v = prev[idx]; if (v > 0) acc = __dadd_rn(acc, __dmul_rn(v, b3)); }
            cur[t] = acc;
        }
        __syncthreads();
    }
    const int lf = cum_lo[W - 1], hf = cum_hi[W - 1];
    for (int t = tid; t < L; t += kDpThreads) pmf_out[t] = (t >= lf && t <= hf) ? cur[t] : 0.0;
}

// ---------------------------------------------------------------------------------------
// block-wide scans over a table of L entries split into 1024 contiguous segments
constexpr int kScanThreads = 1024;

// p_table[s] = (sum_{t>=s} pmf[t]) / (sum_t pmf[t])   -- O(1) form of
// `pval_mat[score:].sum() / pval_mat.sum()` (score_sequences.py:390-391).
// Blocked suffix sum that stays EXACTLY monotone: thread t sums its contiguous segment top-down
// (local running sums L_j), one lane chains the 1024 segment totals top-down (carry c_t), and
// suffix[j] = c_t + L_j.  fl(c + L) is monotone in L, and at a segment's bottom c_t + L = c_t + s_t
// is the very operation that produced the carry of the segment below, so no boundary can step
// the wrong way; p_table[s] == 1.0 exactly for every s at or below the lowest reachable score.
// (A scan with mixed association orders broke monotonicity by 1 ulp; a fully sequential chain
// took 0.5 ms.)
__global__ void __launch_bounds__(kScanThreads)
ptable_kernel(const double *__restrict__ pmf, int L, int lo, int hi, double *__restrict__ ptable)
{
    __shared__ double carry[kScanThreads];
    __shared__ double tot_s;
    const int tid = threadIdx.x;
    const int nb = hi - lo + 1;
    const int per = (nb + kScanThreads - 1) / kScanThreads;
    const int a = lo + min(tid * per, nb), b = lo + min(tid * per + per, nb);
    double run = 0.0;
    for (int j = b - 1; j >= a; --j) {
        run += pmf[j];
        ptable[j] = run;        // local running sum, finished below
    }
    carry[tid] = run;
    __syncthreads();
    if (tid == 0) {
        double c = 0.0;
        for (int t = kScanThreads - 1; t >= 0; --t) {
            const double s = carry[t];
            carry[t] = c;       // everything above segment t
            c = c + s;
        }
        tot_s = c;
    }
    __syncthreads();
    const double c = carry[tid], tot = tot_s;
    for (int j = a; j < b; ++j) ptable[j] = (c + ptable[j]) / tot;
    for (int j = tid; j < lo; j += kScanThreads) ptable[j] = tot / tot;
    for (int j = hi + 1 + tid; j < L; j += kScanThreads) ptable[j] = 0.0;
}

// q-value of every score from the histogram (Benjamini-Hochberg as statsmodels'
// fdrcorrection evaluates it: raw = p / (rank/n), reverse cumulative minimum, clip 1),
// plus the selection cutoff.  Ranks: all rows sharing a score share a p-value; the
// largest rank in the tie group is C(s) = #rows with score >= s, and the cumulative
// minimum makes the whole group take p(s) / (C(s)/n).
// Only the reachable window [lo, hi] can hold counts, plus bin min_val for rows with an N
// (below the window: p = 1, rank = n, raw = 1).
//
// Three small multi-block kernels, one bin per thread (256-thread blocks, a handful of
// registers), instead of one big workgroup: a 1024-thread workgroup holding the window in
// registers needs an EMPTY CU, and next to the persistent score grid it found none -- on the
// tail stream it simply waited for the score kernel to end (measured: 19 us alone, 76-95 us
// "overlapped", gating the pipeline).  Small blocks slot in beside resident score workgroups.
//   q_count_kernel : per-block bin totals
//   q_raw_kernel   : C(s) by block-suffix + in-block scan, raw(s) -> qtable (temporary), block minima
//   q_final_kernel : prefix minimum -> q(s), cutoff, clears
constexpr int kQThreads = 256;
struct QWork {
    unsigned long long blk_cnt[256];
    double blk_min[256];
    unsigned long long n_rows_N;
};

__device__ inline unsigned long long block_sum_u64(unsigned long long v, unsigned long long *sh)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < kQThreads / kWave; ++w) t += sh[w];
    __syncthreads();
    return t;
}

__global__ void __launch_bounds__(kQThreads)
q_count_kernel(const unsigned long long *__restrict__ hist, int L, int lo, int hi, int min_val,
               QWork *__restrict__ ws, int *__restrict__ cutoff_out)
{
    __shared__ unsigned long long sh[kQThreads / kWave];
    const int j = lo + blockIdx.x * kQThreads + threadIdx.x;
    const unsigned long long h = j <= hi ? hist[j] : 0ull;
    const unsigned long long tot = block_sum_u64(h, sh);
    if (threadIdx.x == 0) {
        ws->blk_cnt[blockIdx.x] = tot;
        if (blockIdx.x == 0) {
            const bool n_outside = min_val < lo || min_val > hi;
            ws->n_rows_N = n_outside ? hist[min_val] : 0ull;
            if (cutoff_out) *cutoff_out = L;
        }
    }
}

__global__ void __launch_bounds__(kQThreads)
q_raw_kernel(const unsigned long long *__restrict__ hist, const double *__restrict__ ptable, int lo,
             int hi, QWork *__restrict__ ws, double *__restrict__ raw_out)
{
    __shared__ unsigned long long sh[kQThreads / kWave];
    __shared__ double shm[kQThreads / kWave];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = gridDim.x, blk = blockIdx.x;
    // totals: all rows, and the rows in blocks above this one
    unsigned long long a = 0, t = 0;
    for (int b = tid; b < nblk; b += kQThreads) {
        const unsigned long long v = ws->blk_cnt[b];
        t += v;
        if (b > blk) a += v;
    }
    const unsigned long long n = block_sum_u64(t, sh) + ws->n_rows_N;
    const unsigned long long above_blocks = block_sum_u64(a, sh);
    const double nd = (double)n;
    const int j = lo + blk * kQThreads + tid;
    const bool ok = j <= hi;
    const unsigned long long h = ok ? hist[j] : 0ull;
    // inclusive suffix sum inside the block
    unsigned long long cs = h;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const unsigned long long v = __shfl_down(cs, d);
        if (lane + d < kWave) cs += v;
    }
    if (lane == 0) sh[wave] = cs;
    __syncthreads();
    unsigned long long waves_above = 0;
#pragma unroll
    for (int w = 0; w < kQThreads / kWave; ++w)
        if (w > wave) waves_above += sh[w];
    const unsigned long long c_ge = cs + waves_above + above_blocks;
    const double raw = h ? ptable[j] / ((double)c_ge / nd) : INFINITY;
    if (ok) raw_out[j] = raw;
    double m = raw;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmin(m, __shfl_down(m, d));
    if (lane == 0) shm[wave] = m;
    __syncthreads();
    if (tid == 0) {
        double bm = INFINITY;
        for (int w = 0; w < kQThreads / kWave; ++w) bm = fmin(bm, shm[w]);
        ws->blk_min[blk] = bm;
    }
}

__global__ void __launch_bounds__(kQThreads)
q_final_kernel(const unsigned long long *hist, const double *__restrict__ ptable, int L, int lo,
               int hi, int min_val, double threshold, int on_qvalue, const QWork *__restrict__ ws,
               double *qtable, int *__restrict__ cutoff_out, unsigned long long *__restrict__ nrows_out,
               unsigned long long *__restrict__ clear)
{
    __shared__ unsigned long long sh[kQThreads / kWave];
    __shared__ double shm[kQThreads / kWave];
    __shared__ int first_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = gridDim.x, blk = blockIdx.x;
    if (tid == 0) first_s = L;
    unsigned long long t = 0;
    double below = INFINITY, all = INFINITY;
    for (int b = tid; b < nblk; b += kQThreads) {
        t += ws->blk_cnt[b];
        const double v = ws->blk_min[b];
        all = fmin(all, v);
        if (b < blk) below = fmin(below, v);
    }
    const unsigned long long n_rows_N = ws->n_rows_N;
    const unsigned long long n = block_sum_u64(t, sh) + n_rows_N;
    const double nd = (double)n;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        below = fmin(below, __shfl_down(below, d));
        all = fmin(all, __shfl_down(all, d));
    }
    __shared__ double shb[kQThreads / kWave], sha[kQThreads / kWave];
    if (lane == 0) { shb[wave] = below; sha[wave] = all; }
    __syncthreads();
    below = INFINITY; all = INFINITY;
#pragma unroll
    for (int w = 0; w < kQThreads / kWave; ++w) { below = fmin(below, shb[w]); all = fmin(all, sha[w]); }
    // rows holding an N sit below every other score: rank n, p = p_table[min_val] (= 1)
    const double base = n_rows_N ? ptable[min_val] / (nd / nd) : INFINITY;
    const int j = lo + blk * kQThreads + tid;
    const bool ok = j <= hi;
    double ms = ok ? qtable[j] : INFINITY;   // raw value left by q_raw_kernel
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const double v = __shfl_up(ms, d);
        if (lane >= d) ms = fmin(ms, v);
    }
    if (lane == kWave - 1) shm[wave] = ms;
    __syncthreads();
    double waves_below = INFINITY;
#pragma unroll
    for (int w = 0; w < kQThreads / kWave; ++w)
        if (w < wave) waves_below = fmin(waves_below, shm[w]);
    const double q = fmin(fmin(fmin(ms, waves_below), fmin(below, base)), 1.0);
    if (ok) {
        qtable[j] = q;
        const double val = on_qvalue ? q : ptable[j];
        if (val < threshold) atomicMin(&first_s, j);
        if (clear) clear[j] = 0ull;
    }
    // outside the window: 1 below it (p = 1 there), the last running minimum above it
    const int gtid = blk * kQThreads + tid, gsz = nblk * kQThreads;
    const double q_above = fmin(fmin(base, all), 1.0);
    for (int jj = gtid; jj < lo; jj += gsz) qtable[jj] = fmin(base, 1.0);
    for (int jj = hi + 1 + gtid; jj < L; jj += gsz) qtable[jj] = q_above;
    __syncthreads();
    if (tid == 0) {
        if (cutoff_out && first_s < L) atomicMin(cutoff_out, first_s);
        if (blk == 0) {
            if (nrows_out) *nrows_out = n;
            const bool n_outside = min_val < lo || min_val > hi;
            if (clear && n_outside) clear[min_val] = 0ull;
        }
    }
}

}  // namespace

// =======================================================================================
// host side
struct gfm_motif {
    int W = 0, L = 0, min_val = 0, scale = 1, ndw = 0;
    double offset = 0.0;
    int lo = 0, hi = 0, nb = 0;      // reachable score range [lo, hi], nb bins
    int hlo = 0, hnb = 0;            // LDS histogram window of a single-motif launch (== lo, nb when it fits)
    struct Window { int bins, lo; double mass; };
    std::vector<Window> windows;     // cache of best_window() per window size
    int device = 0;
    int n_cu = 256;
    int max_slabs = 0;
    int sel_slabs = 0;
    int reserve_cus = kReserveCUs;  // GRAFIMO_RESERVE_CUS overrides
    size_t lds_bytes = 0;
    std::vector<int64_t> sm;
    double bg[4] = {0, 0, 0, 0};
    std::vector<double> h_ptable;
    uint16_t *d_tab = nullptr;
    double *d_pmf = nullptr;
    double *d_ptable = nullptr;
    // Scoring workspace, double-buffered by call parity so that the post kernel of call k (on a
    // tail stream) may run while the score kernel of call k+1 fills the other set.
    unsigned *d_partials[2] = {nullptr, nullptr};   // [max_slabs][hnb+1] histogram slabs
    unsigned *d_spill[2] = {nullptr, nullptr};      // [nb] rows outside a partial window
    long long *d_resid[2] = {nullptr, nullptr};     // [max_slabs][kResidPerWG] residual hits
    int *d_resid_n[2] = {nullptr, nullptr};         // [max_slabs]
    QWork *d_qwork = nullptr;        // q-value kernels' block totals / minima
    double *d_qscratch = nullptr;    // [L] raw BH values when the caller wants no q-table
    HitCtl *d_ctl = nullptr;
    unsigned call_no = 0;           // score calls: HitCtl slot call_no % 3, workspace call_no % 2
    hipEvent_t ev_scored[2] = {nullptr, nullptr};   // score kernel of the last call of a parity done
    hipEvent_t ev_posted[2] = {nullptr, nullptr};   // its post kernel done (workspace free again)
    bool posted_valid[2] = {false, false};
    // separate workspace of gfm_select_hits (runs on the caller's tail stream, next to scoring)
    long long *d_sel_resid = nullptr;
    int *d_sel_resid_n = nullptr;
    HitCtl *d_sel_ctl = nullptr;
    unsigned sel_call_no = 0;
    // measurement aid: ring of event pairs around the score kernel
    std::vector<hipEvent_t> ev0, ev1;
    int ev_next = 0, ev_used = 0, ev_every = 1;
    unsigned ev_calls = 0;
};

namespace {

int ensure_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(GFM_ERR_NODEVICE, "no HIP device available (%s)",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    return GFM_OK;
}

// window of reachable scores after each DP position
void cumulative_windows(const int64_t *sm, int W, std::vector<int> &lo, std::vector<int> &hi)
{
    lo.resize(W);
    hi.resize(W);
    long long l = 0, h = 0;
    for (int j = 0; j < W; ++j) {
        int64_t mn = sm[j], mx = sm[j];
        for (int nuc = 1; nuc < 4; ++nuc) {
            mn = std::min(mn, sm[nuc * W + j]);
            mx = std::max(mx, sm[nuc * W + j]);
        }
        l += mn;
        h += mx;
        lo[j] = (int)l;
        hi[j] = (int)h;
    }
}

int validate_matrix(const int64_t *sm, int W)
{
    if (!sm) return fail(GFM_ERR_INVALID, "score matrix is NULL");
    if (W < 1 || W > GFM_MAX_WIDTH)
        return fail(GFM_ERR_INVALID, "motif width %d outside [1, %d]", W, GFM_MAX_WIDTH);
    for (int i = 0; i < 4 * W; ++i)
        if (sm[i] < 0 || sm[i] > kRange)
            return fail(GFM_ERR_INVALID, "scaled score %lld outside [0, %d]", (long long)sm[i], kRange);
    return GFM_OK;
}

// runs the DP for one motif on the current device; d_pmf receives L doubles
int run_dp(const int64_t *sm, int W, const double *bg, double *d_pmf, hipStream_t st)
{
    const int L = kRange * W + 1;
    std::vector<int> lo, hi, sm32(4 * W);
    cumulative_windows(sm, W, lo, hi);
    for (int i = 0; i < 4 * W; ++i) sm32[i] = (int)sm[i];
    DevBuf<int> d_sm, d_lo, d_hi;
    DevBuf<double> d_bg, d_buf;
    HIP_TRY(d_sm.alloc(4 * (size_t)W));
    HIP_TRY(d_lo.alloc(W));
    HIP_TRY(d_hi.alloc(W));
    HIP_TRY(d_bg.alloc(4));
    HIP_TRY(d_buf.alloc(2 * (size_t)L));
    HIP_TRY(hipMemcpyAsync(d_sm, sm32.data(), sizeof(int) * 4 * W, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_lo, lo.data(), sizeof(int) * W, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_hi, hi.data(), sizeof(int) * W, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_bg, bg, sizeof(double) * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(pvalue_dp_kernel, dim3(1), dim3(kDpThreads), 0, st, d_sm.p, d_bg.p, W, L, d_lo.p,
                       d_hi.p, d_buf.p, d_pmf);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));   // the temporaries are freed on return
    return GFM_OK;
}

// LDS bytes of one score launch: MM table sets | 16 wave strips (chunk + MM hit queues) | the
// LDS histogram windows of the motifs that use one
size_t score_lds_bytes(int W, int ndw, int mm, const int *nb_lds /* nb+1 or 0 per motif */)
{
    // one motif: uint16 pair tables; 2-3 motifs: ONE table of packed 64-bit entries (4x the bytes)
    size_t b = (size_t)(mm == 1 ? 1 : 4) * (2 * ndw * 64 * 2) + (size_t)waves_for_ndw(ndw) * stage_stride_bytes(W, mm);
    for (int i = 0; i < mm; ++i) b += sizeof(unsigned) * (size_t)nb_lds[i];
    return b;
}

template <int NDW, int MM>
int launch_score_t(gfm_motif *timer, const uint8_t *d_kmers, long long n, int W, long long row_base,
                   const ScoreArgs<MM> &args, size_t lds, int nslabs, hipStream_t st, bool prepare_only)
{
    auto kern = score_hist_kernel<NDW, MM>;
    if (prepare_only) {  // once per width from gfm_motif_create (never inside a stream capture)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes));
        return GFM_OK;
    }
    const bool prof = timer && !timer->ev0.empty() && (timer->ev_calls++ % (unsigned)timer->ev_every) == 0;
    const int slot = prof ? timer->ev_next : 0;
    if (prof) HIP_TRY(hipEventRecord(timer->ev0[slot], st));
    hipLaunchKernelGGL(kern, dim3(nslabs), dim3(waves_for_ndw(NDW) * kWave), lds, st, d_kmers, n, W, row_base, args);
    HIP_TRY(hipGetLastError());
    if (prof) {
        HIP_TRY(hipEventRecord(timer->ev1[slot], st));
        timer->ev_next = (slot + 1) % (int)timer->ev0.size();
        timer->ev_used = std::min(timer->ev_used + 1, (int)timer->ev0.size());
    }
    return GFM_OK;
}

template <int MM>
int dispatch_score(int ndw, gfm_motif *timer, const uint8_t *d_kmers, long long n, int W,
                   long long row_base, const ScoreArgs<MM> &args, size_t lds, int nslabs,
                   hipStream_t st, bool prepare_only)
{
#define GFM_CASE(N)                                                                               \
    case N:                                                                                       \
        return launch_score_t<N, MM>(timer, d_kmers, n, W, row_base, args, lds, nslabs, st, prepare_only);
    switch (ndw) {
        GFM_CASE(1) GFM_CASE(2) GFM_CASE(3) GFM_CASE(4) GFM_CASE(5) GFM_CASE(6) GFM_CASE(7) GFM_CASE(8)
        GFM_CASE(9) GFM_CASE(10) GFM_CASE(11) GFM_CASE(12) GFM_CASE(13) GFM_CASE(14) GFM_CASE(15) GFM_CASE(16)
        default: return fail(GFM_ERR_INVALID, "unsupported width %d", W);
    }
#undef GFM_CASE
}

// The `bins` consecutive scores that hold the most background probability
// (P(s >= a) - P(s >= a + bins) from the tail table): where a partial LDS histogram window goes.
gfm_motif::Window best_window(gfm_motif *m, int bins)
{
    if (bins >= m->nb) return {m->nb, m->lo, 1.0};
    for (const auto &w : m->windows)
        if (w.bins == bins) return w;
    gfm_motif::Window best{bins, m->lo, -1.0};
    for (int a = m->lo; a + bins - 1 <= m->hi; ++a) {
        const int e = a + bins;
        const double mass = m->h_ptable[a] - (e < m->L ? m->h_ptable[e] : 0.0);
        if (mass > best.mass) { best.mass = mass; best.lo = a; }
    }
    m->windows.push_back(best);
    return best;
}

// one launch after a scoring / selection kernel: histogram slabs -> hist64, hit slabs -> list
int launch_post(gfm_motif *m, const unsigned *partials, int hist_slabs, unsigned long long *d_hist,
                int win_lo, int win_nb, unsigned *spill, const long long *resid, const int *resid_n, int hit_slabs, HitCtl *ctl, int ctl_slot,
                long long *d_hit_rows, long long cap, unsigned long long *d_hit_count, hipStream_t st)
{
    const int bin_blocks = (win_nb + 1 + 255) / 256;
    const int groups = (hist_slabs + kSlabsPerGroup - 1) / kSlabsPerGroup;
    const int hist_blocks = d_hist ? bin_blocks * groups : 0;
    const int spill_blocks = (d_hist && spill && win_nb < m->nb) ? (m->nb + 255) / 256 : 0;
    const int total = hist_blocks + spill_blocks + hit_slabs;
    if (total == 0) return GFM_OK;
    hipLaunchKernelGGL(post_kernel, dim3(total), dim3(256), 0, st, partials, hist_slabs, win_nb, win_lo,
                       m->min_val, d_hist, bin_blocks, hist_blocks, spill, m->lo, m->nb, spill_blocks,
                       resid, resid_n, hit_slabs, ctl, ctl_slot, d_hit_rows, cap, d_hit_count);
    HIP_TRY(hipGetLastError());
    return GFM_OK;
}

void fill_motif_args(MotifArgs &a, gfm_motif *m, int ws, int slot, int use_hist, int win_lo, int win_nb,
                     int cutoff,
                     int *d_scores, long long *d_hit_rows, long long cap,
                     const unsigned long long *d_hit_count_or_null)
{
    a.tab = m->d_tab;
    a.lo = win_lo;
    a.nb = win_nb;
    a.min_val = m->min_val;
    a.use_hist = use_hist;
    a.spill_lo = m->lo;
    a.spill = m->d_spill[ws];
    a.cutoff = cutoff;
    a.slot = slot;
    a.scores = d_scores;
    a.partials = m->d_partials[ws];
    a.hit_rows = d_hit_rows;
    a.hit_cap = cap;
    a.hit_count = d_hit_count_or_null;
    a.ctl = m->d_ctl;
    a.resid = m->d_resid[ws];
    a.resid_n = m->d_resid_n[ws];
}

}  // namespace

// --------------------------------------------------------------------------------------- API
extern "C" __attribute__((visibility("hidden"))) void gfm_set_error_(const char *msg) { g_err = msg ? msg : ""; }

GFM_API int gfm_abi_version(void) { return GFM_ABI_VERSION; }
GFM_API const char *gfm_last_error(void) { return g_err.c_str(); }

GFM_API int gfm_device_count(int *count)
{
    if (!count) return fail(GFM_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return GFM_OK;
}

GFM_API int gfm_set_device(int ordinal)
{
    int rc = ensure_device();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ordinal));
    return GFM_OK;
}

GFM_API int gfm_compute_log_odds(const double *probs, int W, const double *bg, double *out)
{
    if (!probs || !bg || !out) return fail(GFM_ERR_INVALID, "NULL argument");
    if (W <= 0) return fail(GFM_ERR_INVALID, "Forbidden motif width.");
    double totBG = 0.0, totFG = 0.0;
    for (int n = 0; n < 4; ++n) {
        if (!(bg[n] > 0)) return fail(GFM_ERR_ASSERT, "assert bg > 0 (motif_processing.pyx:496)");
        totBG += bg[n];
        for (int j = 0; j < W; ++j) {
            const double p = probs[n * W + j];
            if (!(p > 0)) return fail(GFM_ERR_ASSERT, "assert prob > 0 (motif_processing.pyx:500)");
            totFG += p;
            const double odds = p / bg[n];
            out[n * W + j] = std::log(odds) * kLogFactor;
        }
    }
    if (!(totBG - 1.0 < 0.001)) return fail(GFM_ERR_ASSERT, "assert totBG - 1.0 < epsilon (motif_processing.pyx:505)");
    if (!(totFG - (double)W < 0.001)) return fail(GFM_ERR_ASSERT, "assert totFG - width < epsilon (motif_processing.pyx:506)");
    return GFM_OK;
}

GFM_API int gfm_scale_pwm(const double *lo, int W, int64_t *sm, int *min_val, int *max_val,
                          int *scale, double *offset)
{
    if (!lo || !sm || !min_val || !max_val || !scale || !offset)
        return fail(GFM_ERR_INVALID, "NULL argument");
    if (W <= 0) return fail(GFM_ERR_INVALID, "Forbidden motif width.");
    double lower = lo[0], upper = lo[0];
    for (int i = 1; i < 4 * W; ++i) {
        lower = std::min(lower, lo[i]);
        upper = std::max(upper, lo[i]);
    }
    if (lower == upper) lower = upper - 1.0;
    lower = std::floor(lower);
    const double off = std::nearbyint(std::floor(lower));
    const double sf = std::floor((double)kRange / (upper - lower));
    int64_t mn = 0, mx = 0;
    for (int i = 0; i < 4 * W; ++i) {
        sm[i] = (int64_t)std::nearbyint((lo[i] - off) * sf);  // half-to-even like np.round
        if (i == 0 || sm[i] < mn) mn = sm[i];
        if (i == 0 || sm[i] > mx) mx = sm[i];
    }
    *min_val = (int)mn;
    *max_val = (int)mx;
    *scale = (int)sf;
    *offset = off;
    return GFM_OK;
}

GFM_API int gfm_comp_pval_mat(const int64_t *sm, int W, const double *bg, double *h_pmf)
{
    if (!bg || !h_pmf) return fail(GFM_ERR_INVALID, "NULL argument");
    int rc = validate_matrix(sm, W);
    if (rc) return rc;
    for (int n = 0; n < 4; ++n)
        if (!(bg[n] > 0)) return fail(GFM_ERR_ASSERT, "assert bg > 0 (motif_processing.pyx:592)");
    rc = ensure_device();
    if (rc) return rc;
    const int L = kRange * W + 1;
    DevBuf<double> d_pmf;
    HIP_TRY(d_pmf.alloc((size_t)L));
    rc = run_dp(sm, W, bg, d_pmf, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(h_pmf, d_pmf, sizeof(double) * (size_t)L, hipMemcpyDeviceToHost));
    return GFM_OK;
}

GFM_API void gfm_motif_destroy(gfm_motif_t m)
{
    if (!m) return;
    if (m->d_tab) (void)hipFree(m->d_tab);
    if (m->d_pmf) (void)hipFree(m->d_pmf);
    if (m->d_ptable) (void)hipFree(m->d_ptable);
    for (int i = 0; i < 2; ++i) {
        if (m->d_partials[i]) (void)hipFree(m->d_partials[i]);
        if (m->d_spill[i]) (void)hipFree(m->d_spill[i]);
        if (m->d_resid[i]) (void)hipFree(m->d_resid[i]);
        if (m->d_resid_n[i]) (void)hipFree(m->d_resid_n[i]);
        if (m->ev_scored[i]) (void)hipEventDestroy(m->ev_scored[i]);
        if (m->ev_posted[i]) (void)hipEventDestroy(m->ev_posted[i]);
    }
    if (m->d_ctl) (void)hipFree(m->d_ctl);
    if (m->d_qwork) (void)hipFree(m->d_qwork);
    if (m->d_qscratch) (void)hipFree(m->d_qscratch);
    if (m->d_sel_resid) (void)hipFree(m->d_sel_resid);
    if (m->d_sel_resid_n) (void)hipFree(m->d_sel_resid_n);
    if (m->d_sel_ctl) (void)hipFree(m->d_sel_ctl);
    for (auto e : m->ev0) (void)hipEventDestroy(e);
    for (auto e : m->ev1) (void)hipEventDestroy(e);
    delete m;
}

GFM_API int gfm_motif_create(const int64_t *sm, int W, const double *bg, int min_val, int scale,
                             double offset, const double *h_pmf, gfm_motif_t *out)
{
    if (!out || !bg) return fail(GFM_ERR_INVALID, "NULL argument");
    *out = nullptr;
    int rc = validate_matrix(sm, W);
    if (rc) return rc;
    if (scale <= 0) return fail(GFM_ERR_INVALID, "scale must be a positive integer");
    for (int n = 0; n < 4; ++n)
        if (!(bg[n] > 0)) return fail(GFM_ERR_ASSERT, "assert bg > 0");
    rc = ensure_device();
    if (rc) return rc;

    gfm_motif *m = new (std::nothrow) gfm_motif();
    if (!m) return fail(GFM_ERR_NOMEM, "out of host memory");
    m->W = W;
    m->L = kRange * W + 1;
    m->min_val = min_val;
    m->scale = scale;
    m->offset = offset;
    m->ndw = (W + 3) / 4;
    m->sm.assign(sm, sm + 4 * W);
    std::memcpy(m->bg, bg, sizeof m->bg);
    std::vector<int> clo, chi;
    cumulative_windows(sm, W, clo, chi);
    m->lo = clo[W - 1];
    m->hi = chi[W - 1];
    m->nb = m->hi - m->lo + 1;

    auto bail = [&](int code) { gfm_motif_destroy(m); return code; };
#define HIP_TRY_M(expr)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return bail(fail(GFM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)));   \
    } while (0)

    HIP_TRY_M(hipGetDevice(&m->device));
    hipDeviceProp_t prop;
    HIP_TRY_M(hipGetDeviceProperties(&prop, m->device));
    m->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char *e = std::getenv("GRAFIMO_RESERVE_CUS")) m->reserve_cus = std::max(0, std::min(atoi(e), m->n_cu - 1));

    // LDS lookup tables: [2*ndw base pairs][8 x 8 codes] uint16, code = (ascii >> 1) & 7
    // (A 0, C 1, T 2, G 3; 4..7 invalid), index = code(first) + 8 * code(second).
    // A valid pair holds sm[first][2p] + sm[second][2p+1]; a pair with an invalid code holds
    // kPoison; a position >= W contributes 0 and accepts any code.
    std::vector<uint16_t> tab((size_t)2 * m->ndw * 64, 0);
    auto base_score = [&](int pos, int code, bool *bad) -> unsigned {
        if (pos >= W) return 0u;
        static const int nuc_of_code[4] = {0, 1, 3, 2};  // code 2 = T (row 3), code 3 = G (row 2)
        if (code > 3) { *bad = true; return 0u; }
        return (unsigned)sm[nuc_of_code[code] * W + pos];
    };
    for (int pr = 0; pr < 2 * m->ndw; ++pr)
        for (int c1 = 0; c1 < 8; ++c1)
            for (int c0 = 0; c0 < 8; ++c0) {
                bool bad = false;
                const unsigned v = base_score(2 * pr, c0, &bad) + base_score(2 * pr + 1, c1, &bad);
                tab[(size_t)pr * 64 + c0 + 8 * c1] = (uint16_t)(bad ? kPoison : v);
            }
    HIP_TRY_M(hipMalloc(&m->d_tab, tab.size() * sizeof(uint16_t)));
    HIP_TRY_M(hipMemcpy(m->d_tab, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice));

    HIP_TRY_M(hipMalloc(&m->d_pmf, sizeof(double) * (size_t)m->L));
    HIP_TRY_M(hipMalloc(&m->d_ptable, sizeof(double) * (size_t)m->L));
    if (h_pmf) {
        HIP_TRY_M(hipMemcpy(m->d_pmf, h_pmf, sizeof(double) * (size_t)m->L, hipMemcpyHostToDevice));
    } else {
        rc = run_dp(sm, W, bg, m->d_pmf, nullptr);
        if (rc) return bail(rc);
    }
    hipLaunchKernelGGL(ptable_kernel, dim3(1), dim3(kScanThreads), 0, nullptr, m->d_pmf, m->L,
                       m->lo, m->hi, m->d_ptable);
    HIP_TRY_M(hipGetLastError());
    m->h_ptable.resize(m->L);
    HIP_TRY_M(hipMemcpy(m->h_ptable.data(), m->d_ptable, sizeof(double) * (size_t)m->L,
                        hipMemcpyDeviceToHost));

    // score-kernel LDS plan: table | 8 wave strips | histogram window (+1 N bin)
    const int zero_nb = 0;
    const size_t fixed = score_lds_bytes(W, m->ndw, 1, &zero_nb);
    const long long room = ((long long)kMaxLdsBytes - (long long)fixed) / (long long)sizeof(unsigned) - 1;
    if (room < 256) return bail(fail(GFM_ERR_INVALID, "no LDS left for a histogram window at width %d", W));
    m->hnb = (int)std::min<long long>(m->nb, room);
    m->hlo = best_window(m, m->hnb).lo;   // partial when the range does not fit: the rest spills
    m->lds_bytes = fixed + sizeof(unsigned) * (size_t)(m->hnb + 1);
    int per_cu = (int)std::min<size_t>(kWGsPerCU, (size_t)kMaxLdsBytes / m->lds_bytes);
    per_cu = std::max(per_cu, 1);
    m->max_slabs = m->n_cu * per_cu;
    for (int i = 0; i < 2; ++i) {
        HIP_TRY_M(hipMalloc(&m->d_partials[i], sizeof(unsigned) * (size_t)m->max_slabs * (size_t)(m->hnb + 1)));
        HIP_TRY_M(hipMalloc(&m->d_spill[i], sizeof(unsigned) * (size_t)m->nb));
        HIP_TRY_M(hipMemset(m->d_spill[i], 0, sizeof(unsigned) * (size_t)m->nb));
        HIP_TRY_M(hipMalloc(&m->d_resid[i], sizeof(long long) * (size_t)m->max_slabs * kResidPerWG));
        HIP_TRY_M(hipMalloc(&m->d_resid_n[i], sizeof(int) * (size_t)m->max_slabs));
        HIP_TRY_M(hipEventCreateWithFlags(&m->ev_scored[i], hipEventDisableTiming | hipEventReleaseToDevice));
        HIP_TRY_M(hipEventCreateWithFlags(&m->ev_posted[i], hipEventDisableTiming | hipEventReleaseToDevice));
    }
    HIP_TRY_M(hipMalloc(&m->d_ctl, sizeof(HitCtl)));
    HIP_TRY_M(hipMemset(m->d_ctl, 0, sizeof(HitCtl)));
    HIP_TRY_M(hipMalloc(&m->d_qwork, sizeof(QWork)));
    HIP_TRY_M(hipMalloc(&m->d_qscratch, sizeof(double) * (size_t)m->L));
    m->sel_slabs = 4 * m->n_cu;
    HIP_TRY_M(hipMalloc(&m->d_sel_resid, sizeof(long long) * (size_t)m->sel_slabs * kResidPerWG));
    HIP_TRY_M(hipMalloc(&m->d_sel_resid_n, sizeof(int) * (size_t)m->sel_slabs));
    HIP_TRY_M(hipMalloc(&m->d_sel_ctl, sizeof(HitCtl)));
    HIP_TRY_M(hipMemset(m->d_sel_ctl, 0, sizeof(HitCtl)));
#undef HIP_TRY_M
    {   // allow up to the whole LDS for every instantiation this width can use
        ScoreArgs<1> a1{};
        ScoreArgs<2> a2{};
        ScoreArgs<3> a3{};
        rc = dispatch_score<1>(m->ndw, nullptr, nullptr, 0, W, 0, a1, 0, 1, nullptr, true);
        if (!rc) rc = dispatch_score<2>(m->ndw, nullptr, nullptr, 0, W, 0, a2, 0, 1, nullptr, true);
        if (!rc) rc = dispatch_score<3>(m->ndw, nullptr, nullptr, 0, W, 0, a3, 0, 1, nullptr, true);
        if (rc) return bail(rc);
    }
    *out = m;
    return GFM_OK;
}

GFM_API int gfm_motif_width(gfm_motif_t m) { return m ? m->W : 0; }
GFM_API int gfm_motif_table_len(gfm_motif_t m) { return m ? m->L : 0; }

GFM_API int gfm_motif_score_range(gfm_motif_t m, int32_t *lo, int32_t *hi)
{
    if (!m) return fail(GFM_ERR_INVALID, "motif is NULL");
    if (lo) *lo = m->lo;
    if (hi) *hi = m->hi;
    return GFM_OK;
}

GFM_API int gfm_motif_tables(gfm_motif_t m, double *h_pmf, double *h_ptable)
{
    if (!m) return fail(GFM_ERR_INVALID, "motif is NULL");
    if (h_pmf) HIP_TRY(hipMemcpy(h_pmf, m->d_pmf, sizeof(double) * (size_t)m->L, hipMemcpyDeviceToHost));
    if (h_ptable) std::memcpy(h_ptable, m->h_ptable.data(), sizeof(double) * (size_t)m->L);
    return GFM_OK;
}

GFM_API int gfm_motif_pvalue_cutoff(gfm_motif_t m, double threshold, int32_t *cutoff)
{
    if (!m || !cutoff) return fail(GFM_ERR_INVALID, "NULL argument");
    // p_table is non-increasing in s: first index with p < threshold
    int a = 0, b = m->L;
    while (a < b) {
        const int mid = (a + b) / 2;
        if (m->h_ptable[mid] < threshold) b = mid; else a = mid + 1;
    }
    *cutoff = a;
    return GFM_OK;
}

GFM_API int gfm_motif_annotate(gfm_motif_t m, const int32_t *scores, int64_t n, double *lo_out,
                               double *p_out)
{
    if (!m || (!scores && n)) return fail(GFM_ERR_INVALID, "NULL argument");
    for (int64_t i = 0; i < n; ++i) {
        const int s = scores[i];
        if (s < 0 || s >= m->L) return fail(GFM_ERR_INVALID, "score %d outside the table", s);
        if (lo_out) lo_out[i] = ((double)s / (double)m->scale) + ((double)m->W * m->offset);
        if (p_out) p_out[i] = m->h_ptable[s];
    }
    return GFM_OK;
}

GFM_API int gfm_score_kmers(gfm_motif_t m, const uint8_t *d_kmers, int64_t n, int32_t *d_scores,
                            uint64_t *d_hist, int32_t select_cutoff, int64_t row_base,
                            int64_t *d_hit_rows, int64_t hit_capacity, uint64_t *d_hit_count,
                            uint32_t flags, void *stream, void *tail_stream)
{
    if (!m) return fail(GFM_ERR_INVALID, "motif is NULL");
    if (n < 0) return fail(GFM_ERR_INVALID, "negative row count");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipStream_t tail = tail_stream ? static_cast<hipStream_t>(tail_stream) : st;
    const bool split = tail != st;
    if (n == 0) {
        if ((flags & GFM_FLAG_RESET_HITS) && d_hit_count) {
            if (split) {  // keep the tail stream ordered behind what the main stream holds
                HIP_TRY(hipEventRecord(m->ev_scored[0], st));
                HIP_TRY(hipStreamWaitEvent(tail, m->ev_scored[0], 0));
            }
            HIP_TRY(hipMemsetAsync(d_hit_count, 0, sizeof(uint64_t), tail));
        }
        return GFM_OK;
    }
    if (!d_kmers || !d_scores) return fail(GFM_ERR_INVALID, "NULL device buffer");
    if ((reinterpret_cast<uintptr_t>(d_kmers) & 15u) != 0)
        return fail(GFM_ERR_INVALID, "d_kmers must be 16-byte aligned");
    if (n > (int64_t)kChunk * 0x7fffff00ll)
        return fail(GFM_ERR_INVALID, "too many rows for one launch (split the batch)");
    const bool select = select_cutoff != GFM_NO_SELECT;
    if (select && (!d_hit_rows || !d_hit_count))
        return fail(GFM_ERR_INVALID, "selection requested without hit buffers");
    const int use_hist = d_hist ? 1 : 0;
    const long long nchunks = (n + kChunk - 1) / kChunk;
    const long long want = (nchunks + waves_for_ndw(m->ndw) - 1) / waves_for_ndw(m->ndw);
    // with a tail stream a few CUs are left free so that its kernels (post, q-table, RCCL) find
    // room without evicting a persistent score workgroup (which would delay the whole grid)
    const int avail = split ? std::max(1, m->max_slabs - m->reserve_cus * (m->max_slabs / m->n_cu)) : m->max_slabs;
    const int nslabs = (int)std::min<long long>(want, avail);

    const unsigned k = m->call_no++;
    const int ws = (int)(k & 1u), slot = (int)(k % 3u);
    const bool reset = (flags & GFM_FLAG_RESET_HITS) != 0;
    // workspace `ws` was last used by call k-2: its post kernel must be done.  Appending to a
    // hit list additionally needs the count published by call k-1.
    if (m->posted_valid[ws] && !(flags & GFM_FLAG_CALLER_ORDERS_REUSE))
        HIP_TRY(hipStreamWaitEvent(st, m->ev_posted[ws], 0));
    if (split && select && !reset && m->posted_valid[ws ^ 1])
        HIP_TRY(hipStreamWaitEvent(st, m->ev_posted[ws ^ 1], 0));
    ScoreArgs<1> args{};
    fill_motif_args(args.m[0], m, ws, slot, use_hist, m->hlo, m->hnb, select_cutoff, d_scores,
                    reinterpret_cast<long long *>(d_hit_rows), hit_capacity,
                    reset ? nullptr : reinterpret_cast<unsigned long long *>(d_hit_count));
    int rc = dispatch_score<1>(m->ndw, m, d_kmers, n, m->W, row_base, args, m->lds_bytes, nslabs, st, false);
    if (rc) return rc;
    if (split) {
        HIP_TRY(hipEventRecord(m->ev_scored[ws], st));
        HIP_TRY(hipStreamWaitEvent(tail, m->ev_scored[ws], 0));
    }
    rc = launch_post(m, m->d_partials[ws], nslabs, reinterpret_cast<unsigned long long *>(d_hist),
                     m->hlo, m->hnb, m->d_spill[ws], m->d_resid[ws], m->d_resid_n[ws], select ? nslabs : 0, m->d_ctl, slot,
                     reinterpret_cast<long long *>(d_hit_rows), hit_capacity,
                     reinterpret_cast<unsigned long long *>(d_hit_count), tail);
    if (rc) return rc;
    if (split) {
        HIP_TRY(hipEventRecord(m->ev_posted[ws], tail));
        m->posted_valid[ws] = true;
    } else {
        m->posted_valid[ws] = false;  // same stream: plain stream order already protects the workspace
    }
    return GFM_OK;
}

GFM_API int gfm_score_kmers_multi(const gfm_motif_t *motifs, int n_motifs, const uint8_t *d_kmers,
                                  int64_t n, int32_t *const *d_scores, uint64_t *const *d_hist,
                                  const int32_t *select_cutoffs, int64_t row_base,
                                  int64_t *const *d_hit_rows, const int64_t *hit_capacity,
                                  uint64_t *const *d_hit_count, uint32_t flags, void *stream)
{
    if (!motifs || n_motifs < 1 || !d_scores) return fail(GFM_ERR_INVALID, "NULL argument");
    if (n < 0) return fail(GFM_ERR_INVALID, "negative row count");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool reset = (flags & GFM_FLAG_RESET_HITS) != 0;
    for (int i = 0; i < n_motifs; ++i) {
        if (!motifs[i]) return fail(GFM_ERR_INVALID, "motif %d is NULL", i);
        if (motifs[i]->W != motifs[0]->W)
            return fail(GFM_ERR_INVALID, "motifs of one batched launch must share their width (%d vs %d)",
                        motifs[i]->W, motifs[0]->W);
        for (int j = 0; j < i; ++j)
            if (motifs[j] == motifs[i]) return fail(GFM_ERR_INVALID, "motif %d listed twice", i);
        const bool sel = select_cutoffs && select_cutoffs[i] != GFM_NO_SELECT;
        if (sel && (!d_hit_rows || !d_hit_rows[i] || !d_hit_count || !d_hit_count[i] || !hit_capacity))
            return fail(GFM_ERR_INVALID, "selection requested for motif %d without hit buffers", i);
        if (!d_scores[i]) return fail(GFM_ERR_INVALID, "d_scores[%d] is NULL", i);
    }
    if (n == 0) {
        if (reset && d_hit_count)
            for (int i = 0; i < n_motifs; ++i)
                if (d_hit_count[i]) HIP_TRY(hipMemsetAsync(d_hit_count[i], 0, sizeof(uint64_t), st));
        return GFM_OK;
    }
    if (!d_kmers) return fail(GFM_ERR_INVALID, "NULL device buffer");
    if ((reinterpret_cast<uintptr_t>(d_kmers) & 15u) != 0)
        return fail(GFM_ERR_INVALID, "d_kmers must be 16-byte aligned");
    if (n > (int64_t)kChunk * 0x7fffff00ll)
        return fail(GFM_ERR_INVALID, "too many rows for one launch (split the batch)");
    const int W = motifs[0]->W, ndw = motifs[0]->ndw;
    const long long nchunks = (n + kChunk - 1) / kChunk;
    const long long want = (nchunks + waves_for_ndw(ndw) - 1) / waves_for_ndw(ndw);

    // grouping: the largest group (<= 3 motifs) whose LDS histogram windows still hold
    // kMinWindowMass of each motif's background score distribution.  Windows share what the tables
    // and strips leave: a motif whose whole range fits takes it, the others split the rest and
    // spill the rows outside their window (partial windows, see score_hist_kernel).
    constexpr double kMinWindowMass = 0.999;
    int i = 0;
    while (i < n_motifs) {
        int mm = std::min(3, n_motifs - i), nb_lds[3] = {0, 0, 0};
        int win_lo[3] = {0, 0, 0}, win_nb[3] = {0, 0, 0};
        for (; mm >= 1; --mm) {
            const int zero[3] = {0, 0, 0};
            long long room = ((long long)kMaxLdsBytes - (long long)score_lds_bytes(W, ndw, mm, zero)) /
                             (long long)sizeof(unsigned);
            bool open[3] = {false, false, false};
            int users = 0;
            for (int k = 0; k < mm; ++k) {
                win_nb[k] = 0;
                win_lo[k] = motifs[i + k]->lo;
                open[k] = d_hist && d_hist[i + k];
                users += open[k];
            }
            bool ok = room > 0 || users == 0;
            // water-filling: ranges that fit their equal share are served whole, the rest share again
            for (bool again = true; ok && again && users > 0;) {
                again = false;
                const long long share = room / users - 1;
                for (int k = 0; k < mm; ++k)
                    if (open[k] && motifs[i + k]->nb <= share) {
                        win_nb[k] = motifs[i + k]->nb;
                        room -= win_nb[k] + 1;
                        open[k] = false;
                        --users;
                        again = true;
                    }
            }
            if (ok && users > 0) {
                const long long share = room / users - 1;
                if (share < 256) ok = false;
                for (int k = 0; ok && k < mm; ++k)
                    if (open[k]) {
                        const gfm_motif::Window w = best_window(motifs[i + k], (int)share);
                        win_nb[k] = w.bins;
                        win_lo[k] = w.lo;
                        if (mm > 1 && w.mass < kMinWindowMass) ok = false;
                    }
            }
            if (ok || mm == 1) break;
        }
        if (mm < 1) mm = 1;
        for (int k = 0; k < mm; ++k) nb_lds[k] = (d_hist && d_hist[i + k]) ? win_nb[k] + 1 : 0;
        const size_t lds = score_lds_bytes(W, ndw, mm, nb_lds);
        int nslabs = (int)std::min<long long>(want, motifs[i]->max_slabs);
        for (int k = 1; k < mm; ++k) nslabs = std::min(nslabs, motifs[i + k]->max_slabs);
        int ws[3], slot[3], uh[3];
        ScoreArgs<1> a1{};
        ScoreArgs<2> a2{};
        ScoreArgs<3> a3{};
        for (int k = 0; k < mm; ++k) {
            gfm_motif *mo = motifs[i + k];
            const unsigned c = mo->call_no++;
            ws[k] = (int)(c & 1u);
            slot[k] = (int)(c % 3u);
            mo->posted_valid[ws[k]] = false;   // single stream: stream order protects the workspace
            uh[k] = (d_hist && d_hist[i + k]) ? 1 : 0;
            const int cut = select_cutoffs ? select_cutoffs[i + k] : GFM_NO_SELECT;
            const bool sel = cut != GFM_NO_SELECT;
            MotifArgs &dst = mm == 1 ? a1.m[k] : (mm == 2 ? a2.m[k] : a3.m[k]);
            fill_motif_args(dst, mo, ws[k], slot[k], uh[k], win_lo[k], win_nb[k], cut, d_scores[i + k],
                            sel ? reinterpret_cast<long long *>(d_hit_rows[i + k]) : nullptr,
                            sel ? hit_capacity[i + k] : 0,
                            (sel && !reset) ? reinterpret_cast<const unsigned long long *>(d_hit_count[i + k])
                                            : nullptr);
        }
        int rc;
        if (mm == 1) rc = dispatch_score<1>(ndw, motifs[i], d_kmers, n, W, row_base, a1, lds, nslabs, st, false);
        else if (mm == 2) rc = dispatch_score<2>(ndw, motifs[i], d_kmers, n, W, row_base, a2, lds, nslabs, st, false);
        else rc = dispatch_score<3>(ndw, motifs[i], d_kmers, n, W, row_base, a3, lds, nslabs, st, false);
        if (rc) return rc;
        for (int k = 0; k < mm; ++k) {
            gfm_motif *mo = motifs[i + k];
            const int cut = select_cutoffs ? select_cutoffs[i + k] : GFM_NO_SELECT;
            const bool sel = cut != GFM_NO_SELECT;
            rc = launch_post(mo, mo->d_partials[ws[k]], nslabs,
                             uh[k] ? reinterpret_cast<unsigned long long *>(d_hist[i + k]) : nullptr,
                             win_lo[k], win_nb[k], mo->d_spill[ws[k]], mo->d_resid[ws[k]], mo->d_resid_n[ws[k]], sel ? nslabs : 0, mo->d_ctl, slot[k],
                             sel ? reinterpret_cast<long long *>(d_hit_rows[i + k]) : nullptr,
                             sel ? hit_capacity[i + k] : 0,
                             sel ? reinterpret_cast<unsigned long long *>(d_hit_count[i + k]) : nullptr, st);
            if (rc) return rc;
        }
        i += mm;
    }
    return GFM_OK;
}

GFM_API int gfm_profile_enable(gfm_motif_t m, int slots, int every)
{
    if (!m || slots < 0 || every < 1) return fail(GFM_ERR_INVALID, "bad argument");
    m->ev_every = every;
    m->ev_calls = 0;
    for (auto e : m->ev0) (void)hipEventDestroy(e);
    for (auto e : m->ev1) (void)hipEventDestroy(e);
    m->ev0.clear();
    m->ev1.clear();
    m->ev_next = m->ev_used = 0;
    for (int i = 0; i < slots; ++i) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        m->ev0.push_back(a);
        m->ev1.push_back(b);
    }
    return GFM_OK;
}

GFM_API int gfm_profile_read(gfm_motif_t m, float *h_ms, int capacity, int *n_out)
{
    if (!m || !n_out || (!h_ms && capacity)) return fail(GFM_ERR_INVALID, "NULL argument");
    const int slots = (int)m->ev0.size();
    const int used = std::min(m->ev_used, capacity);
    for (int i = 0; i < used; ++i) {
        const int slot = ((m->ev_next - m->ev_used + i) % slots + slots) % slots;
        HIP_TRY(hipEventSynchronize(m->ev1[slot]));
        HIP_TRY(hipEventElapsedTime(&h_ms[i], m->ev0[slot], m->ev1[slot]));
    }
    *n_out = used;
    m->ev_used = 0;
    return GFM_OK;
}

GFM_API int gfm_qvalue_table(gfm_motif_t m, uint64_t *d_hist, double threshold, int on_qvalue,
                             double *d_qtable, int32_t *d_cutoff, uint64_t *d_nrows, uint32_t flags,
                             void *stream)
{
    if (!m || !d_hist) return fail(GFM_ERR_INVALID, "NULL argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned long long *hc = reinterpret_cast<const unsigned long long *>(d_hist);
    unsigned long long *clr =
        (flags & GFM_FLAG_CLEAR_HIST) ? reinterpret_cast<unsigned long long *>(d_hist) : nullptr;
    unsigned long long *nr = reinterpret_cast<unsigned long long *>(d_nrows);
    const int nblk = (m->nb + kQThreads - 1) / kQThreads;   // <= 251 for W <= 64
    hipLaunchKernelGGL(q_count_kernel, dim3(nblk), dim3(kQThreads), 0, st, hc, m->L, m->lo, m->hi,
                       m->min_val, m->d_qwork, d_cutoff);
    // q_raw leaves raw(s) in a table of L doubles: the caller's q-table, or ours when none is asked
    double *qt = d_qtable ? d_qtable : m->d_qscratch;
    hipLaunchKernelGGL(q_raw_kernel, dim3(nblk), dim3(kQThreads), 0, st, hc, m->d_ptable, m->lo, m->hi,
                       m->d_qwork, qt);
    hipLaunchKernelGGL(q_final_kernel, dim3(nblk), dim3(kQThreads), 0, st, hc, m->d_ptable, m->L, m->lo,
                       m->hi, m->min_val, threshold, on_qvalue, m->d_qwork, qt, d_cutoff, nr, clr);
    HIP_TRY(hipGetLastError());
    return GFM_OK;
}

GFM_API int gfm_select_hits(gfm_motif_t m, const int32_t *d_scores, int64_t n, const int32_t *d_cutoff,
                            int64_t row_base, int64_t *d_hit_rows, int64_t hit_capacity,
                            uint64_t *d_hit_count, uint32_t flags, void *stream)
{
    if (!m) return fail(GFM_ERR_INVALID, "motif is NULL");
    if (n < 0) return fail(GFM_ERR_INVALID, "negative row count");
    if (n == 0) {
        if ((flags & GFM_FLAG_RESET_HITS) && d_hit_count)
            HIP_TRY(hipMemsetAsync(d_hit_count, 0, sizeof(uint64_t), static_cast<hipStream_t>(stream)));
        return GFM_OK;
    }
    if (!d_scores || !d_cutoff || !d_hit_rows || !d_hit_count)
        return fail(GFM_ERR_INVALID, "NULL device buffer");
    if ((reinterpret_cast<uintptr_t>(d_scores) & 15u) != 0)
        return fail(GFM_ERR_INVALID, "d_scores must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long n4 = (n + 3) >> 2;
    long long blocks = (n4 + kSelThreads - 1) / kSelThreads;
    blocks = std::max<long long>(1, std::min<long long>(blocks, m->sel_slabs));
    const int slot = (int)(m->sel_call_no++ % 3u);
    hipLaunchKernelGGL(select_hits_kernel, dim3((unsigned)blocks), dim3(kSelThreads), 0, st, d_scores,
                       (long long)n, d_cutoff, (long long)row_base,
                       reinterpret_cast<long long *>(d_hit_rows), (long long)hit_capacity,
                       (flags & GFM_FLAG_RESET_HITS)
                           ? nullptr
                           : reinterpret_cast<const unsigned long long *>(d_hit_count),
                       m->d_sel_ctl, slot, m->d_sel_resid, m->d_sel_resid_n);
    HIP_TRY(hipGetLastError());
    return launch_post(m, nullptr, 0, nullptr, 0, 0, nullptr, m->d_sel_resid, m->d_sel_resid_n, (int)blocks,
                       m->d_sel_ctl, slot, reinterpret_cast<long long *>(d_hit_rows), hit_capacity,
                       reinterpret_cast<unsigned long long *>(d_hit_count), st);
}

GFM_API int gfm_scan_host(gfm_motif_t m, const uint8_t *h_kmers, int64_t n, double threshold,
                          int on_qvalue, int want_qvalues, int64_t capacity, int64_t *h_rows,
                          int32_t *h_scores, double *h_logodds, double *h_pvalue, double *h_qvalue,
                          int64_t *n_hits)
{
    if (!m || !n_hits) return fail(GFM_ERR_INVALID, "NULL argument");
    *n_hits = 0;
    if (n < 0 || capacity < 0) return fail(GFM_ERR_INVALID, "negative size");
    if (!(threshold > 0 && threshold <= 1)) return fail(GFM_ERR_INVALID, "threshold must be in (0, 1]");
    if (on_qvalue && !want_qvalues) return fail(GFM_ERR_INVALID, "q-value threshold without q-values");
    if (n == 0) return GFM_OK;
    if (!h_kmers) return fail(GFM_ERR_INVALID, "h_kmers is NULL");

    const size_t kbytes = (size_t)n * (size_t)m->W;
    uint8_t *d_kmers = nullptr;
    int *d_scores = nullptr, *d_cutoff = nullptr;
    unsigned long long *d_hist = nullptr, *d_count = nullptr;
    long long *d_rows = nullptr;
    double *d_q = nullptr;
    hipStream_t st = nullptr;
    int rc = GFM_OK;
    std::vector<long long> rows;
    std::vector<double> q;
    unsigned long long cnt = 0;
    const long long cap = std::max<long long>(capacity, 1);
    const bool need_hist = want_qvalues != 0;

#define SCAN_TRY(expr)                                                                  \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            rc = fail(GFM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));      \
            goto done;                                                                  \
        }                                                                               \
    } while (0)
#define SCAN_RC(expr)            \
    do {                         \
        rc = (expr);             \
        if (rc) goto done;       \
    } while (0)

    SCAN_TRY(hipStreamCreate(&st));
    SCAN_TRY(hipMalloc(&d_kmers, kbytes + 16));
    SCAN_TRY(hipMalloc(&d_scores, sizeof(int) * (size_t)n));
    SCAN_TRY(hipMalloc(&d_rows, sizeof(long long) * (size_t)cap));
    SCAN_TRY(hipMalloc(&d_count, sizeof(unsigned long long)));
    SCAN_TRY(hipMalloc(&d_cutoff, sizeof(int)));
    SCAN_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), st));
    if (need_hist) {
        SCAN_TRY(hipMalloc(&d_hist, sizeof(unsigned long long) * (size_t)m->L));
        SCAN_TRY(hipMalloc(&d_q, sizeof(double) * (size_t)m->L));
        SCAN_TRY(hipMemsetAsync(d_hist, 0, sizeof(unsigned long long) * (size_t)m->L, st));
    }
    SCAN_TRY(hipMemcpyAsync(d_kmers, h_kmers, kbytes, hipMemcpyHostToDevice, st));
    if (!on_qvalue) {
        int32_t cutoff = 0;
        SCAN_RC(gfm_motif_pvalue_cutoff(m, threshold, &cutoff));
        SCAN_RC(gfm_score_kmers(m, d_kmers, n, d_scores, reinterpret_cast<uint64_t *>(d_hist), cutoff,
                                0, reinterpret_cast<int64_t *>(d_rows), cap,
                                reinterpret_cast<uint64_t *>(d_count), GFM_FLAG_RESET_HITS, st, nullptr));
        if (need_hist)
            SCAN_RC(gfm_qvalue_table(m, reinterpret_cast<uint64_t *>(d_hist), threshold, 0, d_q,
                                     nullptr, nullptr, 0, st));
    } else {
        SCAN_RC(gfm_score_kmers(m, d_kmers, n, d_scores, reinterpret_cast<uint64_t *>(d_hist),
                                GFM_NO_SELECT, 0, nullptr, 0, nullptr, 0, st, nullptr));
        SCAN_RC(gfm_qvalue_table(m, reinterpret_cast<uint64_t *>(d_hist), threshold, 1, d_q, d_cutoff,
                                 nullptr, 0, st));
        SCAN_RC(gfm_select_hits(m, d_scores, n, d_cutoff, 0, reinterpret_cast<int64_t *>(d_rows), cap,
                                reinterpret_cast<uint64_t *>(d_count), GFM_FLAG_RESET_HITS, st));
    }
    SCAN_TRY(hipMemcpyAsync(&cnt, d_count, sizeof cnt, hipMemcpyDeviceToHost, st));
    SCAN_TRY(hipStreamSynchronize(st));
    *n_hits = (int64_t)cnt;
    if ((long long)cnt > capacity) {
        rc = fail(GFM_ERR_OVERFLOW, "%llu hits exceed the capacity of %lld rows", cnt, (long long)capacity);
        goto done;
    }
    if (cnt) {
        rows.resize(cnt);
        SCAN_TRY(hipMemcpyAsync(rows.data(), d_rows, sizeof(long long) * cnt, hipMemcpyDeviceToHost, st));
        if (need_hist) {
            q.resize(m->L);
            SCAN_TRY(hipMemcpyAsync(q.data(), d_q, sizeof(double) * (size_t)m->L, hipMemcpyDeviceToHost, st));
        }
        SCAN_TRY(hipStreamSynchronize(st));
        std::sort(rows.begin(), rows.end());  // packed (row << 20 | score): ascending by row
        for (size_t i = 0; i < cnt; ++i) {
            const int s = (int)(rows[i] & ((1ll << GFM_HIT_SCORE_BITS) - 1));
            if (h_rows) h_rows[i] = rows[i] >> GFM_HIT_SCORE_BITS;
            if (h_scores) h_scores[i] = s;
            if (h_logodds) h_logodds[i] = ((double)s / (double)m->scale) + ((double)m->W * m->offset);
            if (h_pvalue) h_pvalue[i] = m->h_ptable[s];
            if (h_qvalue && need_hist) h_qvalue[i] = q[s];
        }
    }
done:
#undef SCAN_TRY
#undef SCAN_RC
    if (d_kmers) (void)hipFree(d_kmers);
    if (d_scores) (void)hipFree(d_scores);
    if (d_rows) (void)hipFree(d_rows);
    if (d_count) (void)hipFree(d_count);
    if (d_cutoff) (void)hipFree(d_cutoff);
    if (d_hist) (void)hipFree(d_hist);
    if (d_q) (void)hipFree(d_q);
    if (st) (void)hipStreamDestroy(st);
    return rc;
}
