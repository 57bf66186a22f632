// gfm_score_quad.hpp -- score_quad_kernel<W, MM>: the score kernel (the dominant, HBM-bound one)
// Part of libgrafimo_hip.so (instantiated in score_quad_tu.hip).
// Reference lines cited as file:line are relative to /root/reference/src/grafimo/.
#pragma once

#include "gfm_score_kernels.hpp"
#include "gfm_quad_launch.hpp"

namespace {

// ---------------------------------------------------------------------------------------
// score_quad_kernel<W, MM>: compute_score_seq (score_sequences.py:331-396) for a dense batch of k-mers against
// MM motifs of width W (MM = 1: the single-motif launch; 2, 3: same-width motifs sharing ONE read of the k-mers,
// BASELINE config 5 -- per (k-mer, motif) pair the algorithmic bytes drop from W + 4 to W/MM + 4).
//
// Data path: coalesced 16 B/lane non-temporal loads into registers one chunk ahead, a wave-private LDS
// strip (no workgroup barrier in the loop), table lookups of TWO bases at a time (bits 1..3 of two ASCII codes
// -> 6-bit index into that pair's 64-entry table; the 16 hot entries sit in 8 distinct banks: conflict free;
// a code 4..7 -- N, any byte that is not A,C,G,T -- hits a poison entry: the k-mer scores min_val,
// score_sequences.py:376-378; positions >= W contribute 0 whatever their byte), an LDS histogram window per motif
// flushed once per workgroup as a plain-store slab (rows outside a partial window go to a global spill
// array), hits compacted with ballots into per-wave LDS queues (hitq_push).  MM == 1: uint16 table entries;
// MM > 1: ONE table of 64-bit entries packing the motifs' partial scores in 19-bit fields (a k-mer's score is
// <= 64000 < 2^19, so fields never carry) plus, in bits 57.., a count of invalid codes: one ds_read_b64 + one
// 64-bit add per base pair serves all motifs of the launch.
// Work split: a wave takes 256 k-mers per step and every lane scores FOUR consecutive
// rows.  Four rows are 4*W bytes = W dwords, so lane r's rows start dword aligned at r*W dwords:
//   * the row-wise re-read of the strip is W aligned dwords per lane, 4.75 per k-mer at W = 19
//     instead of 6, and free of bank conflicts: the lanes of a read are W dwords apart, W odd ->
//     32 distinct banks (W = 2 mod 4: ds_read_b64 on 64 banks; W = 4 mod 8: ds_read_b128; W = 0
//     mod 8: ds_read_b128 with 16 bytes of padding per lane).  The row-per-lane form at a 19-byte
//     pitch costs 8.1 LDS cycles per ds_read2_b32 against 4.4 here (scripts/micro/lds_cost.hip);
//   * the shifts that re-align rows 1..3 are compile-time constants (W is a template parameter),
//     row 0 needs none;
//   * the four scores of a lane leave as ONE 16-byte store: a wave writes 1 KiB contiguous.
// LDS of a workgroup: pair tables | per wave: strip of 64 x (4W + pad) bytes, MM hit queues | MM histogram
// windows.  8 or 16 waves per workgroup (blockDim.x), chosen by the host so that the windows fit.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x2_t lds_cu32x2;
typedef __attribute__((address_space(3))) const u32x4_t lds_cu32x4;
typedef int i32x4 __attribute__((ext_vector_type(4)));

// STORE = false: the instantiation for callers that pass d_scores == NULL (the product's scans: histogram and hits only).  A
// per-launch branch instead cost the storing launch 1.5 % (84.6 against 83.3 us for 2e7 rows, same box, scripts/lab_bench.sh).
template <int W, int MM, bool STORE = true>
__global__ void __launch_bounds__(kThreads)
score_quad_kernel(const uint8_t *__restrict__ kmers, long long n, long long row_base, const ScoreArgs<MM> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NDW = (W + 3) / 4;
    constexpr int kTabBytes = quad_tab_bytes(W, MM);
    constexpr int CB = kQuadRows * W;                 // bytes of a chunk (a multiple of 256)
    constexpr int kLoads = (CB + 1023) / 1024;        // 16 B loads per lane per chunk (== NDW)
    constexpr int GP = quad_pitch(W);
    constexpr int PAD = quad_pad(W);
    constexpr int SSTRIDE = quad_strip_stride(W, MM);

    if (lds_offset(smem) != 0u) __builtin_trap();     // the lookups below use absolute LDS offsets
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_waves = (int)(blockDim.x >> 6);
    const int wg_threads = (int)blockDim.x;
    unsigned char *tab = smem;
    unsigned char *stage_base = smem + kTabBytes;
    unsigned *hist[MM];
    {
        unsigned *h = reinterpret_cast<unsigned *>(stage_base + n_waves * SSTRIDE);
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            hist[m] = h;
            if (a.m[m].use_hist) h += a.m[m].nb + 1;
        }
    }

    unsigned char *stage = stage_base + wave * SSTRIDE;
    const unsigned stage_off = (unsigned)(kTabBytes + wave * SSTRIDE);   // its absolute LDS offset
    // The loop below takes whole chunks only -- every byte in range, every row live: no bounds logic in
    // the hot code (the ragged-end handling of an earlier version, unrolled per load, was most of the loop's
    // instruction bytes).  The < 256 rows behind the last whole chunk are scored row by row behind it.
    const long long nfull = n / kQuadRows;
    const long long cstride = (long long)gridDim.x * n_waves;
    bool vec_store = true;                            // uniform: every score buffer 16-byte aligned
    const bool through = a.store_through == 1;        // uniform: cache policy of the score stores
    constexpr bool no_store = !STORE;                 // d_scores == NULL: the caller asked for no scores at all
#pragma unroll
    for (int m = 0; m < MM; ++m) vec_store = vec_store && (reinterpret_cast<uintptr_t>(a.m[m].scores) & 15u) == 0;

    uint4 pre[kQuadDepth][kLoads];
    auto fetch = [&](uint4 (&dst)[kLoads], long long c) {
        const uint8_t *src = kmers + c * (long long)CB + lane * 16;
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
            if (i * 1024 + 1024 <= CB || i * 1024 + lane * 16 < CB) {   // the last piece may cover fewer lanes
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                // once-read stream: non-temporal policy (plain loads: 5.1 TB/s, nt: 6.2 TB/s on the same
                // byte mix, scripts/micro/stream_bw_nt.hip)
                const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + i * 1024));
                dst[i] = make_uint4(t.x, t.y, t.z, t.w);
            }
        }
    };

    long long *hitq[MM];
    int qn[MM];   // wave-uniform
    bool select[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        hitq[m] = reinterpret_cast<long long *>(stage + quad_stage_bytes(W)) + m * kHitQueue;
        qn[m] = 0;
        select[m] = a.m[m].cutoff != GFM_NO_SELECT;
        if (select[m] && blockIdx.x == 0 && tid == 0)
            a.m[m].ctl->snap[a.m[m].slot] = a.m[m].hit_count ? *a.m[m].hit_count : 0ull;
    }

    // Histogram and hit selection of a scored chunk ("booking").  It is done one step late, at the top of the
    // next step and in front of the wait for that step's k-mers: the LDS atomics and the queue code then run
    // while the loads are still in flight instead of lengthening the part of the step that follows them.
    int p_score[MM][4];
    long long p_crow = -1;      // chunk whose scores are waiting to be booked (-1: none)
    auto book = [&]() {
        const int k0 = 4 * lane;
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            const MotifArgs &ma = a.m[m];
            if (ma.use_hist) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned off = (unsigned)(p_score[m][j] - ma.lo);
                    if (off < (unsigned)ma.nb)
                        atomicAdd(&hist[m][off], 1u);
                    else if (p_score[m][j] == ma.min_val)   // a row holding N scores min_val, below every reachable
                        atomicAdd(&hist[m][ma.nb], 1u);     // sum unless the window holds min_val itself
                    else
                        atomicAdd(&ma.spill[(size_t)(blockIdx.x & (kSpillCopies - 1)) * (size_t)ma.spill_n + (size_t)(p_score[m][j] - ma.spill_lo)], 1u);   // outside the window: rare
                }
            }
            if (select[m]) {
                const int best = max(max(p_score[m][0], p_score[m][1]), max(p_score[m][2], p_score[m][3]));
                if (__builtin_amdgcn_ballot_w64(best >= ma.cutoff)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        hitq_push(hitq[m], qn[m], p_score[m][j] >= ma.cutoff,
                                  ((row_base + p_crow + k0 + j) << GFM_HIT_SCORE_BITS) | (long long)p_score[m][j], lane,
                                  ma.hit_count, &ma.ctl->mid[ma.slot], ma.hit_rows, ma.hit_cap);
                }
            }
        }
    };

    long long c = (long long)blockIdx.x * n_waves + wave;
#pragma unroll
    for (int d = 0; d < kQuadDepth; ++d)
        if (c + d * cstride < nfull) fetch(pre[d], c + d * cstride);
    // tables and the zeroed windows AFTER the first loads went out: their latency hides behind this
    if constexpr (MM == 1) {
        for (int i = tid; i < kTabBytes / 2; i += wg_threads) reinterpret_cast<uint16_t *>(tab)[i] = a.m[0].tab[i];
    } else {
        for (int i = tid; i < kTabBytes / 8; i += wg_threads) {
            unsigned long long v = 0;
            if (a.m[0].tab[i] == kPoison) {
                v = 1ull << 57;                       // the same codes are invalid for every motif
            } else {
#pragma unroll
                for (int m = 0; m < MM; ++m) v |= (unsigned long long)a.m[m].tab[i] << (19 * m);
            }
            reinterpret_cast<unsigned long long *>(tab)[i] = v;
        }
    }
#pragma unroll
    for (int m = 0; m < MM; ++m)
        if (a.m[m].use_hist)
            for (int i = tid; i <= a.m[m].nb; i += wg_threads) hist[m][i] = 0u;
    __syncthreads();

    while (c < nfull) {
#pragma unroll
        for (int d = 0; d < kQuadDepth; ++d) {
            if (c >= nfull) break;
            const long long crow = c * kQuadRows;   // first row of the chunk
            if (p_crow >= 0) book();
#pragma unroll
            for (int i = 0; i < kLoads; ++i) {
                const int off = i * 1024 + lane * 16;
                if (i * 1024 + 1024 <= CB || off < CB) {
                    // W % 8 == 0: 16 bytes of padding after every lane's 4*W bytes (a piece never
                    // straddles two lanes' data then: 4*W is a multiple of 16)
                    const int dst = PAD ? off + PAD * (off / (4 * W)) : off;
                    *reinterpret_cast<uint4 *>(stage + dst) = pre[d][i];
                }
            }
            // the next chunk's loads go out before this chunk is scored.  ONE chunk ahead: with two or
            // three ahead the kernel took 7 us longer (96 vs 89 us at 2e7 rows) -- a wave that finds the
            // memory pipeline backed up stalls at the load ISSUE, in front of its own compute
            if (c + kQuadDepth * cstride < nfull) fetch(pre[d], c + kQuadDepth * cstride);
            // LDS ops of one wave execute in program order; the fence only pins the compiler.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();

            // the lane's four rows: W dwords (+ a zero behind them for the funnel shift of the last dword)
            unsigned w[W + 1];
            {
                const unsigned base = stage_off + (unsigned)(lane * GP);
                if constexpr (W % 2 == 1) {
#pragma unroll
                    for (int t = 0; t < W; ++t) w[t] = ((lds_cu32 *)(uintptr_t)base)[t];
                } else if constexpr (W % 4 == 2) {
#pragma unroll
                    for (int t = 0; t < W / 2; ++t) {
                        const u32x2_t v = ((lds_cu32x2 *)(uintptr_t)base)[t];
                        w[2 * t] = v.x;
                        w[2 * t + 1] = v.y;
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < W / 4; ++t) {
                        const u32x4_t v = ((lds_cu32x4 *)(uintptr_t)base)[t];
                        w[4 * t] = v.x;
                        w[4 * t + 1] = v.y;
                        w[4 * t + 2] = v.z;
                        w[4 * t + 3] = v.w;
                    }
                }
                w[W] = 0u;
            }
            int score[MM][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = (j * W) & 3;     // byte phase of row j inside its first dword: a constant
                bool any_n;
                int acc[MM];
                if constexpr (MM == 1) {
                    int s1 = 0;
#pragma unroll
                    for (int t = 0; t < NDW; ++t) {
                        const int q = (j * W + 4 * t) >> 2;
                        const unsigned x = r ? __builtin_amdgcn_alignbit(w[q + 1], w[q], (unsigned)(8 * r)) : w[q];
                        // bits 1..3 of each 16-bit half from x (first base of a pair), the other bits from
                        // x >> 5 (bits 4..6: second base); then one mask per table offset
                        unsigned y;
                        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(y) : "s"(0x000E000Eu), "v"(x), "v"(x >> 5));
                        const unsigned e0 = y & 0x7Eu;
                        const unsigned e1 = (y >> 16) & 0x7Eu;
                        s1 += *(lds_cu16 *)(uintptr_t)(e0 + (unsigned)((2 * t) * 128));
                        s1 += *(lds_cu16 *)(uintptr_t)(e1 + (unsigned)((2 * t + 1) * 128));
                    }
                    any_n = (unsigned)s1 >= kPoison;   // a base that is not A,C,G,T: min_val (:376-378)
                    acc[0] = s1;
                } else {
                    unsigned long long s64 = 0;
#pragma unroll
                    for (int t = 0; t < NDW; ++t) {
                        const int q = (j * W + 4 * t) >> 2;
                        const unsigned x = r ? __builtin_amdgcn_alignbit(w[q + 1], w[q], (unsigned)(8 * r)) : w[q];
                        unsigned y;
                        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(y) : "s"(0x000E000Eu), "v"(x), "v"(x >> 5));
                        const unsigned e0 = (y & 0x7Eu) << 2;            // 8-byte entries
                        const unsigned e1 = ((y >> 16) & 0x7Eu) << 2;
                        s64 += *(lds_cu64 *)(uintptr_t)(e0 + (unsigned)((2 * t) * 512));
                        s64 += *(lds_cu64 *)(uintptr_t)(e1 + (unsigned)((2 * t + 1) * 512));
                    }
                    any_n = (s64 >> 57) != 0;
#pragma unroll
                    for (int m = 0; m < MM; ++m) acc[m] = (int)((s64 >> (19 * m)) & 0x7FFFFull);
                }
#pragma unroll
                for (int m = 0; m < MM; ++m) score[m][j] = any_n ? a.m[m].min_val : acc[m];
            }
            const int k0 = 4 * lane;
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                // Scores leave through buffer stores, one descriptor per chunk and motif (base = the chunk's 1 KiB
                // of scores: wave-uniform, SGPRs).  Two cache policies, chosen per launch by the host
                // (score_store_through() in grafimo_hip.hip, where the measurements are): write-through while the
                // launch's scores fit the Infinity Cache, streaming (nt) beyond.
                const __amdgpu_buffer_rsrc_t rs =
                    __builtin_amdgcn_make_buffer_rsrc(a.m[m].scores + crow, 0, kQuadRows * 4, 0x00020000);
                if (no_store) {
                    // d_scores == NULL: what the product's scans ask for -- hits and histogram come out of this kernel, and
                    // the int32 [N] array (80 MB of the 460 MB a 2e7-row launch moves) would never be read
                } else if (vec_store) {   // one 16-byte store per lane and motif: the wave writes 1 KiB contiguous
                    const u32x4_t out = {(unsigned)score[m][0], (unsigned)score[m][1], (unsigned)score[m][2],
                                         (unsigned)score[m][3]};
                    if (through)
                        __builtin_amdgcn_raw_buffer_store_b128(out, rs, k0 * 4, 0, kStoreThrough);
                    else
                        __builtin_amdgcn_raw_buffer_store_b128(out, rs, k0 * 4, 0, kStoreStream);
                } else {           // a score buffer that is only 4-byte aligned
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (through)
                            __builtin_amdgcn_raw_buffer_store_b32((unsigned)score[m][j], rs, (k0 + j) * 4, 0, kStoreThrough);
                        else
                            __builtin_amdgcn_raw_buffer_store_b32((unsigned)score[m][j], rs, (k0 + j) * 4, 0, kStoreStream);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) p_score[m][j] = score[m][j];
            }
            p_crow = crow;
            // the strip is rewritten next iteration: keep this iteration's reads ahead of it
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            c += cstride;
        }
    }
    if (p_crow >= 0) book();
    // The rows the chunked loop leaves over (fewer than kQuadRows, at the end of the batch): one row per
    // lane, bytes straight from global memory.  Cold and small (rolled loops): once per launch, one wave.
    if (blockIdx.x == gridDim.x - 1 && wave == n_waves - 1 && nfull * kQuadRows < n) {
        const long long first = nfull * kQuadRows;
        const int count = (int)(n - first);
#pragma unroll 1
        for (int r0 = 0; r0 < count; r0 += kWave) {
            const int k = r0 + lane;
            const bool live = k < count;
            const uint8_t *row = kmers + (first + (live ? k : 0)) * (long long)W;
            unsigned long long s64 = 0;
            bool bad = false;
#pragma unroll 1
            for (int p = 0; p < 2 * NDW; ++p) {
                const unsigned b0 = 2 * p < W ? row[2 * p] : (unsigned)'A';        // positions >= W: any valid code
                const unsigned b1 = 2 * p + 1 < W ? row[2 * p + 1] : (unsigned)'A';
                const unsigned e = p * 64 + ((b0 >> 1) & 7u) + 8u * ((b1 >> 1) & 7u);
                if constexpr (MM == 1) {
                    const unsigned v = reinterpret_cast<const uint16_t *>(tab)[e];
                    bad = bad || v == kPoison;
                    s64 += v;
                } else {
                    s64 += reinterpret_cast<const unsigned long long *>(tab)[e];
                }
            }
            if constexpr (MM > 1) bad = (s64 >> 57) != 0;
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                const MotifArgs &ma = a.m[m];
                const int sc = bad ? ma.min_val : (int)((MM == 1 ? s64 : (s64 >> (19 * m))) & 0x7FFFFull);
                if (live) {
                    if (!no_store) ma.scores[first + k] = sc;
                    if (ma.use_hist) {
                        const unsigned off = (unsigned)(sc - ma.lo);
                        if (off < (unsigned)ma.nb)
                            atomicAdd(&hist[m][off], 1u);
                        else if (sc == ma.min_val)
                            atomicAdd(&hist[m][ma.nb], 1u);
                        else
                            atomicAdd(&ma.spill[(size_t)(blockIdx.x & (kSpillCopies - 1)) * (size_t)ma.spill_n + (size_t)(sc - ma.spill_lo)], 1u);
                    }
                }
                if (select[m])
                    hitq_push(hitq[m], qn[m], live && sc >= ma.cutoff,
                              ((row_base + first + k) << GFM_HIT_SCORE_BITS) | (long long)sc, lane, ma.hit_count,
                              &ma.ctl->mid[ma.slot], ma.hit_rows, ma.hit_cap);
            }
        }
    }

    // the lookup tables are dead once every wave has left the loop: their LDS holds the per-wave
    // queue lengths (no static LDS)
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        const MotifArgs &ma = a.m[m];
        if (select[m]) {
            int *wq_n = reinterpret_cast<int *>(tab) + m * kWavesPerWG;
            if (lane == 0) wq_n[wave] = qn[m];
            __syncthreads();
            int base = 0, tot = 0;
            for (int w2 = 0; w2 < n_waves; ++w2) {
                const int v = wq_n[w2];
                if (w2 < wave) base += v;
                tot += v;
            }
            long long *slab = ma.resid + (size_t)blockIdx.x * kResidPerWG;
            for (int i = lane; i < qn[m]; i += kWave) slab[base + i] = hitq[m][i];
            if (tid == 0) ma.resid_n[blockIdx.x] = tot;
        }
        if (ma.use_hist) {
            unsigned *slab = ma.partials + (size_t)blockIdx.x * (size_t)(ma.nb + 1);
            for (int i = tid; i <= ma.nb; i += wg_threads) slab[i] = hist[m][i];
        }
    }
}

}  // namespace
