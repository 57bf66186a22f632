// gfm_stats_kernels.hpp -- p-value DP, tail table and BH q-value kernels
// Part of libgrafimo_hip.so (one translation unit: included by grafimo_hip.hip only).
// Reference lines cited as file:line are relative to /root/reference/src/grafimo/.
#pragma once

#include "gfm_common.hpp"

namespace {

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
constexpr int kQStreams = 8;     // streams per motif handle that may run q-table kernels side by side (one scratch set each)
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

__device__ inline void
q_count_body(const unsigned long long *__restrict__ hist, int L, int lo, int hi, int min_val,
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

__device__ inline void
q_raw_body(const unsigned long long *__restrict__ hist, const double *__restrict__ ptable, int lo,
           int hi, QWork *__restrict__ ws, double *__restrict__ raw_out, int nblk)
{
    __shared__ unsigned long long sh[kQThreads / kWave];
    __shared__ double shm[kQThreads / kWave];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = blockIdx.x;
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

__device__ inline void
q_final_body(const unsigned long long *hist, const double *__restrict__ ptable, int L, int lo,
             int hi, int min_val, double threshold, int on_qvalue, const QWork *__restrict__ ws,
             double *qtable, int *__restrict__ cutoff_out, unsigned long long *__restrict__ nrows_out,
             unsigned long long *__restrict__ clear, int nblk)
{
    __shared__ unsigned long long sh[kQThreads / kWave];
    __shared__ double shm[kQThreads / kWave];
    __shared__ int first_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = blockIdx.x;
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

// The three passes over up to kQJobs motifs in ONE launch each (blockIdx.y = motif): the q-tables of a motif set
// are 3 launches per kQJobs motifs instead of 3 per motif -- in stream order each of these latency-bound launches
// costs ~5 us whatever it computes (fifty motifs: 0.8 ms of a 10.6 ms step).
constexpr int kQJobs = 8;
struct QJob {
    const unsigned long long *hist;
    const double *ptable;
    QWork *ws;
    double *qtable;               // raw values between q_raw and q_final, then q(s)
    int *cutoff;
    unsigned long long *nrows;
    unsigned long long *clear;    // == hist when the histogram is handed back zeroed, else nullptr
    int L, lo, hi, min_val, nblk;
};
struct QJobs {
    QJob j[kQJobs];
    double threshold;
    int on_qvalue;
};

__global__ void __launch_bounds__(kQThreads) q_count_kernel(const QJobs jobs)
{
    const QJob &q = jobs.j[blockIdx.y];
    if ((int)blockIdx.x >= q.nblk) return;
    q_count_body(q.hist, q.L, q.lo, q.hi, q.min_val, q.ws, q.cutoff);
}

__global__ void __launch_bounds__(kQThreads) q_raw_kernel(const QJobs jobs)
{
    const QJob &q = jobs.j[blockIdx.y];
    if ((int)blockIdx.x >= q.nblk) return;
    q_raw_body(q.hist, q.ptable, q.lo, q.hi, q.ws, q.qtable, q.nblk);
}

__global__ void __launch_bounds__(kQThreads) q_final_kernel(const QJobs jobs)
{
    const QJob &q = jobs.j[blockIdx.y];
    if ((int)blockIdx.x >= q.nblk) return;
    q_final_body(q.hist, q.ptable, q.L, q.lo, q.hi, q.min_val, jobs.threshold, jobs.on_qvalue, q.ws, q.qtable,
                 q.cutoff, q.nrows, q.clear, q.nblk);
}

}  // namespace
