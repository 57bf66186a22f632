// grafimo_hip.hip -- host side and C ABI of the MI355X (gfx950 / CDNA4) k-mer scoring path of GRAFIMO.
// The kernels live in gfm_score_kernels.hpp (scoring, histogram, selection) and
// gfm_stats_kernels.hpp (p-value DP, tail table, BH q-values), included below: one translation unit.
//
// Written for gfx950 only: 64-lane wavefronts, 160 KiB LDS per CU, 256 CUs in 8 XCDs.
// The hot op is an HBM-bound gather (W table lookups + W integer adds per k-mer,
// < 1 op per byte): no MFMA.  See DESIGN.md for the roofline and the data layout.
//
// Reference lines cited as file:line are relative to /root/reference/src/grafimo/.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
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

#include "gfm_hit_sort.hpp"
#include "gfm_common.hpp"
#include "gfm_score_kernels.hpp"
#include "gfm_quad_launch.hpp"
#include "gfm_stats_kernels.hpp"

// =======================================================================================
// host side
struct gfm_motif {
    int W = 0, L = 0, min_val = 0, scale = 1, ndw = 0;
    double offset = 0.0;
    int lo = 0, hi = 0, nb = 0;      // reachable score range [lo, hi], nb bins
    int hlo = 0, hnb = 0;            // LDS histogram window of a single-motif launch (== lo, nb when it fits)
    int q_waves = kWavesPerWG;       // waves per workgroup of the single-motif launch (16, or 8 when the window needs the room)
    int part_nb = 0;                 // bins per histogram slab the workspace was sized for
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
    unsigned char *d_slab = nullptr;   // one allocation behind every device pointer below
    uint16_t *d_tab = nullptr;
    unsigned *d_ftab = nullptr;        // the fused graph kernels' table [W][8]: sm[code][j] | sm[comp(code)][W-1-j] << 16
    double *d_pmf = nullptr;
    double *d_ptable = nullptr;
    // Scoring workspace, a ring of kWorkspaces sets taken in call order, so that the post kernel of call k
    // (on a tail stream) may run while the score kernels of calls k+1.. fill the other sets.
    unsigned *d_partials[kWorkspaces] = {};   // [max_slabs][hnb+1] histogram slabs
    unsigned *d_spill[kWorkspaces] = {};      // [kSpillCopies][nb] rows outside a partial window
    long long *d_resid[kWorkspaces] = {};     // [max_slabs][kResidPerWG] residual hits
    int *d_resid_n[kWorkspaces] = {};         // [max_slabs]
    // q-value kernels' scratch, one set PER STREAM that has called gfm_qvalue_table on this handle (kQStreams of
    // them): two pipelines on one handle -- two scanners, each with its tail stream -- run their q-table kernels side
    // by side, and on ONE shared set of block totals each read the other's (a lost-rows count was how it showed).
    // Calls of one stream are ordered by the stream.
    QWork *d_qwork = nullptr;        // [kQStreams] block totals / minima
    double *d_qscratch = nullptr;    // [kQStreams][L] raw BH values when the caller wants no q-table
    hipStream_t q_stream[kQStreams] = {};
    int q_streams_used = 0;
    unsigned long long q_last_use[kQStreams] = {}, q_calls = 0;
    // the selection workspace (gfm_select_hits*) is one set: users on different streams are ordered by this event
    hipEvent_t ev_selected = nullptr;
    hipStream_t sel_last_stream = nullptr;
    bool sel_valid = false;
    HitCtl *d_ctl = nullptr;
    unsigned call_no = 0;           // score calls: HitCtl slot call_no % kCtlSlots, workspace call_no % kWorkspaces
    // GFM_FLAG_CALLER_ORDERS_REUSE is a promise about the call kWorkspaces back ON THIS HANDLE, which a pipelined caller can
    // only keep for its own calls: the flag is honoured when the kWorkspaces calls before this one came from the
    // same (stream, tail stream) pair, and ignored (the library waits on its own event) when another user of the
    // handle was in between.
    hipStream_t last_st = nullptr, last_tail = nullptr;
    unsigned same_pair_run = 0;     // consecutive calls made with (last_st, last_tail) so far
    hipEvent_t ev_scored[kWorkspaces] = {};   // score kernel of the last call on a workspace done
    hipEvent_t ev_posted[kWorkspaces] = {};   // its post kernel done (workspace free again)
    bool posted_valid[kWorkspaces] = {};
    // separate workspace of gfm_select_hits (runs on the caller's tail stream, next to scoring)
    long long *d_sel_resid = nullptr;
    int *d_sel_resid_n = nullptr;
    HitCtl *d_sel_ctl = nullptr;
    unsigned sel_call_no = 0;
    // measurement aid: ring of event pairs around the score kernel
    std::vector<hipEvent_t> ev0, ev1;
    int ev_next = 0, ev_used = 0, ev_every = 1;
    unsigned ev_calls = 0;
    // ... and around the TAIL of a timed call (post kernel and whatever the caller enqueues behind it: collective,
    // q-table, gather): tev0 is recorded on the tail stream behind its wait for the score kernel, tev1 by
    // gfm_profile_mark_tail on the stream the caller names
    std::vector<hipEvent_t> tev0, tev1;
    int ev_last = -1;               // ring index of the last timed launch (-1: the last launch was not timed)
    int tail_pending = -1;          // ring index whose tev0 is recorded and whose tev1 is still to come
    std::vector<int> tail_order;    // ring indices with both tail events recorded, oldest first
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
// device scratch of one DP run: [4W + W + W ints | pad to 8 | 4 + 2L doubles]
size_t dp_scratch_bytes(int W)
{
    const size_t L = (size_t)kRange * W + 1;
    return ((sizeof(int) * 6 * (size_t)W + 7) & ~(size_t)7) + sizeof(double) * (4 + 2 * L);
}

// runs the DP for one motif on the current device; d_pmf receives L doubles.  `scratch`
// (dp_scratch_bytes, 8-byte aligned) saves the temporaries' allocations; nullptr = allocate here.
int run_dp(const int64_t *sm, int W, const double *bg, double *d_pmf, hipStream_t st, void *scratch = nullptr)
{
    const int L = kRange * W + 1;
    std::vector<int> lo, hi, ints(6 * (size_t)W);
    cumulative_windows(sm, W, lo, hi);
    for (int i = 0; i < 4 * W; ++i) ints[(size_t)i] = (int)sm[i];
    std::copy(lo.begin(), lo.end(), ints.begin() + 4 * W);
    std::copy(hi.begin(), hi.end(), ints.begin() + 5 * W);
    DevBuf<unsigned char> own;
    if (!scratch) {
        HIP_TRY(own.alloc(dp_scratch_bytes(W)));
        scratch = own.p;
    }
    int *d_int = static_cast<int *>(scratch);
    double *d_dbl = reinterpret_cast<double *>(static_cast<unsigned char *>(scratch) +
                                               ((sizeof(int) * 6 * (size_t)W + 7) & ~(size_t)7));
    HIP_TRY(hipMemcpyAsync(d_int, ints.data(), sizeof(int) * ints.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_dbl, bg, sizeof(double) * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(pvalue_dp_kernel, dim3(1), dim3(kDpThreads), 0, st, d_int, d_dbl, W, L, d_int + 4 * W,
                       d_int + 5 * W, d_dbl + 4, d_pmf);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));   // the host vectors and the temporaries go away on return
    return GFM_OK;
}

// score_quad_kernel<W, MM> is instantiated in score_quad_tu.hip: eight translation units of sixteen widths and one
// MM each (compiled side by side: one unit with all of them took four minutes).  `args` = ScoreArgs<mm>.
// `done` (optional, in/out): an event to complete with this kernel (it rides on the dispatch packet).  When the
// launch is also a timed one, the timer's stop event takes that place and is handed back through `done`: whoever
// must wait for the kernel waits on what `done` holds after the call.
int dispatch_quad(int W, int mm, gfm_motif *timer, const uint8_t *d_kmers, long long n, long long row_base,
                  const void *args, size_t lds, int nslabs, int waves, hipStream_t st, bool prepare_only,
                  hipEvent_t *done = nullptr)
{
    if (W < 1 || W > GFM_MAX_WIDTH) return fail(GFM_ERR_INVALID, "unsupported width %d", W);
    if (mm < 1 || mm > 3 || (mm > 1 && W > kQuadMaxBatchWidth))
        return fail(GFM_ERR_INVALID, "no kernel for %d motifs of width %d in one launch", mm, W);
    hipEvent_t e0 = nullptr, e1 = done ? *done : nullptr;
    if (!prepare_only && timer && !timer->ev0.empty() && (timer->ev_calls++ % (unsigned)timer->ev_every) == 0) {
        e0 = timer->ev0[timer->ev_next];
        e1 = timer->ev1[timer->ev_next];
        if (done) *done = e1;
        timer->ev_last = timer->ev_next;
        timer->ev_next = (timer->ev_next + 1) % (int)timer->ev0.size();
        timer->ev_used = std::min(timer->ev_used + 1, (int)timer->ev0.size());
    } else if (timer) {
        timer->ev_last = -1;
    }
    typedef int (*launch_fn)(int, const uint8_t *, long long, long long, const void *, size_t, int, int, void *, int,
                             void *, void *);
    static const launch_fn table[3][4] = {
        {gfm_quad_launch_g0_m1, gfm_quad_launch_g1_m1, gfm_quad_launch_g2_m1, gfm_quad_launch_g3_m1},
        {gfm_quad_launch_g0_m2, gfm_quad_launch_g1_m2, nullptr, nullptr},
        {gfm_quad_launch_g0_m3, gfm_quad_launch_g1_m3, nullptr, nullptr}};
    return table[mm - 1][(W - 1) / 16](W, d_kmers, n, row_base, args, lds, nslabs, waves, st, prepare_only ? 1 : 0, e0, e1);
}

// Cache policy of a launch's score stores.  Measured on MI355X, W=19 (profiles/r02_lab_variants.txt, same box per
// line): 2e7 rows (80 MB of scores) nt 85.7-89.2 us, plain 84.4-86.6, sc1 82.4-83.4, sc0 sc1 82.1-83.1;
// 1.25e8 rows (500 MB) nt 560 us, plain 575-619, sc1 600, sc0 sc1 600.  Write-through stores leave their lines
// in the 256 MB Infinity Cache, which pays while a launch's scores fit it next to the k-mer stream and costs
// once they do not.  (Lab builds: GRAFIMO_STORE_POLICY=through|stream overrides the choice.)
constexpr long long kStoreThroughMaxBytes = 96ll << 20;   // crossover between 76 and 114 MiB (scripts/size_sweep.py)
int score_store_through(long long n, int mm)
{
#ifdef GFM_LAB      // lab builds only (scripts/lab_build.sh -DGFM_LAB): the product reads no timing knob
    if (const char *e = std::getenv("GRAFIMO_STORE_POLICY")) {
        if (!std::strcmp(e, "through")) return 1;
        if (!std::strcmp(e, "stream")) return 0;
    }
#endif
    return n * 4ll * mm <= kStoreThroughMaxBytes ? 1 : 0;
}

// LDS bytes of a launch of mm motifs before the histogram windows: pair tables | strips + hit queues
size_t quad_fixed_lds(int W, int waves, int mm = 1)
{
    return (size_t)quad_tab_bytes(W, mm) + (size_t)waves * (size_t)quad_strip_stride(W, mm);
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

// what follows a scoring / selection kernel: histogram slabs -> hist64, hit slabs -> list.  post_job() describes
// one motif's share, launch_posts() runs up to kPostJobs of them in one launch.
PostJob post_job(gfm_motif *m, const unsigned *partials, int hist_slabs, unsigned long long *d_hist,
                 int win_lo, int win_nb, unsigned *spill, const long long *resid, const int *resid_n, int hit_slabs,
                 HitCtl *ctl, int ctl_slot, long long *d_hit_rows, long long cap, unsigned long long *d_hit_count)
{
    PostJob q{};
    q.bin_blocks = (win_nb + 1 + 255) / 256;
    const int groups = (hist_slabs + kSlabsPerGroup - 1) / kSlabsPerGroup;
    q.hist_blocks = d_hist ? q.bin_blocks * groups : 0;
    q.spill_blocks = (d_hist && spill && win_nb < m->nb) ? (m->nb + 255) / 256 : 0;
    q.total = q.hist_blocks + q.spill_blocks + hit_slabs;
    q.partials = partials; q.hist64 = d_hist; q.spill = spill; q.resid = resid; q.resid_n = resid_n; q.ctl = ctl;
    q.hit_rows = d_hit_rows; q.hit_count = d_hit_count; q.hit_cap = cap;
    q.nslabs = hist_slabs; q.nb = win_nb; q.lo = win_lo; q.min_val = m->min_val;
    q.spill_lo = m->lo; q.spill_n = m->nb; q.hit_slabs = hit_slabs; q.par = ctl_slot;
    return q;
}

int launch_posts(const PostJob *jobs, int count, hipStream_t st)
{
    PostJobs all{};
    int live = 0, max_total = 0;
    for (int k = 0; k < count; ++k) {
        const PostJob &q = jobs[k];
        if (q.total == 0) {   // nothing to post, but the rotating hit counter is still handed on zeroed (see post_kernel)
            if (q.ctl)
                HIP_TRY(hipMemsetAsync(&q.ctl->mid[(q.par + kCtlAhead) % kCtlSlots], 0, sizeof(unsigned long long), st));
            continue;
        }
        all.j[live++] = q;
        max_total = std::max(max_total, q.total);
    }
    if (live == 0) return GFM_OK;
    hipLaunchKernelGGL(post_kernel, dim3((unsigned)max_total, (unsigned)live), dim3(256), 0, st, all);
    HIP_TRY(hipGetLastError());
    return GFM_OK;
}

int launch_post(gfm_motif *m, const unsigned *partials, int hist_slabs, unsigned long long *d_hist,
                int win_lo, int win_nb, unsigned *spill, const long long *resid, const int *resid_n, int hit_slabs, HitCtl *ctl, int ctl_slot,
                long long *d_hit_rows, long long cap, unsigned long long *d_hit_count, hipStream_t st)
{
    const PostJob q = post_job(m, partials, hist_slabs, d_hist, win_lo, win_nb, spill, resid, resid_n, hit_slabs, ctl,
                               ctl_slot, d_hit_rows, cap, d_hit_count);
    return launch_posts(&q, 1, st);
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
    a.spill_n = m->nb;
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

// Grouping of a batched launch: the largest group (<= 3 motifs; 1 beyond kQuadMaxBatchWidth) at the head of
// `motifs` whose LDS histogram windows still hold kMinWindowMass of each motif's background score distribution.
// Per group size the shapes are tried in this order: 16 waves per workgroup with every window WHOLE, 8 waves (half the
// strips) with every window whole, then 16 and 8 waves with partial windows.  Windows share what the tables and strips
// leave: a motif whose whole range fits takes it, the others split the rest and spill the rows outside their window
// through global atomics -- which is why whole windows at 8 waves go first: on config 5 (1e8 rows per width) the
// three-motif launches with 8 waves and whole windows run at 0.66-0.75 of the HBM peak, the two widths (12, 14) that
// got 16 waves with 99.9 % windows at 0.51 and 0.62 (profiles/r04_kernel_stats_config5.csv): 1e5 spilled rows per
// motif cost more than half the waves.  with_hist[k]: motif k accumulates a histogram.  -> false: not even one motif fits.
bool plan_group(const gfm_motif_t *motifs, int n_left, const bool *with_hist, int *mm_out, int *waves_out,
                int *win_lo, int *win_nb)
{
    constexpr double kMinWindowMass = 0.999;
    const int W = motifs[0]->W;
    int mm = std::min(W <= kQuadMaxBatchWidth ? 3 : 1, n_left);
    int waves = kWavesPerWG;
    bool found = false;
    for (; mm >= 1 && !found; --mm) {
        // (a lone motif keeps the single-motif launch's order: 16 waves, partial window if need be, before 8 waves)
        for (int pass = mm > 1 ? 0 : 1; pass < 2 && !found; ++pass) {       // 0: whole windows only, 1: partial windows allowed
            for (waves = kWavesPerWG; waves >= kWavesPerWG / 2 && !found; waves /= 2) {
                const long long spare = (long long)kMaxLdsBytes - (long long)quad_fixed_lds(W, waves, mm);   // bytes
                long long room = spare / (long long)sizeof(unsigned);
                bool open[3] = {false, false, false};
                int users = 0;
                for (int k = 0; k < mm; ++k) {
                    win_nb[k] = 0;
                    win_lo[k] = motifs[k]->lo;
                    open[k] = with_hist[k];
                    users += open[k];
                }
                // (tables, strips and queues must fit even when no motif wants a window: wide motifs get 8 waves)
                bool ok = spare >= 0 && (room > 0 || users == 0);
                // water-filling: ranges that fit their equal share are served whole, the rest share again
                for (bool again = true; ok && again && users > 0;) {
                    again = false;
                    const long long share = room / users - 1;
                    for (int k = 0; k < mm; ++k)
                        if (open[k] && motifs[k]->nb <= share) {
                            win_nb[k] = motifs[k]->nb;
                            room -= win_nb[k] + 1;
                            open[k] = false;
                            --users;
                            again = true;
                        }
                }
                if (ok && users > 0) {
                    const long long share = room / users - 1;
                    if (pass == 0 || share < 256) ok = false;
                    for (int k = 0; ok && k < mm; ++k)
                        if (open[k]) {
                            const gfm_motif::Window w = best_window(motifs[k], (int)share);
                            win_nb[k] = w.bins;
                            win_lo[k] = w.lo;
                            // a partial window is acceptable for a lone motif at 8 waves (nothing smaller exists)
                            if ((mm > 1 || waves > kWavesPerWG / 2) && w.mass < kMinWindowMass) ok = false;
                        }
                }
                found = ok;
                if (found) break;
            }
            if (found) break;
        }
        if (found) break;
    }
    *mm_out = mm;
    *waves_out = waves;
    return found;
}

}  // namespace

// --------------------------------------------------------------------------------------- API
extern "C" __attribute__((visibility("hidden"))) void gfm_set_error_(const char *msg) { g_err = msg ? msg : ""; }

// What the fused extraction -> scoring kernels of graph_extract.hip need of a motif (inside the library only): the score
// matrix as the caller gave it (rows A, C, G, T), the score a k-mer holding N gets, the reachable range, and the
// `max_bins` consecutive scores that hold the most background probability (where an LDS histogram window goes).
extern "C" __attribute__((visibility("hidden"))) int gfm_motif_view_(gfm_motif_t m, int max_bins, int small_bins, const int64_t **sm,
                                                                     int *W, int *min_val, int *L, int *win_lo, int *win_nb,
                                                                     int *device, int *n_cu, const unsigned **d_ftab)
{
    if (!m) return fail(GFM_ERR_INVALID, "motif is NULL");
    // `small_bins` consecutive scores if they hold 90 % of the background mass (a window that leaves room for two workgroups
    // of the fused kernels per CU; the rows outside it are booked by global atomics, spread over thousands of bins: W = 40,
    // 10^7 rows, 0.36 ms with the wide window and one workgroup per CU, 0.25 ms with this one), else up to `max_bins`
    gfm_motif::Window w = best_window(m, std::max(1, std::min(max_bins, m->nb)));
    if (small_bins > 0 && small_bins < w.bins) {
        const gfm_motif::Window ws = best_window(m, small_bins);
#ifndef GFM_GRAPH_SMALL_MASS          // (lab builds vary it: scripts/lab_build.sh <tag> -DGFM_GRAPH_SMALL_MASS=0.99)
#define GFM_GRAPH_SMALL_MASS 0.9
#endif
#ifdef GFM_LAB
        static const double need = [] { const char *e = std::getenv("GRAFIMO_FUSED_SMALL_MASS"); return e ? atof(e) : GFM_GRAPH_SMALL_MASS; }();
#else
        constexpr double need = GFM_GRAPH_SMALL_MASS;
#endif
        if (ws.mass >= need) w = ws;
    }
    *sm = m->sm.data();
    *W = m->W;
    *min_val = m->min_val;
    *L = m->L;
    *win_lo = w.lo;
    *win_nb = w.bins;
    *device = m->device;
    *n_cu = m->n_cu;
    if (d_ftab) *d_ftab = m->d_ftab;
    return GFM_OK;
}

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
    if (m->d_slab) (void)hipFree(m->d_slab);
    for (int i = 0; i < kWorkspaces; ++i) {
        if (m->ev_scored[i]) (void)hipEventDestroy(m->ev_scored[i]);
        if (m->ev_posted[i]) (void)hipEventDestroy(m->ev_posted[i]);
    }
    for (auto e : m->ev0) (void)hipEventDestroy(e);
    for (auto e : m->ev1) (void)hipEventDestroy(e);
    for (auto e : m->tev0) (void)hipEventDestroy(e);
    for (auto e : m->tev1) (void)hipEventDestroy(e);
    if (m->ev_selected) (void)hipEventDestroy(m->ev_selected);
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
    {   // min_val indexes the histogram and the tables on the device: it must be the matrix minimum
        // (Motif.min_val, motif_ops.py:1106), the score of a k-mer holding N (score_sequences.py:376-378)
        int64_t mn = sm[0];
        for (int i = 1; i < 4 * W; ++i) mn = std::min(mn, sm[i]);
        if ((int64_t)min_val != mn)
            return fail(GFM_ERR_INVALID, "min_val %d is not the minimum of the score matrix (%lld)", min_val, (long long)mn);
    }
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
    {   // compute-unit count of the device, queried once per process (hipGetDeviceProperties is slow)
        static std::atomic<int> cu_of[64];
        int cu = cu_of[m->device & 63].load(std::memory_order_acquire);
        if (cu == 0) {
            HIP_TRY_M(hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, m->device));
            if (cu <= 0) cu = 256;
            cu_of[m->device & 63].store(cu, std::memory_order_release);
        }
        m->n_cu = cu;
    }
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
    // score-kernel LDS plan: pair tables | wave strips + hit queues | histogram window (+1 N bin).  With 16 waves
    // the strips leave room16 bins; a reachable range that does not fit gets the whole LDS of an 8-wave
    // workgroup (half the strips) unless a 16-wave window still covers practically all of the background
    // mass (decided below, once the tail table exists).
    const long long room16 = ((long long)kMaxLdsBytes - (long long)quad_fixed_lds(W, kWavesPerWG)) / (long long)sizeof(unsigned) - 1;
    const long long room8 = ((long long)kMaxLdsBytes - (long long)quad_fixed_lds(W, kWavesPerWG / 2)) / (long long)sizeof(unsigned) - 1;
    if (room8 < 256) return bail(fail(GFM_ERR_INVALID, "no LDS left for a histogram window at width %d", W));
    // (batched launches carry bigger tables and more hit queues: their windows are never larger than these)
    m->part_nb = (int)std::min<long long>(m->nb, std::max(room16, room8));
    m->max_slabs = m->n_cu * kWGsPerCU;
    m->sel_slabs = 4 * m->n_cu;

    // ONE device allocation for the motif (a motif set creates hundreds of these; ~20 hipMalloc /
    // hipFree pairs per motif were most of the creation time)
    size_t slab_bytes = 0;
    auto carve = [&](size_t bytes) { const size_t at = slab_bytes; slab_bytes += (bytes + 255) & ~(size_t)255; return at; };
    const size_t o_tab = carve(tab.size() * sizeof(uint16_t));
    // the fused extraction -> scoring kernels' table (gfm_graph_fused.hpp): ONE lookup per base serves both strands -- the
    // reverse complement holds comp(base j) at position W-1-j; code 2 = T (row 3), code 3 = G (row 2); codes 4..7 hold 0
    std::vector<unsigned> ftab((size_t)W * 8, 0u);
    {
        static const int row_of_code[4] = {0, 1, 3, 2};
        for (int j = 0; j < W; ++j)
            for (int c = 0; c < 4; ++c) {
                const unsigned fwd = (unsigned)sm[(size_t)row_of_code[c] * W + j];
                const unsigned rc_ = (unsigned)sm[(size_t)row_of_code[c ^ 2] * W + (W - 1 - j)];   // comp: A <-> T, C <-> G
                ftab[(size_t)j * 8 + c] = fwd | (rc_ << 16);
            }
    }
    const size_t o_ftab = carve(ftab.size() * sizeof(unsigned));
    const size_t o_pmf = carve(sizeof(double) * (size_t)m->L);
    const size_t o_ptable = carve(sizeof(double) * (size_t)m->L);
    const size_t o_dp = carve(dp_scratch_bytes(W));
    const size_t o_qwork = carve(sizeof(QWork) * kQStreams);
    const size_t o_qscratch = carve(sizeof(double) * (size_t)m->L * kQStreams);
    size_t o_partials[kWorkspaces], o_resid[kWorkspaces], o_resid_n[kWorkspaces], o_spill[kWorkspaces];
    for (int i = 0; i < kWorkspaces; ++i) {
        o_partials[i] = carve(sizeof(unsigned) * (size_t)m->max_slabs * (size_t)(m->part_nb + 1));
        o_resid[i] = carve(sizeof(long long) * (size_t)m->max_slabs * kResidPerWG);
        o_resid_n[i] = carve(sizeof(int) * (size_t)m->max_slabs);
    }
    const size_t o_sel_resid = carve(sizeof(long long) * (size_t)m->sel_slabs * kResidPerWG);
    const size_t o_sel_resid_n = carve(sizeof(int) * (size_t)m->sel_slabs);
    const size_t o_zero = slab_bytes;                 // what follows starts out zeroed
    for (int i = 0; i < kWorkspaces; ++i) o_spill[i] = carve(sizeof(unsigned) * (size_t)m->nb * (size_t)kSpillCopies);
    const size_t o_ctl = carve(sizeof(HitCtl));
    const size_t o_sel_ctl = carve(sizeof(HitCtl));
    HIP_TRY_M(hipMalloc(&m->d_slab, slab_bytes));
    HIP_TRY_M(hipMemsetAsync(m->d_slab + o_zero, 0, slab_bytes - o_zero, nullptr));
    m->d_tab = reinterpret_cast<uint16_t *>(m->d_slab + o_tab);
    m->d_ftab = reinterpret_cast<unsigned *>(m->d_slab + o_ftab);
    m->d_pmf = reinterpret_cast<double *>(m->d_slab + o_pmf);
    m->d_ptable = reinterpret_cast<double *>(m->d_slab + o_ptable);
    m->d_qwork = reinterpret_cast<QWork *>(m->d_slab + o_qwork);
    m->d_qscratch = reinterpret_cast<double *>(m->d_slab + o_qscratch);
    for (int i = 0; i < kWorkspaces; ++i) {
        m->d_partials[i] = reinterpret_cast<unsigned *>(m->d_slab + o_partials[i]);
        m->d_resid[i] = reinterpret_cast<long long *>(m->d_slab + o_resid[i]);
        m->d_resid_n[i] = reinterpret_cast<int *>(m->d_slab + o_resid_n[i]);
        m->d_spill[i] = reinterpret_cast<unsigned *>(m->d_slab + o_spill[i]);
        HIP_TRY_M(hipEventCreateWithFlags(&m->ev_scored[i], hipEventDisableTiming | hipEventReleaseToDevice));
        HIP_TRY_M(hipEventCreateWithFlags(&m->ev_posted[i], hipEventDisableTiming | hipEventReleaseToDevice));
    }
    HIP_TRY_M(hipEventCreateWithFlags(&m->ev_selected, hipEventDisableTiming));
    m->d_sel_resid = reinterpret_cast<long long *>(m->d_slab + o_sel_resid);
    m->d_sel_resid_n = reinterpret_cast<int *>(m->d_slab + o_sel_resid_n);
    m->d_ctl = reinterpret_cast<HitCtl *>(m->d_slab + o_ctl);
    m->d_sel_ctl = reinterpret_cast<HitCtl *>(m->d_slab + o_sel_ctl);

    HIP_TRY_M(hipMemcpyAsync(m->d_tab, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice, nullptr));
    HIP_TRY_M(hipMemcpyAsync(m->d_ftab, ftab.data(), ftab.size() * sizeof(unsigned), hipMemcpyHostToDevice, nullptr));
    if (h_pmf) {
        HIP_TRY_M(hipMemcpyAsync(m->d_pmf, h_pmf, sizeof(double) * (size_t)m->L, hipMemcpyHostToDevice, nullptr));
    } else {
        rc = run_dp(sm, W, bg, m->d_pmf, nullptr, m->d_slab + o_dp);
        if (rc) return bail(rc);
    }
    hipLaunchKernelGGL(ptable_kernel, dim3(1), dim3(kScanThreads), 0, nullptr, m->d_pmf, m->L,
                       m->lo, m->hi, m->d_ptable);
    HIP_TRY_M(hipGetLastError());
    m->h_ptable.resize(m->L);
    HIP_TRY_M(hipMemcpy(m->h_ptable.data(), m->d_ptable, sizeof(double) * (size_t)m->L,
                        hipMemcpyDeviceToHost));       // also completes the async copies of `tab` / h_pmf
    {   // waves per workgroup and histogram window of the single-motif launch
        constexpr double kWindowMass16 = 0.9999;   // a partial window beside 16 waves must hold this much
        m->q_waves = kWavesPerWG;
        if (m->nb <= room16) {
            m->hnb = m->nb;
        } else if (room16 >= 256 && best_window(m, (int)room16).mass >= kWindowMass16) {
            m->hnb = (int)room16;
        } else {
            m->q_waves = kWavesPerWG / 2;
            m->hnb = (int)std::min<long long>(m->nb, room8);
        }
#ifdef GFM_LAB
        if (const char *e = std::getenv("GRAFIMO_SCORE_WAVES")) {   // force 8 or 16 waves
            const int wv = atoi(e);
            if (wv == 8 || (wv == 16 && room16 >= 256)) {
                m->q_waves = wv;
                m->hnb = (int)std::min<long long>(m->nb, wv == 8 ? room8 : room16);
            }
        }
#endif
        m->hlo = best_window(m, m->hnb).lo;   // partial when the range does not fit: the rest spills
        m->lds_bytes = quad_fixed_lds(W, m->q_waves) + sizeof(unsigned) * (size_t)(m->hnb + 1);
    }
#undef HIP_TRY_M
    {   // allow up to the whole LDS for every instantiation this width can use
        ScoreArgs<3> none{};
        rc = GFM_OK;
        for (int mm = 1; mm <= (W <= kQuadMaxBatchWidth ? 3 : 1) && !rc; ++mm)
            rc = dispatch_quad(W, mm, nullptr, nullptr, 0, 0, &none, 0, 1, m->q_waves, nullptr, true);
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
    if (!d_kmers) return fail(GFM_ERR_INVALID, "NULL device buffer");
    {
        int dev = -1;
        HIP_TRY(hipGetDevice(&dev));
        if (dev != m->device)
            return fail(GFM_ERR_INVALID, "the motif lives on device %d, the current device is %d", m->device, dev);
    }
    if ((reinterpret_cast<uintptr_t>(d_kmers) & 15u) != 0)
        return fail(GFM_ERR_INVALID, "d_kmers must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(d_scores) & 3u) != 0)
        return fail(GFM_ERR_INVALID, "d_scores must be 4-byte aligned");
    if (!d_scores && !d_hist && select_cutoff == GFM_NO_SELECT)
        return fail(GFM_ERR_INVALID, "nothing to do: no scores, no histogram, no selection");
    if (n > (int64_t)kQuadRows * 0x7fffff00ll)
        return fail(GFM_ERR_INVALID, "too many rows for one launch (split the batch)");
    const bool select = select_cutoff != GFM_NO_SELECT;
    if (select && (!d_hit_rows || !d_hit_count))
        return fail(GFM_ERR_INVALID, "selection requested without hit buffers");
    const int use_hist = d_hist ? 1 : 0;
    const long long nchunks = (n + kQuadRows - 1) / kQuadRows;
    const long long want = (nchunks + m->q_waves - 1) / m->q_waves;
    // with a tail stream a few CUs are left free so that its kernels (post, q-table, RCCL) find
    // room without evicting a persistent score workgroup (which would delay the whole grid)
    const int avail = split ? std::max(1, m->max_slabs - m->reserve_cus * (m->max_slabs / m->n_cu)) : m->max_slabs;
    int nslabs = (int)std::min<long long>(want, avail);
    if (want > avail) {
        // Persistent grid, chunks dealt round-robin: every wave takes ceil(chunks / waves) turns, and the last turn
        // is mostly idle unless the counts divide well (2e7 rows: 78 125 chunks over 4 096 waves = 19.07 turns ->
        // 20 turns, 4.6 % of the slots empty; over 245 workgroups = 19.93 turns -> 0.4 %).  The kernel is HBM-bound
        // and a few workgroups fewer move the same bytes, so take the grid in [7/8 avail, avail] that wastes least.
        long long best_cap = -1;
        for (int g = avail; g >= std::max(1, avail - avail / 8); --g) {
            const long long per_turn = (long long)g * m->q_waves;
            const long long cap = (nchunks + per_turn - 1) / per_turn * per_turn;
            if (best_cap < 0 || cap < best_cap) { best_cap = cap; nslabs = g; }
        }
#ifdef GFM_LAB
        if (const char *e = std::getenv("GRAFIMO_SCORE_GRID")) {   // force the grid size
            const int g = atoi(e);
            if (g >= 1 && g <= m->max_slabs) nslabs = g;
        }
#endif
    }

    const unsigned k = m->call_no++;
    const int ws = (int)(k % (unsigned)kWorkspaces), slot = (int)(k % (unsigned)kCtlSlots);
    const int ws_prev = (int)((k + kWorkspaces - 1) % (unsigned)kWorkspaces);
    const bool reset = (flags & GFM_FLAG_RESET_HITS) != 0;
    // workspace `ws` was last used by call k - kWorkspaces: its post kernel must be done.  Appending to a
    // hit list additionally needs the count published by call k-1.
    const bool same_owner = st == m->last_st && tail == m->last_tail;
    const bool caller_orders = (flags & GFM_FLAG_CALLER_ORDERS_REUSE) && same_owner && m->same_pair_run >= (unsigned)kWorkspaces;
    m->same_pair_run = same_owner ? m->same_pair_run + 1 : 1;
    m->last_st = st;
    m->last_tail = tail;
    if (m->posted_valid[ws] && !caller_orders)
        HIP_TRY(hipStreamWaitEvent(st, m->ev_posted[ws], 0));
    if (split && select && !reset && m->posted_valid[ws_prev])
        HIP_TRY(hipStreamWaitEvent(st, m->ev_posted[ws_prev], 0));
    ScoreArgs<1> args{};
    args.store_through = d_scores ? score_store_through(n, 1) : 2;      // 2: no score store at all
    fill_motif_args(args.m[0], m, ws, slot, use_hist, m->hlo, m->hnb, select_cutoff, d_scores,
                    reinterpret_cast<long long *>(d_hit_rows), hit_capacity,
                    reset ? nullptr : reinterpret_cast<unsigned long long *>(d_hit_count));
    hipEvent_t scored = split ? m->ev_scored[ws] : nullptr;
    int rc = dispatch_quad(m->W, 1, m, d_kmers, n, row_base, &args, m->lds_bytes, nslabs, m->q_waves, st, false,
                           split ? &scored : nullptr);
    if (rc) return rc;
    if (split) HIP_TRY(hipStreamWaitEvent(tail, scored, 0));
    m->tail_pending = -1;
    if (split && m->ev_last >= 0 && !m->tev0.empty()) {   // a timed call: the tail's clock starts when the score kernel is done
        HIP_TRY(hipEventRecord(m->tev0[m->ev_last], tail));
        m->tail_pending = m->ev_last;
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
    if (!motifs || n_motifs < 1) return fail(GFM_ERR_INVALID, "NULL argument");
    if (n < 0) return fail(GFM_ERR_INVALID, "negative row count");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool reset = (flags & GFM_FLAG_RESET_HITS) != 0;
    // d_scores == NULL (or every entry NULL): no score is stored -- histograms and hit lists only
    bool no_scores = d_scores == nullptr;
    if (d_scores) {
        int n_null = 0;
        for (int i = 0; i < n_motifs; ++i) n_null += d_scores[i] ? 0 : 1;
        if (n_null != 0 && n_null != n_motifs) return fail(GFM_ERR_INVALID, "d_scores: every entry or none (scores are stored for all motifs of a call or for none)");
        no_scores = n_null == n_motifs;
    }
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
    if (n > (int64_t)kQuadRows * 0x7fffff00ll)
        return fail(GFM_ERR_INVALID, "too many rows for one launch (split the batch)");
    for (int i = 0; i < n_motifs && !no_scores; ++i)
        if ((reinterpret_cast<uintptr_t>(d_scores[i]) & 3u) != 0)
            return fail(GFM_ERR_INVALID, "d_scores[%d] must be 4-byte aligned", i);
    const int W = motifs[0]->W;
    const long long nchunks = (n + kQuadRows - 1) / kQuadRows;

    int i = 0;
    while (i < n_motifs) {
        int mm = 0, waves = kWavesPerWG, nb_lds[3] = {0, 0, 0};
        int win_lo[3] = {0, 0, 0}, win_nb[3] = {0, 0, 0};
        bool with_hist[3];
        for (int k = 0; k < 3; ++k) with_hist[k] = i + k < n_motifs && d_hist && d_hist[i + k];
        if (!plan_group(motifs + i, n_motifs - i, with_hist, &mm, &waves, win_lo, win_nb))
            return fail(GFM_ERR_INVALID, "no LDS left for a histogram window at width %d", W);
        for (int k = 0; k < mm; ++k) {
            if (win_nb[k] > motifs[i + k]->part_nb)
                return fail(GFM_ERR_INVALID, "internal error: histogram window larger than the workspace");
            nb_lds[k] = (d_hist && d_hist[i + k]) ? win_nb[k] + 1 : 0;
        }
        size_t lds = quad_fixed_lds(W, waves, mm);
        for (int k = 0; k < mm; ++k) lds += sizeof(unsigned) * (size_t)nb_lds[k];
        const long long want = (nchunks + waves - 1) / waves;
        int nslabs = (int)std::min<long long>(want, motifs[i]->max_slabs);
        for (int k = 1; k < mm; ++k) nslabs = std::min(nslabs, motifs[i + k]->max_slabs);
        int ws[3], slot[3], uh[3];
        ScoreArgs<1> a1{};
        ScoreArgs<2> a2{};
        ScoreArgs<3> a3{};
        a1.store_through = a2.store_through = a3.store_through = no_scores ? 2 : score_store_through(n, mm);
        for (int k = 0; k < mm; ++k) {
            gfm_motif *mo = motifs[i + k];
            const unsigned c = mo->call_no++;
            mo->same_pair_run = 0;             // another user of the handle: see gfm_motif::same_pair_run
            mo->last_st = mo->last_tail = nullptr;
            ws[k] = (int)(c % (unsigned)kWorkspaces);
            slot[k] = (int)(c % (unsigned)kCtlSlots);
            mo->posted_valid[ws[k]] = false;   // single stream: stream order protects the workspace
            uh[k] = (d_hist && d_hist[i + k]) ? 1 : 0;
            const int cut = select_cutoffs ? select_cutoffs[i + k] : GFM_NO_SELECT;
            const bool sel = cut != GFM_NO_SELECT;
            MotifArgs &dst = mm == 1 ? a1.m[k] : (mm == 2 ? a2.m[k] : a3.m[k]);
            fill_motif_args(dst, mo, ws[k], slot[k], uh[k], win_lo[k], win_nb[k], cut, no_scores ? nullptr : d_scores[i + k],
                            sel ? reinterpret_cast<long long *>(d_hit_rows[i + k]) : nullptr,
                            sel ? hit_capacity[i + k] : 0,
                            (sel && !reset) ? reinterpret_cast<const unsigned long long *>(d_hit_count[i + k])
                                            : nullptr);
        }
        const void *args = mm == 1 ? static_cast<const void *>(&a1) : (mm == 2 ? static_cast<const void *>(&a2) : &a3);
        int rc = dispatch_quad(W, mm, motifs[i], d_kmers, n, row_base, args, lds, nslabs, waves, st, false);
        if (rc) return rc;
        PostJob posts[kPostJobs];
        for (int k = 0; k < mm; ++k) {
            gfm_motif *mo = motifs[i + k];
            const int cut = select_cutoffs ? select_cutoffs[i + k] : GFM_NO_SELECT;
            const bool sel = cut != GFM_NO_SELECT;
            posts[k] = post_job(mo, mo->d_partials[ws[k]], nslabs,
                                uh[k] ? reinterpret_cast<unsigned long long *>(d_hist[i + k]) : nullptr,
                                win_lo[k], win_nb[k], mo->d_spill[ws[k]], mo->d_resid[ws[k]], mo->d_resid_n[ws[k]],
                                sel ? nslabs : 0, mo->d_ctl, slot[k],
                                sel ? reinterpret_cast<long long *>(d_hit_rows[i + k]) : nullptr,
                                sel ? hit_capacity[i + k] : 0,
                                sel ? reinterpret_cast<unsigned long long *>(d_hit_count[i + k]) : nullptr);
        }
        rc = launch_posts(posts, mm, st);
        if (rc) return rc;
        i += mm;
    }
    return GFM_OK;
}

GFM_API int gfm_score_kmers_multi_plan(const gfm_motif_t *motifs, int n_motifs, const int32_t *with_hist,
                                       int32_t *group_size_out, int32_t *waves_out)
{
    if (!motifs || n_motifs < 1) return fail(GFM_ERR_INVALID, "NULL argument");
    for (int i = 0; i < n_motifs; ++i) {
        if (!motifs[i]) return fail(GFM_ERR_INVALID, "motif %d is NULL", i);
        if (motifs[i]->W != motifs[0]->W)
            return fail(GFM_ERR_INVALID, "motifs of one batched launch must share their width (%d vs %d)",
                        motifs[i]->W, motifs[0]->W);
    }
    int i = 0;
    while (i < n_motifs) {
        int mm = 0, waves = 0, win_lo[3], win_nb[3];
        bool wh[3];
        for (int k = 0; k < 3; ++k) wh[k] = i + k < n_motifs && (!with_hist || with_hist[i + k]);
        if (!plan_group(motifs + i, n_motifs - i, wh, &mm, &waves, win_lo, win_nb))
            return fail(GFM_ERR_INVALID, "no LDS left for a histogram window at width %d", motifs[0]->W);
        for (int k = 0; k < mm; ++k) {
            if (group_size_out) group_size_out[i + k] = mm;
            if (waves_out) waves_out[i + k] = waves;
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
    for (auto e : m->tev0) (void)hipEventDestroy(e);
    for (auto e : m->tev1) (void)hipEventDestroy(e);
    m->ev0.clear();
    m->ev1.clear();
    m->tev0.clear();
    m->tev1.clear();
    m->tail_order.clear();
    m->ev_next = m->ev_used = 0;
    m->ev_last = m->tail_pending = -1;
    for (int i = 0; i < slots; ++i) {
        hipEvent_t a, b, c, d;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        HIP_TRY(hipEventCreate(&c));
        HIP_TRY(hipEventCreate(&d));
        m->ev0.push_back(a);
        m->ev1.push_back(b);
        m->tev0.push_back(c);
        m->tev1.push_back(d);
    }
    return GFM_OK;
}

GFM_API int gfm_profile_mark_tail(gfm_motif_t m, void *stream)
{
    if (!m) return fail(GFM_ERR_INVALID, "motif is NULL");
    if (m->tail_pending < 0) return GFM_OK;               // the last call was not a timed one
    HIP_TRY(hipEventRecord(m->tev1[m->tail_pending], static_cast<hipStream_t>(stream)));
    if ((int)m->tail_order.size() < (int)m->tev0.size()) m->tail_order.push_back(m->tail_pending);
    m->tail_pending = -1;
    return GFM_OK;
}

GFM_API int gfm_profile_read_tail(gfm_motif_t m, float *h_ms, int capacity, int *n_out)
{
    if (!m || !n_out || (!h_ms && capacity)) return fail(GFM_ERR_INVALID, "NULL argument");
    const int used = std::min((int)m->tail_order.size(), capacity);
    for (int i = 0; i < used; ++i) {
        const int slot = m->tail_order[(size_t)i];
        HIP_TRY(hipEventSynchronize(m->tev1[slot]));
        HIP_TRY(hipEventElapsedTime(&h_ms[i], m->tev0[slot], m->tev1[slot]));
    }
    *n_out = used;
    m->tail_order.clear();
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

namespace {
// q-tables of `count` motifs (<= kQJobs): three launches, blockIdx.y = motif
int launch_qtables(const gfm_motif_t *motifs, int count, uint64_t *const *d_hist, double threshold, int on_qvalue,
                   double *const *d_qtable, int32_t *const *d_cutoff, uint64_t *const *d_nrows, uint32_t flags,
                   hipStream_t st)
{
    QJobs jobs{};
    jobs.threshold = threshold;
    jobs.on_qvalue = on_qvalue;
    int max_blk = 0;
    for (int k = 0; k < count; ++k) {
        gfm_motif *m = motifs[k];
        QJob &q = jobs.j[k];
        q.hist = reinterpret_cast<const unsigned long long *>(d_hist[k]);
        q.ptable = m->d_ptable;
        int qs = 0;                 // this stream's scratch set on this handle
        while (qs < m->q_streams_used && m->q_stream[qs] != st) ++qs;
        if (qs == m->q_streams_used) {
            if (qs < kQStreams) {
                ++m->q_streams_used;
            } else {                // every set is taken: the one whose stream called longest ago changes hands
                qs = 0;
                for (int c = 1; c < kQStreams; ++c)
                    if (m->q_last_use[c] < m->q_last_use[qs]) qs = c;
            }
            m->q_stream[qs] = st;
        }
        m->q_last_use[qs] = ++m->q_calls;
        q.ws = m->d_qwork + qs;
        // q_raw leaves raw(s) in a table of L doubles: the caller's q-table, or ours when none is asked
        q.qtable = (d_qtable && d_qtable[k]) ? d_qtable[k] : m->d_qscratch + (size_t)qs * (size_t)m->L;
        q.cutoff = d_cutoff ? d_cutoff[k] : nullptr;
        q.nrows = d_nrows ? reinterpret_cast<unsigned long long *>(d_nrows[k]) : nullptr;
        q.clear = (flags & GFM_FLAG_CLEAR_HIST) ? reinterpret_cast<unsigned long long *>(d_hist[k]) : nullptr;
        q.L = m->L; q.lo = m->lo; q.hi = m->hi; q.min_val = m->min_val;
        q.nblk = (m->nb + kQThreads - 1) / kQThreads;   // <= 251 for W <= 64
        max_blk = std::max(max_blk, q.nblk);
    }
    const dim3 grid((unsigned)max_blk, (unsigned)count);
    hipLaunchKernelGGL(q_count_kernel, grid, dim3(kQThreads), 0, st, jobs);
    hipLaunchKernelGGL(q_raw_kernel, grid, dim3(kQThreads), 0, st, jobs);
    hipLaunchKernelGGL(q_final_kernel, grid, dim3(kQThreads), 0, st, jobs);
    HIP_TRY(hipGetLastError());
    return GFM_OK;
}
}  // namespace

GFM_API int gfm_qvalue_table(gfm_motif_t m, uint64_t *d_hist, double threshold, int on_qvalue,
                             double *d_qtable, int32_t *d_cutoff, uint64_t *d_nrows, uint32_t flags,
                             void *stream)
{
    if (!m || !d_hist) return fail(GFM_ERR_INVALID, "NULL argument");
    return launch_qtables(&m, 1, &d_hist, threshold, on_qvalue, &d_qtable, &d_cutoff, &d_nrows, flags,
                          static_cast<hipStream_t>(stream));
}

GFM_API int gfm_qvalue_table_multi(const gfm_motif_t *motifs, int n_motifs, uint64_t *const *d_hist,
                                   double threshold, int on_qvalue, double *const *d_qtable,
                                   int32_t *const *d_cutoff, uint64_t *const *d_nrows, uint32_t flags, void *stream)
{
    if (!motifs || n_motifs < 1 || !d_hist) return fail(GFM_ERR_INVALID, "NULL argument");
    for (int i = 0; i < n_motifs; ++i) {
        if (!motifs[i] || !d_hist[i]) return fail(GFM_ERR_INVALID, "motif or histogram %d is NULL", i);
        for (int j = 0; j < i; ++j)   // the passes of one motif share its scratch (QWork)
            if (motifs[j] == motifs[i]) return fail(GFM_ERR_INVALID, "motif %d listed twice", i);
    }
    for (int i = 0; i < n_motifs; i += kQJobs) {
        const int count = std::min(kQJobs, n_motifs - i);
        const int rc = launch_qtables(motifs + i, count, d_hist + i, threshold, on_qvalue,
                                      d_qtable ? d_qtable + i : nullptr, d_cutoff ? d_cutoff + i : nullptr,
                                      d_nrows ? d_nrows + i : nullptr, flags, static_cast<hipStream_t>(stream));
        if (rc) return rc;
    }
    return GFM_OK;
}

namespace {
// The selection workspace of a handle is one set.  A call on another stream than the last one waits for that one's
// selection (event); inside a stream capture nothing is recorded or waited for (one stream: its order is enough).
int order_selection(gfm_motif *m, hipStream_t st, bool before)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap != hipStreamCaptureStatusNone) return GFM_OK;
    if (before) {
        if (m->sel_valid && m->sel_last_stream != st) HIP_TRY(hipStreamWaitEvent(st, m->ev_selected, 0));
    } else {
        HIP_TRY(hipEventRecord(m->ev_selected, st));
        m->sel_last_stream = st;
        m->sel_valid = true;
    }
    return GFM_OK;
}
}  // namespace

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
    const int slot = (int)(m->sel_call_no++ % (unsigned)kCtlSlots);
    {
        const int rc0 = order_selection(m, st, true);
        if (rc0) return rc0;
    }
    hipLaunchKernelGGL(select_hits_kernel, dim3((unsigned)blocks), dim3(kSelThreads), 0, st, d_scores,
                       (long long)n, d_cutoff, (long long)row_base,
                       reinterpret_cast<long long *>(d_hit_rows), (long long)hit_capacity,
                       (flags & GFM_FLAG_RESET_HITS)
                           ? nullptr
                           : reinterpret_cast<const unsigned long long *>(d_hit_count),
                       m->d_sel_ctl, slot, m->d_sel_resid, m->d_sel_resid_n,
                       static_cast<const unsigned long long *>(nullptr), 0ll);
    HIP_TRY(hipGetLastError());
    const int rc = launch_post(m, nullptr, 0, nullptr, 0, 0, nullptr, m->d_sel_resid, m->d_sel_resid_n, (int)blocks,
                               m->d_sel_ctl, slot, reinterpret_cast<long long *>(d_hit_rows), hit_capacity,
                               reinterpret_cast<unsigned long long *>(d_hit_count), st);
    return rc ? rc : order_selection(m, st, false);
}

GFM_API int gfm_select_hits_from(gfm_motif_t m, const int32_t *d_scores, int64_t n, const int32_t *d_cutoff,
                                 int64_t row_base, const int64_t *d_cand_rows, int64_t cand_capacity,
                                 const uint64_t *d_cand_count, int64_t *d_hit_rows, int64_t hit_capacity,
                                 uint64_t *d_hit_count, void *stream)
{
    if (!m) return fail(GFM_ERR_INVALID, "motif is NULL");
    if (n < 0 || cand_capacity < 0) return fail(GFM_ERR_INVALID, "negative row count or capacity");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n == 0) {
        if (d_hit_count) HIP_TRY(hipMemsetAsync(d_hit_count, 0, sizeof(uint64_t), st));
        return GFM_OK;
    }
    if (!d_cutoff || !d_cand_rows || !d_cand_count || !d_hit_rows || !d_hit_count)
        return fail(GFM_ERR_INVALID, "NULL device buffer");
    if (d_cand_rows == d_hit_rows || reinterpret_cast<const void *>(d_cand_count) == d_hit_count)
        return fail(GFM_ERR_INVALID, "the candidate list and the hit list must be different buffers");
    if ((reinterpret_cast<uintptr_t>(d_scores) & 15u) != 0)
        return fail(GFM_ERR_INVALID, "d_scores must be 16-byte aligned");
    {
        const int rc0 = order_selection(m, st, true);
        if (rc0) return rc0;
    }
    // 1. the candidates that reach the cutoff -> hit list (restarted)
    {
        const long long blocks = std::max<long long>(1, (cand_capacity + kFilterPerBlock - 1) / kFilterPerBlock);
        HIP_TRY(hipMemsetAsync(d_hit_count, 0, sizeof(uint64_t), st));
        hipLaunchKernelGGL(filter_hits_kernel, dim3((unsigned)blocks), dim3(kSelThreads), 0, st,
                           reinterpret_cast<const long long *>(d_cand_rows),
                           reinterpret_cast<const unsigned long long *>(d_cand_count), (long long)cand_capacity, d_cutoff,
                           reinterpret_cast<long long *>(d_hit_rows), (long long)hit_capacity,
                           reinterpret_cast<unsigned long long *>(d_hit_count));
        HIP_TRY(hipGetLastError());
    }
    // 2. the pass over every score, which runs only if the candidate list had overflowed -- and only for a caller that kept
    // the scores (d_scores == NULL: the caller has read *d_cand_count and knows the list is complete)
    if (!d_scores) return order_selection(m, st, false);
    const long long n4 = (n + 3) >> 2;
    long long blocks = (n4 + kSelThreads - 1) / kSelThreads;
    blocks = std::max<long long>(1, std::min<long long>(blocks, m->sel_slabs));
    const int slot = (int)(m->sel_call_no++ % (unsigned)kCtlSlots);
    hipLaunchKernelGGL(select_hits_kernel, dim3((unsigned)blocks), dim3(kSelThreads), 0, st, d_scores,
                       (long long)n, d_cutoff, (long long)row_base, reinterpret_cast<long long *>(d_hit_rows),
                       (long long)hit_capacity, reinterpret_cast<const unsigned long long *>(d_hit_count),
                       m->d_sel_ctl, slot, m->d_sel_resid, m->d_sel_resid_n,
                       reinterpret_cast<const unsigned long long *>(d_cand_count), (long long)cand_capacity);
    HIP_TRY(hipGetLastError());
    const int rc = launch_post(m, nullptr, 0, nullptr, 0, 0, nullptr, m->d_sel_resid, m->d_sel_resid_n, (int)blocks,
                               m->d_sel_ctl, slot, reinterpret_cast<long long *>(d_hit_rows), hit_capacity,
                               reinterpret_cast<unsigned long long *>(d_hit_count), st);
    return rc ? rc : order_selection(m, st, false);
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
        gfm_hit_sort::sort_packed(reinterpret_cast<int64_t *>(rows.data()), rows.size(), GFM_HIT_SCORE_BITS);  // ascending by row
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
