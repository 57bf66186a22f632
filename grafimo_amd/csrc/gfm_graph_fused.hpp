// gfm_graph_fused.hpp -- extraction FUSED into scoring (included by graph_extract.hip inside its anonymous namespace,
// behind the walk machinery it shares: simulate(), DelEmit, count_by_tables / count_by_bitsets).
//
// The materialising path (gfm_graph_plan + gfm_graph_emit + gfm_score_kmers) writes every row `vg find -K` would print --
// k-mer bytes plus 34 bytes of columns -- reads the k-mers back to score them, and counts haplotypes for all of them,
// although score_seqs only ever needs those columns for the rows that survive the threshold (score_sequences.py:273-321,
// resultsTmp.py:303-310: a p < 1e-4 scan keeps one row in ten thousand).  Here a walk's two k-mers live in LDS /
// registers just long enough to be scored: what reaches HBM per plan is the score histogram (for the q-values, which
// the reference computes over ALL rows) and one 16-byte entry per hit.  graph_annotate_kernel then re-derives the
// columns -- coordinates, vg's ref flag, the bases, the haplotype count -- for the hit rows only.
//
// Work decomposition: the host cuts the regions into TILES of <= 256 consecutive window starts (one region each) and
// finds each tile's first site; a persistent grid of workgroups takes tiles by ticket.  Per tile everything the
// windows read -- site records (with their alternate bases packed in), the reference bases -- is staged in LDS by
// coalesced loads ONCE; the materialising kernels found the same things through a chain of four dependent global round
// trips per wave.  Phase 1, thread per window: first site, number of walks, the reference window's score on both
// strands (a walk differs from it at its alternate alleles only); windows that touch an insertion / deletion are
// listed for graph_score_del_kernel.  Block scan of the walk counts.  Phase 2, thread per walk: the mixed-radix
// digits of its rank, the score adjusted per alternate allele, histogram in an LDS window, hits appended.
//
// Scores: one LDS table of packed entries tab[j][code] = sm[code][j] | sm[comp(code)][W-1-j] << 16 (code =
// (ascii >> 1) & 7: A 0, C 1, T 2, G 3; 4..7 = N and the like), so ONE lookup per base serves both strands: the
// reverse complement holds comp(base j) at position W-1-j.  A k-mer with an invalid code scores min_val on both
// strands (score_sequences.py:376-378).  Sums stay below 2^16 per half (<= 64 x 1000), and packed adds / subtracts are
// exact modulo 2^32 as long as the final halves are, which they are: they are scores.

struct FusedTab { unsigned v[GFM_MAX_WIDTH * 8]; };
struct GraphHit { int w, score; long long q2; };      // window of the call, scaled score, walk * 2 + strand (0 '+', 1 '-')
struct Tile {
    long long p0, limit;      // first window start, end of the region (a walk must end inside it)
    int n_win, region;        // windows p0 .. p0 + n_win - 1
    int i_lo, i_hi;           // sites [i_lo, i_hi): pos >= p0 - 1 ... pos < p0 + n_win - 1 + W
    int w_base, pad;          // index of the tile's first window among the call's windows
};
struct DelWin { long long p, limit; int w, i0, region, pad; };   // a listed window for graph_score_del_kernel
struct HitRec {               // what graph_annotate_kernel writes per hit (120 bytes; numpy dtype in extract_regions.py)
    long long start, stop, freq, q2;
    double qvalue;
    int w, score, region;
    unsigned char strand, is_ref, keep, pad;
    unsigned char kmer[GFM_MAX_WIDTH];
};

constexpr long long kFusedMaxWalks = 1ll << 40;      // per window; beyond it the product of allele counts is refused
constexpr int kFusedThreads = 256;                   // = windows per tile
constexpr int kFusedSites = 448;                     // site records staged per tile (more: read from global memory)
constexpr int kFusedRefBytes = kFusedThreads + GFM_MAX_WIDTH + 8;
constexpr int kFusedDelThreads = 128;
constexpr int kFusedLayouts = 8;

struct FusedArgs {
    int W, forward_only;
    int min_val, cutoff;          // cutoff: rows with score >= cutoff are hits (INT32_MAX: none)
    int hlo, hnb;                 // LDS histogram window [hlo, hlo + hnb) + one bin for min_val; hnb = 0: no histogram
    unsigned long long *hist;     // [L] spill target (scores outside the window) or nullptr
    unsigned *slabs;              // [gridDim.x][hnb + 1]
    GraphHit *hits;
    long long hit_cap;
    unsigned long long *hit_count, *n_rows;
    unsigned *ticket;
};

__device__ __forceinline__ unsigned base_code(unsigned c) { return (c >> 1) & 7u; }

// site record with the alternate bases packed into n_alts' upper bytes (one LDS read instead of two global ones)
__device__ __forceinline__ SiteRec packed_site(const GraphDev &g, int i)
{
    SiteRec r = g.site_rec[i];
    if (i < g.n_sites) {
        const uint8_t *a = g.alt_bases + (size_t)i * kMaxAlts;
        r.n_alts |= ((int)a[0] << 8) | ((int)a[1] << 16) | ((int)a[2] << 24);
    }
    return r;
}
struct TileSites {
    const GraphDev &g;
    const SiteRec *lds;
    int i_lo, staged;
    __device__ __forceinline__ SiteRec at(int i) const
    {
        const unsigned d = (unsigned)(i - i_lo);
        return d < (unsigned)staged ? lds[d] : packed_site(g, i);
    }
};

// hits of one wave: one returning atomic per wave that holds any (a p < 1e-4 scan: a few hundred per plan)
__device__ __forceinline__ void push_hits(const FusedArgs &a, bool hit, int w, long long q2, int score)
{
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
    if (mask == 0ull) return;
    const int lane = threadIdx.x & 63;
    unsigned long long base = 0;
    if (lane == __builtin_ctzll(mask)) base = atomicAdd(a.hit_count, (unsigned long long)__popcll(mask));
    base = ((unsigned long long)__shfl((int)(base >> 32), __builtin_ctzll(mask)) << 32) |
           (unsigned)__shfl((int)(base & 0xffffffffull), __builtin_ctzll(mask));
    if (hit) {
        const unsigned long long at = base + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
        if (at < (unsigned long long)a.hit_cap) a.hits[at] = GraphHit{w, score, q2};
    }
}

__device__ __forceinline__ void book_score(const FusedArgs &a, unsigned *h, int s)
{
    const int d = s - a.hlo;
    if ((unsigned)d < (unsigned)a.hnb) atomicAdd(&h[d], 1u);
    else if (s == a.min_val) atomicAdd(&h[a.hnb], 1u);
    else atomicAdd(&a.hist[s], 1ull);
}

// one window as phase 1 / the annotate kernel see it
struct WinInfo { int i0, ns; long long walks; bool listed; };
template <class S>
__device__ __forceinline__ WinInfo classify_window(const GraphDev &g, const S &sites, long long p, int W, long long limit,
                                                   int i_lo, int i_hi)
{
    int lo = i_lo, hi = i_hi;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sites.at(mid).pos < p) lo = mid + 1; else hi = mid;
    }
    WinInfo w{lo, 0, 1, false};
    bool over = false;
    w.listed = g.n_dels > 0 && covered_by_deletion(g, p, lo);
    for (int i = lo;; ++i, ++w.ns) {
        const SiteRec r = sites.at(i);
        if (r.pos >= p + W) break;
        if (r.del_len || r.ins_len) w.listed = true;
        if (!over) {
            w.walks *= 1 + (r.n_alts & 3);
            over = w.walks > kFusedMaxWalks;
        }
    }
    if (g.n_ins > 0) {
        for (int k = lo - 1; k >= i_lo; --k) {
            const SiteRec r = sites.at(k);
            if (r.pos != p - 1) break;
            if (r.ins_len > 0) w.listed = true;
        }
        if (!w.listed && p + W > limit) { w.walks = 0; over = false; }
    }
    if (p + W > g.ref_len && !w.listed) w.walks = 0;
    if (w.listed) w.walks = 0;
    else if (over) w.walks = -1;          // refused: more than 2^40 walks
    return w;
}

__global__ void __launch_bounds__(kFusedThreads)
graph_score_kernel(GraphDev g, FusedArgs a, FusedTab tab_arg, const Tile *__restrict__ tiles, int n_tiles,
                   DelWin *__restrict__ del_wins, int *__restrict__ del_count, int *__restrict__ overflow)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds[];
    // LDS: table | site records | reference bytes | per-window arrays | block scalars | histogram window
    unsigned *tab = reinterpret_cast<unsigned *>(fused_lds);
    SiteRec *s_rec = reinterpret_cast<SiteRec *>(tab + GFM_MAX_WIDTH * 8);
    unsigned char *s_ref = reinterpret_cast<unsigned char *>(s_rec + kFusedSites);
    long long *w_incl = reinterpret_cast<long long *>(s_ref + kFusedRefBytes);
    int *w_i0 = reinterpret_cast<int *>(w_incl + kFusedThreads);
    int *w_ns = w_i0 + kFusedThreads;
    unsigned *w_score = reinterpret_cast<unsigned *>(w_ns + kFusedThreads);
    int *w_bad = reinterpret_cast<int *>(w_score + kFusedThreads);
    long long *wave_tot = reinterpret_cast<long long *>(w_bad + kFusedThreads);     // [4]
    int *blk = reinterpret_cast<int *>(wave_tot + 4);                               // [0] tile, [1] del base, [2] del count
    unsigned *h = reinterpret_cast<unsigned *>(blk + 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = a.W;
    for (int i = tid; i < W * 8; i += kFusedThreads) tab[i] = tab_arg.v[i];
    for (int i = tid; i <= a.hnb && a.hnb > 0; i += kFusedThreads) h[i] = 0u;
    unsigned long long rows_done = 0;
    for (;;) {
        __syncthreads();                       // everybody is done with the previous tile's LDS
        if (tid == 0) { blk[0] = (int)atomicAdd(a.ticket, 1u); blk[2] = 0; }
        __syncthreads();
        const int ti = blk[0];
        if (ti >= n_tiles) break;
        const Tile t = tiles[ti];
        const int staged = min(t.i_hi - t.i_lo + 1, kFusedSites);      // (+1: the record that ends a window's site scan)
        for (int i = tid; i < staged; i += kFusedThreads) s_rec[i] = packed_site(g, t.i_lo + i);
        {
            const long long span = min((long long)t.n_win + W - 1, g.ref_len - t.p0);
            for (int i = tid; i < kFusedRefBytes; i += kFusedThreads)
                s_ref[i] = i < span ? g.ref[t.p0 + i] : (unsigned char)'N';
        }
        __syncthreads();
        const TileSites sites{g, s_rec, t.i_lo, staged};
        // ---- phase 1: thread per window
        long long walks = 0;
        bool listed = false;
        WinInfo wi{0, 0, 0, false};
        const long long p = t.p0 + tid;
        if (tid < t.n_win) {
            wi = classify_window(g, sites, p, W, t.limit, t.i_lo, t.i_hi);
            listed = wi.listed;
            if (wi.walks < 0) { atomicMax(overflow, 1); wi.walks = 0; }
            walks = wi.walks;
            if (walks > 0) {                   // the reference window's score on both strands
                unsigned sum = 0;
                int bad = 0;
                for (int j = 0; j < W; ++j) {
                    const unsigned c = base_code(s_ref[tid + j]);
                    sum += tab[j * 8 + c];
                    bad += (int)(c >> 2);
                }
                w_score[tid] = sum;
                w_bad[tid] = bad;
            }
            w_i0[tid] = wi.i0;
            w_ns[tid] = wi.ns;
        }
        // listed windows -> graph_score_del_kernel (one global atomic per tile)
        {
            const unsigned long long lm = __builtin_amdgcn_ballot_w64(listed);
            int at = 0;
            if (lm) {
                if (lane == 0) at = atomicAdd(&blk[2], __popcll(lm));
                at = __shfl(at, 0) + __popcll(lm & ((1ull << lane) - 1ull));
            }
            __syncthreads();
            if (tid == 0 && blk[2] > 0) blk[1] = atomicAdd(del_count, blk[2]);
            __syncthreads();
            if (listed) del_wins[blk[1] + at] = DelWin{p, t.limit, t.w_base + tid, wi.i0, t.region, 0};
        }
        // inclusive scan of the walk counts over the block
        long long incl = walks;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int lo_ = __shfl_up((int)(incl & 0xffffffffll), d), hi_ = __shfl_up((int)(incl >> 32), d);
            if (lane >= d) incl += ((long long)hi_ << 32) | (unsigned)lo_;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        long long before = 0, total = 0;
        for (int k = 0; k < kFusedThreads / 64; ++k) {
            if (k < wave) before += wave_tot[k];
            total += wave_tot[k];
        }
        w_incl[tid] = incl + before;
        __syncthreads();
        // ---- phase 2: thread per walk
        for (long long base = 0; base < total; base += kFusedThreads) {
            const long long wt = base + tid;
            const bool live = wt < total;
            int k = 0;
            long long q = 0;
            int s_f = 0, s_r = 0;
            if (live) {
                int lo = 0, hi = t.n_win - 1;                  // first window whose inclusive count exceeds wt
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (w_incl[mid] > wt) hi = mid; else lo = mid + 1;
                }
                k = lo;
                q = wt - (k ? w_incl[k - 1] : 0ll);
                unsigned sum = w_score[k];
                int bad = w_bad[k];
                long long rest = q;
                const int i0 = w_i0[k];
                for (int s = w_ns[k] - 1; s >= 0 && rest; --s) {          // digits, last site first
                    const SiteRec r = sites.at(i0 + s);
                    const int nall = 1 + (r.n_alts & 3);
                    const int al = (int)(rest % nall);
                    rest /= nall;
                    if (al) {
                        const int j = r.pos - (int)(t.p0 + k);
                        const unsigned cr = base_code(s_ref[k + j]), ca = base_code((unsigned)r.n_alts >> (8 * al));
                        sum += tab[j * 8 + ca] - tab[j * 8 + cr];
                        bad += (int)(ca >> 2) - (int)(cr >> 2);
                    }
                }
                s_f = bad ? a.min_val : (int)(sum & 0xffffu);
                s_r = bad ? a.min_val : (int)(sum >> 16);
                if (a.hnb > 0) {
                    book_score(a, h, s_f);
                    if (!a.forward_only) book_score(a, h, s_r);
                }
            }
            push_hits(a, live && s_f >= a.cutoff, t.w_base + k, 2 * q, s_f);
            if (!a.forward_only) push_hits(a, live && s_r >= a.cutoff, t.w_base + k, 2 * q + 1, s_r);
        }
        if (tid == 0) rows_done += (unsigned long long)total * (a.forward_only ? 1ull : 2ull);
    }
    __syncthreads();
    if (a.hnb > 0)
        for (int i = tid; i <= a.hnb; i += kFusedThreads) a.slabs[(size_t)blockIdx.x * (a.hnb + 1) + i] = h[i];
    if (tid == 0 && rows_done) atomicAdd(a.n_rows, rows_done);
}

// visitor of simulate(): the bases of a walk into a k-mer slot (alternate / inserted bases at once, reference bases
// noted and fetched together afterwards); nothing about haplotypes -- the count is the annotate kernel's business
struct ScoreEmit {
    const uint8_t *alt_bases, *ins_bases;
    const int *ins_off;
    uint8_t *fwd;
    int *src;
    static constexpr bool kWantsBases = true;
    __device__ void base(int j, long long x, int snp, int a, int)
    {
        src[j] = (int)x;
        if (snp >= 0 && a) { fwd[j] = alt_bases[(size_t)snp * kMaxAlts + (a - 1)]; src[j] = -1; }
    }
    __device__ void ins_base(int j, int site, int t) { fwd[j] = ins_bases[ins_off[site] + t]; src[j] = -1; }
    __device__ void took(int) {}
    __device__ void passed(int) {}
};

struct FusedLayout { long long cum_end; unsigned choice; int site, t, pad; };

// The walks of the listed windows (they touch an insertion or a deletion: the sites a walk meets depend on its
// decisions).  Workgroup per 128 listed windows: thread per window enumerates its layouts (the odometer over
// simulate()) and keeps the first eight; block scan of the walk counts; then thread per walk: its layout by
// comparison, one replay that writes the bases, the reference bases fetched eight at a time, both strands scored from
// the slot.  (The materialising path does the same in three kernels with a device-wide scan and a host read-back
// between them.)
__global__ void __launch_bounds__(kFusedDelThreads)
graph_score_del_kernel(GraphDev g, FusedArgs a, FusedTab tab_arg, const DelWin *__restrict__ del_wins,
                       const int *__restrict__ del_count, int *__restrict__ overflow, int pitch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds[];
    constexpr int T = kFusedDelThreads;
    unsigned *tab = reinterpret_cast<unsigned *>(fused_lds);
    SiteRec *cache = reinterpret_cast<SiteRec *>(tab + GFM_MAX_WIDTH * 8);                 // [kSiteCache][T]
    FusedLayout *lay = reinterpret_cast<FusedLayout *>(cache + kSiteCache * T);            // [T][kFusedLayouts]
    long long *w_incl = reinterpret_cast<long long *>(lay + T * kFusedLayouts);            // [T]
    long long *wave_tot = w_incl + T;                                                      // [2]
    int *n_lay = reinterpret_cast<int *>(wave_tot + 2);                                    // [T]
    int *src = n_lay + T;                                                                  // [T][W]
    unsigned char *slots = reinterpret_cast<unsigned char *>(src + (size_t)T * a.W);       // [T][pitch]
    unsigned *h = reinterpret_cast<unsigned *>(slots + (size_t)T * pitch);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = a.W;
    for (int i = tid; i < W * 8; i += T) tab[i] = tab_arg.v[i];
    for (int i = tid; i <= a.hnb && a.hnb > 0; i += T) h[i] = 0u;
    const int n_listed = *del_count;
    unsigned long long rows_done = 0;
    for (int batch = blockIdx.x; (long long)batch * T < n_listed; batch += gridDim.x) {
        __syncthreads();
        const int m = batch * T + tid;
        const bool have = m < n_listed;
        DelWin dw{0, 0, 0, 0, 0, 0};
        long long walks = 0;
        int nl = 0;
        if (have) {
            dw = del_wins[m];
#pragma unroll
            for (int k = 0; k < kSiteCache; ++k) cache[k * T + tid] = g.site_rec[dw.i0 + k];
            const CachedSites sites{g.site_rec, cache + tid, dw.i0, T};
            WalkState st;
            NoVisitor nv;
            WalkStart ws;
            bool bad = false;
            do {
                int prefix = 0;
                do {
                    long long prod = 0;
                    const int rc = simulate<NoVisitor, CachedSites, kFusedMaxWalks>(g, sites, dw.p, W, dw.i0, ws, prefix, st, nv, 0, 0,
                                                                                   prod, dw.limit);
                    if (rc == WALK_OK) {
                        walks += prod;
                        if (nl < kFusedLayouts)
                            lay[tid * kFusedLayouts + nl] = FusedLayout{walks, ((unsigned)st.nd << 24) | (st.choice & ((1u << st.nd) - 1u)),
                                                                        ws.site, ws.t, 0};
                        ++nl;
                    }
                    if (rc == WALK_OVERFLOW || walks > kFusedMaxWalks) { bad = true; break; }
                    prefix = next_walk(st);
                } while (prefix >= 0);
            } while (!bad && next_start(g, dw.p, dw.i0, ws));
            if (bad) { walks = 0; nl = 0; atomicMax(overflow, 1); }
        }
        n_lay[tid] = nl;
        long long incl = walks;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int lo_ = __shfl_up((int)(incl & 0xffffffffll), d), hi_ = __shfl_up((int)(incl >> 32), d);
            if (lane >= d) incl += ((long long)hi_ << 32) | (unsigned)lo_;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        const long long total = wave_tot[0] + wave_tot[1];
        w_incl[tid] = incl + (wave ? wave_tot[0] : 0ll);
        __syncthreads();
        for (long long base = 0; base < total; base += T) {
            const long long wt = base + tid;
            const bool live = wt < total;
            int s_f = 0, s_r = 0, w_id = 0;
            long long q0 = 0;
            if (live) {
                int lo = 0, hi = T - 1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (w_incl[mid] > wt) hi = mid; else lo = mid + 1;
                }
                const int o = lo;                                   // the thread that owns the walk's window
                q0 = wt - (o ? w_incl[o - 1] : 0ll);
                const DelWin ow = del_wins[batch * T + o];
                w_id = ow.w;
                const CachedSites sites{g.site_rec, cache + o, ow.i0, T};
                WalkState st;
                WalkStart ws;
                long long q = q0, prod = 0;
                bool found = false;
                {
                    long long lbase = 0;
                    const int nlo = min(n_lay[o], kFusedLayouts);
                    for (int k = 0; k < nlo && !found; ++k) {
                        const FusedLayout rec = lay[o * kFusedLayouts + k];
                        if (q0 < rec.cum_end) {
                            found = true;
                            q = q0 - lbase;
                            prod = rec.cum_end - lbase;
                            st.nd = (int)(rec.choice >> 24);
                            st.choice = rec.choice & 0xffffffu;
                            ws.site = rec.site;
                            ws.t = rec.t;
                        }
                        lbase = rec.cum_end;
                    }
                }
                if (!found) {                                       // beyond the kept layouts: the odometer from the start
                    NoVisitor nv;
                    bool more = true;
                    while (!found && more) {
                        int prefix = 0;
                        for (;;) {
                            const int rc = simulate<NoVisitor, CachedSites, kFusedMaxWalks>(g, sites, ow.p, W, ow.i0, ws, prefix, st, nv,
                                                                                           0, 0, prod, ow.limit);
                            if (rc == WALK_OK) {
                                if (q < prod) { found = true; break; }
                                q -= prod;
                            }
                            prefix = next_walk(st);
                            if (prefix < 0) break;
                        }
                        if (!found) more = next_start(g, ow.p, ow.i0, ws);
                    }
                }
                unsigned char *slot = slots + (size_t)tid * pitch;
                int *my_src = src + (size_t)tid * W;
                if (found) {
                    ScoreEmit em{g.alt_bases, g.ins_bases, g.ins_off, slot, my_src};
                    long long again = 0;
                    simulate<ScoreEmit, CachedSites, kFusedMaxWalks>(g, sites, ow.p, W, ow.i0, ws, st.nd, st, em, q, prod, again, ow.limit);
                    for (int j0 = 0; j0 < W; j0 += 8) {
                        int sx[8];
                        uint8_t c[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) sx[u] = j0 + u < W ? my_src[j0 + u] : -1;
#pragma unroll
                        for (int u = 0; u < 8; ++u) c[u] = sx[u] >= 0 ? g.ref[sx[u]] : (uint8_t)0;
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (sx[u] >= 0) slot[j0 + u] = c[u];
                    }
                    unsigned sum = 0;
                    int bad = 0;
                    for (int j = 0; j < W; ++j) {
                        const unsigned c = base_code(slot[j]);
                        sum += tab[j * 8 + c];
                        bad += (int)(c >> 2);
                    }
                    s_f = bad ? a.min_val : (int)(sum & 0xffffu);
                    s_r = bad ? a.min_val : (int)(sum >> 16);
                    if (a.hnb > 0) {
                        book_score(a, h, s_f);
                        if (!a.forward_only) book_score(a, h, s_r);
                    }
                }
            }
            push_hits(a, live && s_f >= a.cutoff, w_id, 2 * q0, s_f);
            if (!a.forward_only) push_hits(a, live && s_r >= a.cutoff, w_id, 2 * q0 + 1, s_r);
        }
        if (tid == 0) rows_done += (unsigned long long)total * (a.forward_only ? 1ull : 2ull);
    }
    __syncthreads();
    if (a.hnb > 0)
        for (int i = tid; i <= a.hnb; i += T) a.slabs[(size_t)blockIdx.x * (a.hnb + 1) + i] = h[i];
    if (tid == 0 && rows_done) atomicAdd(a.n_rows, rows_done);
}

// histogram slabs of both kernels -> the caller's histogram (thread per bin; the extra bin is min_val's)
__global__ void __launch_bounds__(256)
graph_hist_reduce_kernel(const unsigned *__restrict__ slabs, int n_slabs, int hlo, int hnb, int min_val,
                         unsigned long long *__restrict__ hist)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b > hnb) return;
    unsigned long long sum = 0;
    for (int s = 0; s < n_slabs; ++s) sum += slabs[(size_t)s * (hnb + 1) + b];
    if (sum) atomicAdd(&hist[b < hnb ? hlo + b : min_val], sum);
}

// ---- the columns of the hit rows.  Thread per hit: the window from the tile table, then what the materialising emit
// kernels do for every row -- for a plain window the mixed-radix digits, the bases, the count from the tables (or the
// bitsets, in place: these are a few hundred threads); for a listed window the odometer up to the walk's rank and one
// replay with the visitor that collects the haplotype constraints.
__global__ void __launch_bounds__(64)
graph_annotate_kernel(GraphDev g, const int *__restrict__ allele_count, int W, const Tile *__restrict__ tiles, int n_tiles,
                      const GraphHit *__restrict__ hits, const unsigned long long *__restrict__ hit_count, long long hit_cap,
                      const int *__restrict__ d_cutoff, const double *__restrict__ qtable, HitRec *__restrict__ out)
{
    const long long n = min((long long)*hit_count, hit_cap);
    const long long hi_ = (long long)blockIdx.x * 64 + threadIdx.x;
    if (hi_ >= n) return;
    const GraphHit hit = hits[hi_];
    HitRec rec{};
    rec.w = hit.w;
    rec.score = hit.score;
    rec.q2 = hit.q2;
    rec.keep = (!d_cutoff || hit.score >= *d_cutoff) ? 1 : 0;
    rec.qvalue = qtable ? qtable[hit.score] : 0.0;
    int lo = 0, hi = n_tiles - 1;                      // last tile with w_base <= w
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tiles[mid].w_base <= hit.w) lo = mid; else hi = mid - 1;
    }
    const Tile t = tiles[lo];
    const long long p = t.p0 + (hit.w - t.w_base);
    const long long q = hit.q2 >> 1;
    const bool minus = (hit.q2 & 1) != 0;
    rec.region = t.region;
    rec.strand = minus ? '-' : '+';
    if (!rec.keep) { out[hi_] = rec; return; }          // a p < t candidate that the q-value cutoff drops
    uint8_t km[2 * GFM_MAX_WIDTH];
    long long end_pos = p + W, count = 0;
    bool any_alt = false;
    const GlobalSites gs{g.site_rec};
    const WinInfo wi = classify_window(g, gs, p, W, t.limit, t.i_lo, t.i_hi + 1);
    if (!wi.listed) {
        for (int j = 0; j < W; ++j) km[j] = g.ref[p + j];
        unsigned long long dig[2] = {0ull, 0ull};
        long long rest = q;
        for (int s = wi.ns - 1; s >= 0; --s) {
            const int nall = 1 + g.n_alts[wi.i0 + s];
            const unsigned long long al = (unsigned long long)(rest % nall);
            rest /= nall;
            dig[s >> 5] |= al << (2 * (s & 31));
            if (al) km[g.pos[wi.i0 + s] - p] = g.alt_bases[(size_t)(wi.i0 + s) * kMaxAlts + (al - 1)];
        }
        any_alt = (dig[0] | dig[1]) != 0ull;
        auto at = [&](int k, int &site, int &al) { site = wi.i0 + k; al = (int)((dig[k >> 5] >> (2 * (k & 31))) & 3ull); };
        bool done;
        count = count_by_tables(g, allele_count, wi.ns, at, done);
        if (!done) count = count_by_bitsets(g, wi.ns, at);
    } else {
        WalkState st;
        WalkStart ws;
        NoVisitor nv;
        long long rest = q, prod = 0;
        bool found = false, more = true;
        while (!found && more) {
            int prefix = 0;
            for (;;) {
                const int rc = simulate<NoVisitor, GlobalSites, kFusedMaxWalks>(g, gs, p, W, wi.i0, ws, prefix, st, nv, 0, 0, prod, t.limit);
                if (rc == WALK_OK) {
                    if (rest < prod) { found = true; break; }
                    rest -= prod;
                }
                prefix = next_walk(st);
                if (prefix < 0) break;
            }
            if (!found) more = next_start(g, p, wi.i0, ws);
        }
        if (!found) { rec.keep = 0; out[hi_] = rec; return; }      // cannot happen: the score kernel found this walk
        int src[GFM_MAX_WIDTH];
        int more_cons[kMaxConstraints - 4];
        DelEmit em(g, km, km + W, src, W, more_cons);
        long long again = 0;
        simulate<DelEmit, GlobalSites, kFusedMaxWalks>(g, gs, p, W, wi.i0, ws, st.nd, st, em, rest, prod, again, t.limit);
        for (int j = 0; j < W; ++j)
            if (src[j] >= 0) km[j] = g.ref[src[j]];
        if (!(ws.site >= 0 && st.last == p - 1)) for_covering_deletions(g, p, wi.i0, [&](int dsite) { em.add(dsite, 0); });
        auto at = [&](int k, int &site, int &al) { const int v = em.get(k); site = v >> 4; al = v & 3; };
        bool done;
        count = count_by_tables(g, allele_count, em.n_cons, at, done);
        if (!done) count = count_by_bitsets(g, em.n_cons, at);
        end_pos = st.last + 1;
        any_alt = em.alt;
    }
    rec.freq = count;
    rec.is_ref = any_alt ? 0 : 1;
    rec.start = minus ? end_pos : p;
    rec.stop = minus ? p : end_pos;
    for (int j = 0; j < W; ++j) rec.kmer[j] = minus ? complement(km[W - 1 - j]) : km[j];
    out[hi_] = rec;
}
