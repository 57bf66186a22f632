/*
 * grafimo_hip.h -- C ABI of libgrafimo_hip.so, the MI355X (gfx950) implementation of
 * GRAFIMO's k-mer scoring hot path.
 *
 * GRAFIMO has no FFI of its own for this path: its native pieces are a Cython module
 * (`motif_processing`) and a numba-jitted function.  Each entry point below names the
 * reference interface it replaces (paths relative to /root/reference/src/grafimo/);
 * INTEGRATION.md shows the ctypes stubs a GRAFIMO maintainer would add to call them.
 *
 * Conventions
 *  - plain C types only; matrices are row-major [4][W] with rows A,C,G,T;
 *    L = 1000*W + 1 is the length of every per-score table (RANGE=1000, utils.py:26);
 *  - every function returns GFM_OK (0) or a negative GFM_ERR_* code and never throws;
 *    gfm_last_error() returns a thread-local message for the last failure;
 *  - pointers named h_* are host memory, d_* are device memory of the current HIP device
 *    (e.g. torch tensors' data_ptr()); the library never keeps a caller pointer after return
 *    (one exception, by its purpose: gfm_graph_hit_columns_start, until its _wait has returned);
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Functions that take
 *    a stream only enqueue work (no allocation, no synchronisation: graph-capturable);
 *  - HIP is initialised lazily by the first call that needs a device, never at load time,
 *    so a host that forks workers later (extract_regions.py:128) stays legal.
 */
#ifndef GRAFIMO_HIP_H
#define GRAFIMO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFM_ABI_VERSION 12

#define GFM_OK 0
#define GFM_ERR_INVALID (-1)  /* bad argument (NULL, width out of range, ...)            */
#define GFM_ERR_ASSERT (-2)   /* a reference `assert` would fire (bg<=0, prob<=0, ...)   */
#define GFM_ERR_HIP (-3)      /* HIP runtime failure, text in gfm_last_error()           */
#define GFM_ERR_NOMEM (-4)
#define GFM_ERR_NODEVICE (-5) /* no usable gfx950 device                                 */
#define GFM_ERR_IO (-6)       /* file could not be read / malformed row                  */
#define GFM_ERR_OVERFLOW (-7) /* an output buffer was too small                          */

#define GFM_MAX_WIDTH 64      /* widest motif the kernels are instantiated for           */
#define GFM_NO_SELECT INT32_MAX
/* flags */
#define GFM_FLAG_RESET_HITS 1u /* start the hit list at 0 instead of appending at *d_hit_count */
#define GFM_FLAG_CLEAR_HIST 2u /* gfm_qvalue_table: zero the histogram after reading it        */
/* gfm_score_kmers with a tail stream: the caller guarantees (by its own stream order, or because the
 * host has seen it complete) that the tail work of the call GFM_WORKSPACE_RING (four) before this one on the same handle
 * has finished -- the handle's scoring workspace is a ring of that many -- so the library need not make the main
 * stream wait for it (an event wait is a barrier packet in front of the score kernel: several
 * microseconds of an idle GPU per step).  The library honours the flag only while the GFM_WORKSPACE_RING calls before this one
 * on the handle were made with the same (stream, tail_stream) pair -- a second user of the handle in between (another
 * scanner, a batched call) makes it fall back to its own event wait. */
#define GFM_FLAG_CALLER_ORDERS_REUSE 4u
#define GFM_WORKSPACE_RING 4
/* a hit-list entry packs the global row id and the row's scaled score:
 * entry = (row << GFM_HIT_SCORE_BITS) | score   (score <= 1000*64 < 2^20) */
#define GFM_HIT_SCORE_BITS 20

/* ------------------------------------------------------------------ library / device */
int gfm_abi_version(void);
const char *gfm_last_error(void);
int gfm_device_count(int *count);
int gfm_set_device(int ordinal);

/* ------------------------------------------------------------------ motif preprocessing
 * replaces compute_log_odds(probs_matrix, width, bgs, alphabet, nucsmap, debug)
 * (motif_processing.pyx:512-548 -> :444-507; lg2 utils.py:479-493):
 *   out[n][j] = ln(probs[n][j] / bg[n]) * 1.44269504   (host libm, f64). */
int gfm_compute_log_odds(const double *h_probs, int width, const double *h_bg,
                         double *h_logodds_out);

/* replaces scale_pwm(motif_matrix, alphabet, motif_width, nucsmap, debug)
 * (motif_ops.py:1027-1111): integer scaling to [0,1000], round-half-to-even. */
int gfm_scale_pwm(const double *h_logodds, int width, int64_t *h_score_matrix_out,
                  int *min_val, int *max_val, int *scale, double *offset);

/* replaces comp_pval_mat(motif, debug) (motif_processing.pyx:608-632 -> :552-603):
 * Staden-1994 score-distribution DP, run ON DEVICE; h_pmf_out has L doubles
 * (last DP row, un-normalised), bit-identical to the reference's. */
int gfm_comp_pval_mat(const int64_t *h_score_matrix, int width, const double *h_bg,
                      double *h_pmf_out);

/* ------------------------------------------------------------------ device-resident motif
 * The numeric content of a reference `Motif` (motif.py:18) that scoring reads:
 * score_matrix, bg, min_val, scale, offset, width, pval_matrix. */
typedef struct gfm_motif *gfm_motif_t;

/* Uploads the tables to the current device.  min_val must be the minimum of h_score_matrix
 * (Motif.min_val: what a k-mer holding N scores).  If h_pmf is NULL the DP of
 * gfm_comp_pval_mat runs on device; otherwise h_pmf (L doubles) is taken as the motif's
 * pval_matrix.  Also builds p_table[s] = sum(pmf[s:]) / sum(pmf) on device -- the O(1)
 * form of score_sequences.py:390-391 -- and the kernel workspace. */
int gfm_motif_create(const int64_t *h_score_matrix, int width, const double *h_bg,
                     int min_val, int scale, double offset, const double *h_pmf,
                     gfm_motif_t *out);
void gfm_motif_destroy(gfm_motif_t m);
int gfm_motif_width(gfm_motif_t m);
int gfm_motif_table_len(gfm_motif_t m); /* L */
/* lowest / highest reachable scaled score (support of the pmf) */
int gfm_motif_score_range(gfm_motif_t m, int32_t *lo, int32_t *hi);
/* host copies of the device tables (either pointer may be NULL) */
int gfm_motif_tables(gfm_motif_t m, double *h_pmf_out, double *h_ptable_out);
/* smallest scaled score s with p_table[s] < threshold (strict, resultsTmp.py:303-307);
 * L if none. */
int gfm_motif_pvalue_cutoff(gfm_motif_t m, double threshold, int32_t *cutoff);
/* (scaled score) -> log-odds and p-value for a few rows on the host:
 * logodds = s/scale + W*offset (score_sequences.py:393), p = p_table[s]. */
int gfm_motif_annotate(gfm_motif_t m, const int32_t *h_scores, int64_t n,
                       double *h_logodds_out, double *h_pvalue_out);

/* ------------------------------------------------------------------ scoring
 * replaces the per-row body of score_seqs + compute_score_seq
 * (score_sequences.py:273-321, :331-396) for a dense batch of k-mers.
 *   d_kmers    uint8 [n][W], ASCII, row-major, 16-byte aligned base: the KMER column of
 *              the vg TSV rows.  A/a C/c G/g T/t score; a row holding 'N' (or any other
 *              byte whose bits 1-3 do not name A,C,G,T) scores min_val (:376-378).
 *   d_scores   int32 [n] out: scaled integer scores (16-byte aligned for the fastest stores;
 *              4-byte alignment is accepted); NULL: no score is stored -- the histogram and the hit
 *              list are all the caller wants (what the product's scans ask for: with a threshold the
 *              cutoff is known before scoring, so the array would never be read; 4 of the W + 4
 *              bytes a k-mer moves).
 *   d_hist     uint64 [L] in/out or NULL: d_hist[s] += #rows scored s.
 *   select_cutoff / d_hit_*: if select_cutoff != GFM_NO_SELECT, rows with
 *              score >= select_cutoff get the entry ((row_base + row) << 20 | score)
 *              appended to d_hit_rows (unordered) starting at *d_hit_count (at 0 with
 *              GFM_FLAG_RESET_HITS) and *d_hit_count updated; hits beyond hit_capacity are
 *              counted but not stored.
 * Enqueues the score kernel on `stream` and the small kernel that finishes d_hist and the hit
 * list on `tail_stream` (NULL = `stream`), ordered by events inside the library; with a separate
 * tail stream the next score kernel overlaps that tail and d_hist / d_hit_* are complete when
 * `tail_stream` reaches this point.  No host synchronisation. */
int gfm_score_kmers(gfm_motif_t m, const uint8_t *d_kmers, int64_t n, int32_t *d_scores,
                    uint64_t *d_hist, int32_t select_cutoff, int64_t row_base,
                    int64_t *d_hit_rows, int64_t hit_capacity, uint64_t *d_hit_count,
                    uint32_t flags, void *stream, void *tail_stream);

/* Batched form for motifs of ONE width (BASELINE config 5): the k-mer matrix is read once per
 * group of up to three motifs (as many as fit the LDS: tables, per-motif histogram windows and hit
 * queues), so per (k-mer, motif) pair the bytes moved drop from W + 4 to W/M + 4.  Arrays of
 * n_motifs entries; d_hist / select_cutoffs / d_hit_* may be NULL (or hold NULL / GFM_NO_SELECT
 * entries) exactly like the scalar arguments of gfm_score_kmers; d_scores may be NULL or hold only NULL
 * entries (no scores stored for any motif of the call).  Results are identical to
 * n_motifs separate gfm_score_kmers calls.  Single stream. */
int gfm_score_kmers_multi(const gfm_motif_t *motifs, int n_motifs, const uint8_t *d_kmers,
                          int64_t n, int32_t *const *d_scores, uint64_t *const *d_hist,
                          const int32_t *select_cutoffs, int64_t row_base,
                          int64_t *const *d_hit_rows, const int64_t *hit_capacity,
                          uint64_t *const *d_hit_count, uint32_t flags, void *stream);

/* How gfm_score_kmers_multi would group these motifs (no device work): group_size_out[i] = number of motifs of the
 * launch motif i rides in (1..3), waves_out[i] = waves per workgroup of that launch (16 or 8).  with_hist[i] != 0
 * (NULL = all): motif i accumulates a histogram, which is what limits a group (LDS windows).  Either output may be
 * NULL.  Lets a caller (and the parity tests) see which score_quad_kernel<W, MM> instantiation a call uses. */
int gfm_score_kmers_multi_plan(const gfm_motif_t *motifs, int n_motifs, const int32_t *with_hist,
                               int32_t *group_size_out, int32_t *waves_out);

/* Measurement aid (bench.py): with slots > 0 every `every`-th later gfm_score_kmers call
 * brackets the score kernel ALONE (not the post kernel that follows it) with a hipEvent pair
 * on the launch stream, in a ring of `slots` pairs; slots = 0 turns it off.  (The events ride on the
 * kernel's dispatch packet: start and completion of the kernel itself.)  gfm_profile_read waits for the recorded
 * events and returns the kernel durations in ms, oldest first. */
int gfm_profile_enable(gfm_motif_t m, int slots, int every);
int gfm_profile_read(gfm_motif_t m, float *h_ms_out, int capacity, int *n_out);
/* The TAIL of a timed gfm_score_kmers call that was given a tail stream: the library records a start event on the
 * tail stream behind its wait for the score kernel (i.e. when the tail may begin); gfm_profile_mark_tail records the
 * stop event on `stream` -- call it after everything that belongs to the step's tail has been enqueued (post kernel,
 * the all-reduce of the histogram, q-table, hit gather) on the stream the last of it runs on; a no-op when the last
 * call was not a timed one.  gfm_profile_read_tail returns the durations in ms, oldest first. */
int gfm_profile_mark_tail(gfm_motif_t m, void *stream);
int gfm_profile_read_tail(gfm_motif_t m, float *h_ms_out, int capacity, int *n_out);

/* Measurement aid (bench.py `peak_measured`): what THIS device sustains for a bare stream with the score kernel's
 * byte mix -- every lane issues `loads_per_store` 16-byte non-temporal loads per 16-byte store, loads and stores
 * interleaved in one pass, one step prefetched (the shape score_quad_kernel has; W = 19: 4.75 -> 5 loads per store).
 * in_bytes are read per launch (two input buffers used in turn), in_bytes / loads_per_store written, the output
 * rotating over three buffers;
 * store_policy 0 = nt, 1 = write-through (sc0 sc1).  Allocates and frees its own buffers; synchronous.
 * *us_per_launch = average of `launches` launches (HIP events), *bytes_per_launch = bytes read + written. */
int gfm_calibrate_stream(int loads_per_store, int64_t in_bytes, int store_policy, int launches,
                         double *us_per_launch, double *bytes_per_launch);

/* replaces compute_qvalues(pvalues, debug) (score_sequences.py:401-428; statsmodels
 * fdr_bh): Benjamini-Hochberg q-value of every scaled score, from the score histogram
 * of ALL scored rows:  q(s) = min(1, min_{s'<=s, hist[s']>0} p(s') / (C(s')/n)),
 * C(s') = #rows with score >= s'.  Also the selection cutoff: the smallest s with
 * (on_qvalue ? q(s) : p(s)) < threshold.  All outputs are device memory; any may be NULL. */
int gfm_qvalue_table(gfm_motif_t m, uint64_t *d_hist, double threshold, int on_qvalue,
                     double *d_qtable_out, int32_t *d_cutoff_out, uint64_t *d_nrows_out,
                     uint32_t flags, void *stream);
/* The same for n_motifs DISTINCT motifs (any widths) in three launches per eight motifs instead of three per
 * motif: the per-motif compute_qvalues calls of a motif set (score_sequences.py:401-428 once per motif,
 * grafimo.py:181-195).  Arrays of n_motifs entries; d_qtable_out / d_cutoff_out / d_nrows_out may be NULL or hold
 * NULL entries.  Results are identical to n_motifs gfm_qvalue_table calls. */
int gfm_qvalue_table_multi(const gfm_motif_t *motifs, int n_motifs, uint64_t *const *d_hist,
                           double threshold, int on_qvalue, double *const *d_qtable_out,
                           int32_t *const *d_cutoff_out, uint64_t *const *d_nrows_out, uint32_t flags,
                           void *stream);

/* replaces the threshold filter of ResultTmp.to_df (resultsTmp.py:303-307) on device:
 * appends the packed entry of every row with d_scores[row] >= *d_cutoff. */
int gfm_select_hits(gfm_motif_t m, const int32_t *d_scores, int64_t n, const int32_t *d_cutoff,
                    int64_t row_base, int64_t *d_hit_rows, int64_t hit_capacity,
                    uint64_t *d_hit_count, uint32_t flags, void *stream);
/* The same result -- the hit list restarted, holding the packed entry of every row with d_scores[row] >= *d_cutoff --
 * taken from a CANDIDATE list instead of the n scores when that list is complete: d_cand_rows / *d_cand_count is a
 * hit list as the fused selection of gfm_score_kmers left it for a cutoff that is known to be <= *d_cutoff.  With a
 * q-value threshold t (--qvalueT) the cutoff needs the global histogram, but q >= p (BH), so the rows with q < t
 * are among those with p < t: score with select_cutoff = gfm_motif_pvalue_cutoff(t) into the candidate list,
 * build the q-table, then call this -- it reads the few candidates, not every score (1e8 rows: 400 MB).  If the
 * candidate list overflowed (*d_cand_count > cand_capacity) the rows are taken from d_scores as gfm_select_hits
 * does (decided on the device, no host synchronisation).  Candidate and hit buffers must differ.  d_scores may be NULL for a
 * caller that stored no scores (gfm_score_kmers with d_scores == NULL) and has read *d_cand_count <= cand_capacity itself. */
int gfm_select_hits_from(gfm_motif_t m, const int32_t *d_scores, int64_t n, const int32_t *d_cutoff,
                         int64_t row_base, const int64_t *d_cand_rows, int64_t cand_capacity,
                         const uint64_t *d_cand_count, int64_t *d_hit_rows, int64_t hit_capacity,
                         uint64_t *d_hit_count, void *stream);

/* ------------------------------------------------------------------ strand-max and per-region best hit
 * BASELINE.json north_star: "wavefront-level reductions for forward/reverse-strand max ... RCCL only for the final
 * top-hit gather".  The reference has neither: every strand is a row of its own (score_sequences.py:279-321) and the
 * report lists rows.  Its one consumer of "the best hit of a region" is --top-graphs, which takes the first N
 * distinct sequence_names of the table sorted by p-value (res_writer.py:153-157) -- the regions ranked by their best
 * reported hit.  These entry points compute that key on the device and NEVER change the reported rows;
 * grafimo_amd/top_hits.py builds the top-regions table from them and, under torch.distributed, gathers
 * n_regions entries per rank instead of every hit.
 *
 * Rows taking part: score >= max(min_score, *d_cutoff) (d_cutoff may be NULL; the q-value cutoff of gfm_qvalue_table
 * lives on the device), a region id in [0, n_regions), and -- if d_freq is given -- d_freq[row] > 0 (the rows
 * ResultTmp.to_df keeps without --recomb, resultsTmp.py:309-310).
 *
 * gfm_region_best: d_best[r] = max over the rows of region r of  (score << GFM_BEST_ROW_BITS) | (2^44 - 1 - row),
 *   row = row_base + index: the best score, the lowest row among equals; 0 = no such row.  d_best is in/out
 *   (atomic max: batches and chunks accumulate; the caller zeroes it).  d_region int32 [n], any order -- rows of one
 *   region that are contiguous (TSV files, extraction rows) cost one atomic per wavefront instead of one per row.
 * gfm_region_ids: the d_region array of rows laid out region after region, from d_region_off int64 [n_regions + 1]
 *   (rows [off[r], off[r+1]) belong to region r): the TSV scan's files, row_base per file.
 * gfm_locus_max: d_locus_max[i] = the best score among the rows of i's locus -- same region, same
 *   {start, stop} as a set: the reference span of the k-mer whatever its strand ('-' rows carry start > stop,
 *   score_sequences.py:288-291) -- i.e. the forward/reverse-strand maximum at that span; -1 for rows that take no part.
 *   d_work: gfm_locus_max_workspace(n) bytes of scratch, 8-byte aligned; its first 8 bytes hold, afterwards, the number
 *   of rows that found no table slot (0 unless the workspace was smaller than asked for).  Enqueue only. */
#define GFM_BEST_ROW_BITS 44
int gfm_region_ids(const int64_t *d_region_off, int32_t n_regions, int64_t n, int32_t *d_region_out, void *stream);
int gfm_region_best(const int32_t *d_scores, int64_t n, const int32_t *d_region, int32_t n_regions,
                    const int64_t *d_freq, int32_t min_score, const int32_t *d_cutoff, int64_t row_base,
                    uint64_t *d_best, void *stream);
int64_t gfm_locus_max_workspace(int64_t n_rows);
int gfm_locus_max(const int32_t *d_scores, int64_t n, const int32_t *d_region, int32_t n_regions,
                  const int64_t *d_start, const int64_t *d_stop, const int64_t *d_freq, int32_t min_score,
                  const int32_t *d_cutoff, void *d_work, int64_t work_bytes, int32_t *d_locus_max, void *stream);

/* ------------------------------------------------------------------ one-call host form
 * compute_results' numeric core (score_sequences.py:44-211) for host-resident k-mers:
 * H2D, score, histogram, q-table, threshold, D2H of the hits only.
 *   h_kmers uint8 [n][W]; threshold in (0,1]; on_qvalue: threshold applies to q (--qvalueT);
 *   want_qvalues: compute q-values (0 = --no-qvalue);
 *   outputs (caller-allocated, capacity rows each; h_qvalue_out may be NULL):
 *   hit rows ascending by row index with their scaled score, log-odds, p-value, q-value. */
int gfm_scan_host(gfm_motif_t m, const uint8_t *h_kmers, int64_t n, double threshold,
                  int on_qvalue, int want_qvalues, int64_t capacity, int64_t *h_hit_rows_out,
                  int32_t *h_hit_scores_out, double *h_logodds_out, double *h_pvalue_out,
                  double *h_qvalue_out, int64_t *n_hits_out);

/* ------------------------------------------------------------------ TSV ingest (host, C++)
 * replaces the text handling of score_seqs (score_sequences.py:273-293, :305-307):
 * parses vg's 7-column rows  REGION KMER CHR:START(+|-) CHR:STOP(+|-) COUNT ref|non.ref PATH
 * into columnar arrays.  Two calls: gfm_tsv_open counts, gfm_tsv_read fills. */
typedef struct gfm_tsv *gfm_tsv_t;
/* Parses every file (host threads) and reports the number of kept rows.
 * skip_reverse: drop '-' rows before they are counted (--no-reverse, :281-282).
 * n_threads: the reference's `cores` (score_sequences.py:123); <= 0 = every hardware thread.  The library uses
 * at most that many, and never more than one per file, one per MiB of text, or 96; the threads belong to a crew
 * the library keeps for the life of the process (the VCF reader and the streamed scan share it). */
int gfm_tsv_open(const char *const *paths, int n_paths, int width, int skip_reverse,
                 int n_threads, gfm_tsv_t *out, int64_t *n_rows);
/* The first pass of the streamed scan on its own (gfm_scan_tsv_begin reads a file, counts its rows, and parses it in
 * place once every earlier file is counted): the rows gfm_tsv_open would keep of this file -- lines that hold a field,
 * minus the '-' rows with skip_reverse -- without parsing them.  Equal to the parsed count for every file the parser
 * accepts (the CPU tests hold it to that). */
int gfm_tsv_count_rows(const char *path, int skip_reverse, int64_t *n_rows);
/* Copies the parsed columns out (any pointer may be NULL):
 * kmers uint8[n][W]; start/stop int64[n]; strand uint8[n] ('+'/'-'); freq int64[n];
 * is_ref uint8[n] (1 = "ref" after the indel fix :305-307, 0 = "non.ref");
 * file_id int32[n] (index into paths); name_id int32[n] (index into the REGION names). */
int gfm_tsv_read(gfm_tsv_t t, uint8_t *kmers, int64_t *start, int64_t *stop, uint8_t *strand,
                 int64_t *freq, uint8_t *is_ref, int32_t *file_id, int32_t *name_id);
/* distinct REGION strings: count, total bytes, then offsets[count+1] + bytes */
int gfm_tsv_name_count(gfm_tsv_t t);
int64_t gfm_tsv_names_bytes(gfm_tsv_t t);
int gfm_tsv_names(gfm_tsv_t t, int64_t *offsets, char *bytes);
void gfm_tsv_close(gfm_tsv_t t);

/* ------------------------------------------------------------------ streamed scan of a TSV directory
 * compute_results' numeric core (score_sequences.py:113-157, :194-205) as ONE pipelined pass over the
 * width_W TSV files of a motif: host threads parse the files (in path order) while earlier rows are
 * already copied (pinned chunk buffers, hipMemcpyAsync on a copy stream) and scored (one score launch
 * per chunk, one histogram and one hit list for the whole scan, row ids global in path order);
 * the q-value table and the threshold run once at the end and only the hit rows come back.
 * Buffers are kept per device between calls (gfm_scan_release_buffers frees them).
 *   threshold / on_qvalue / want_qvalues as gfm_scan_host; chunk_rows: rows per chunk (<= 0: default);
 *   *n_rows = rows scored (kept rows of all files), *n_hits = rows under the threshold. */
typedef struct gfm_scan *gfm_scan_t;
typedef struct gfm_scan_stats {
    int64_t n_rows, n_hits, n_chunks, h2d_bytes;
    double total_s;      /* wall time of the call                                           */
    double parse_s;      /* until the last file was parsed (copies and kernels overlap it)   */
    double h2d_s;        /* sum of the host-to-device copy durations (hipEvents)             */
    double tail_s;       /* after the last file was parsed: last chunk, tables, hits back    */
    int32_t parse_threads, reserved;
} gfm_scan_stats_t;
int gfm_scan_tsv(gfm_motif_t m, const char *const *paths, int n_paths, int skip_reverse, int n_threads,
                 double threshold, int on_qvalue, int want_qvalues, int64_t chunk_rows, gfm_scan_t *out,
                 int64_t *n_rows, int64_t *n_hits);
/* The same pass in two phases and for several motifs of ONE width (each chunk is scored by gfm_score_kmers_multi: one
 * read of the k-mers per group of up to three motifs) -- what compute_results_many and the sharded scan
 * (grafimo_amd/distributed.py) run: the reference's loop over a motif set (grafimo.py:177-183) repeats ingest and
 * scoring per motif, and its multiprocessing has no second phase because it has no second machine.
 * gfm_scan_tsv_begin parses, uploads and scores; when it returns every motif's histogram is complete on the device
 * (d_hist[j]: the caller's buffers of L uint64 each -- zeroed here -- or NULL for the library's own) and *n_rows rows
 * were scored.  A sharded caller now all-reduces its histogram buffers over the ranks (and synchronises).
 * gfm_scan_tsv_finish derives the q-tables and cutoffs from the histograms as they are then, selects, and brings the
 * hits back: n_hits[j] rows for motif j (n_hits may be NULL).  Row ids are local to this call's files. */
int gfm_scan_tsv_begin(const gfm_motif_t *motifs, int n_motifs, const char *const *paths, int n_paths,
                       int skip_reverse, int n_threads, double threshold, int on_qvalue, int want_qvalues,
                       int64_t chunk_rows, uint64_t *const *d_hist, gfm_scan_t *out, int64_t *n_rows);
int gfm_scan_tsv_finish(gfm_scan_t s, int64_t *n_hits);
/* gfm_scan_hits for motif `motif` of a multi-motif scan */
int gfm_scan_hits_of(gfm_scan_t s, int motif, int64_t *rows, int32_t *scaled, double *logodds, double *pvalue,
                     double *qvalue, uint8_t *kmers, int64_t *start, int64_t *stop, uint8_t *strand,
                     int64_t *freq, uint8_t *is_ref, int32_t *name_id);
/* The hit rows ascending by row id, each with the columns of its TSV row (any pointer may be NULL;
 * n_hits entries each, kmers n_hits x W bytes; qvalue only if the scan computed q-values). */
int gfm_scan_hits(gfm_scan_t s, int64_t *rows, int32_t *scaled, double *logodds, double *pvalue,
                  double *qvalue, uint8_t *kmers, int64_t *start, int64_t *stop, uint8_t *strand,
                  int64_t *freq, uint8_t *is_ref, int32_t *name_id);
int gfm_scan_stats(gfm_scan_t s, gfm_scan_stats_t *out);
/* the parsed table behind the scan (REGION names: gfm_tsv_name_count / gfm_tsv_names); owned by the scan
 * handle -- do not gfm_tsv_close it */
gfm_tsv_t gfm_scan_table(gfm_scan_t s);
void gfm_scan_close(gfm_scan_t s);
void gfm_scan_release_buffers(void);

/* ------------------------------------------------------------------ k-mer extraction
 * Replaces the rows of `vg find -p CHR:S-E -x XG -H GBWT -K W -E` (extract_regions.py:180,225,326;
 * consumed by score_sequences.py:273-307) for graphs made of a linear reference plus SNP sites
 * insertions and deletions (the inputs of `vg construct -r REF -v VCF`, constructVG.py:332).  Rows are written on the device
 * in the layout gfm_score_kmers reads, so extraction -> scoring needs no TSV.  Pinned by the
 * reference's expected_seqs.tsv and by the 704 rows of real vg output in its scoring fixture
 * (SNPs, deletions, node ids and cuts also by vg's own node tables: tests/test_vg_pins.py).  Insertions, multi-base
 * substitutions and complex alleles are part of the graph too; oracle/extract_oracle.py states what no vg output pins
 * about them (coordinates of k-mers that start or end inside inserted bases, row order).
 *
 * gfm_graph_create: h_ref [ref_len] bases; sites ascending 0-based h_pos [n_sites], h_n_alts [n_sites]
 *   in 1..3, h_alt_bases [n_sites][3]; h_del_len [n_sites] (or NULL): 0 for a SNP, else the site is a
 *   deletion of that many bases after the anchor h_pos (one alternate allele; deletions may overlap -- several
 *   lengths at one anchor, anchors inside another deletion's span; at one position the substitution site comes
 *   first, then the insertions, then the deletions); h_alt_bits [n_sites][3][ceil(H/64)] =
 *   haplotypes carrying each alternate allele (bit h of word h/64; deletion carriers in slot 0), or
 *   NULL / H = 0 for no haplotype counts (freq = 0).
 *   Insertions (UNPINNED semantics, stated in oracle/extract_oracle.py): h_ins_len [n_sites] (or NULL): > 0 for a
 *   site that holds that many inserted bases h_ins_bases[h_ins_off[i] ..] behind its anchor h_pos (one alternate
 *   allele, carriers in slot 0).  At one position the substitution site comes first, then any number of
 *   insertions, then the deletion.  A walk that reads inserted bases consumes window bases but no reference span
 *   (stop = coordinate behind the last reference base it used); walks may start inside an insertion anchored at
 *   p - 1 (start = p); with insertions in the graph windows start at p in [S, E - 1] and a walk is kept if its
 *   stop lies inside the region.
 * gfm_graph_plan: regions [S, E] as vg takes them (windows start at p in [S, E - W]); returns the
 *   number of windows and of rows (2 per walk: forward, reverse complement).  Synchronous.  GFM_ERR_OVERFLOW: the rows do not
 *   fit one plan (more than 2^31 of them; a single window of more than 2^30 walks) -- plan fewer windows at a time
 *   (gfm_graph_plan_windows), or use gfm_graph_score, which writes no rows.
 * gfm_graph_emit: rows of the last plan, window-major, walks in mixed-radix order with the LAST site
 *   of the window varying fastest (windows that touch a deletion: layout-major -- the vectors of
 *   jumps over deletions in lexicographic order, "no jump" first, and on one such layout the mixed
 *   radix of the SNPs it meets), forward row then '-' row:  d_kmers [rows][W],
 *   d_start / d_stop (forward: p, e; '-': e, p; e = reference coordinate after the last base, p + W
 *   unless the walk jumps a deletion), d_strand ('+'/'-'), d_freq, d_is_ref (vg's flag: 1 = every SNP
 *   allele is the reference's, also when a deletion is taken), d_region (index into the plan's
 *   regions), d_walk (walk index inside its window).  Enqueue only. */
typedef struct gfm_graph *gfm_graph_t;
int gfm_graph_create(const uint8_t *h_ref, int64_t ref_len, int32_t n_sites, const int32_t *h_pos,
                     const uint8_t *h_n_alts, const uint8_t *h_alt_bases, const int32_t *h_del_len,
                     const int32_t *h_ins_len, const int32_t *h_ins_off, const uint8_t *h_ins_bases,
                     int64_t ins_bytes, const uint64_t *h_alt_bits, int32_t n_haplotypes, gfm_graph_t *out);
void gfm_graph_destroy(gfm_graph_t g);
int gfm_graph_plan(gfm_graph_t g, int32_t n_regions, const int64_t *h_starts, const int64_t *h_stops,
                   int32_t width, int64_t *n_windows, int64_t *n_rows);
/* gfm_graph_plan for RANGES OF WINDOW STARTS: range r holds the windows p in [h_first[r], h_last[r]] of a region that ends at
 * h_limit[r] (a walk is kept if it ends inside its region) -- what a caller uses to cut a region whose rows do not fit one
 * plan (GFM_ERR_OVERFLOW: more than 2^31 rows) into several; d_region of gfm_graph_emit then indexes the ranges. */
int gfm_graph_plan_windows(gfm_graph_t g, int32_t n_ranges, const int64_t *h_first, const int64_t *h_last,
                           const int64_t *h_limit, int32_t width, int64_t *n_windows, int64_t *n_rows);
int gfm_graph_emit(gfm_graph_t g, uint8_t *d_kmers, int64_t *d_start, int64_t *d_stop, uint8_t *d_strand,
                   int64_t *d_freq, uint8_t *d_is_ref, int32_t *d_region, int32_t *d_walk, void *stream);

/* The files scan_graph leaves for GRAFIMO's own compute_results (extract_regions.py:165-170,180,225: one
 * `vg find ... > width_W/CHR_S-E.tsv` per region), written from the rows of gfm_graph_emit by host threads of the
 * library's crew: seven tab-separated columns per row as vg prints them --
 *     REGION  KMER  CHR:START(+|-)  CHR:STOP(+|-)  COUNT  ref|non.ref  NODE(+|-),NODE(+|-),...
 * -- d_* are the buffers gfm_graph_emit filled for this handle (n_rows rows, region-major), `width` the plan's;
 * h_region_stops [n_regions] the regions' E (a walk ends inside its region: the layouts of a window near the end depend on
 * it), h_labels [n_regions] column 1, h_paths [n_regions] the file of every region, chrom_name the name printed in columns 3
 * and 4.  Column 7 (vg's node ids along the walk, '-' rows reversed; the node numbering of `vg construct -m 32` on this graph,
 * pinned by vg's own .vg / .xg files) is left EMPTY with GFM_TSV_NO_NODEPATH: GRAFIMO's scoring never reads it
 * (score_sequences.py:279-293 uses columns 1-6).  h_seen [n_regions] in/out, zeroed by the caller before the first call of a
 * directory: a region whose file this sequence of calls already wrote is appended to (plans that cut one region into several
 * pieces, gfm_graph_plan_windows); regions that never get a row are left to the caller (vg's redirect leaves an empty file).
 * Synchronous: waits for `stream` (the emit), copies the rows to the host in chunks, returns when the files are closed.
 * n_threads <= 0: as many as the machine has, at most 32. */
#define GFM_TSV_NO_NODEPATH 1u
typedef struct gfm_tsv_write_stats {
    int64_t n_rows, n_files, bytes;
    double total_s, copy_s, format_s;   /* wall time; waiting for device->host copies; formatting + writing */
    int32_t threads, reserved;
} gfm_tsv_write_stats_t;
int gfm_graph_write_tsvs(gfm_graph_t g, const uint8_t *d_kmers, const int64_t *d_start, const int64_t *d_stop,
                         const uint8_t *d_strand, const int64_t *d_freq, const uint8_t *d_is_ref, const int32_t *d_region,
                         const int32_t *d_walk, int64_t n_rows, int32_t width, int32_t n_regions,
                         const int64_t *h_region_stops, const char *const *h_labels, const char *const *h_paths,
                         const char *chrom_name, uint32_t flags, int n_threads, uint8_t *h_seen, void *stream,
                         gfm_tsv_write_stats_t *stats);

/* ------------------------------------------------------------------ extraction fused into scoring
 * The join `extract_seqs -> score_seq` of the hot path without its intermediate: the reference writes every row of
 * `vg find -K W -E` to a TSV (extract_regions.py:180,225) and score_seqs reads them all back (score_sequences.py:273-321),
 * although only the rows under the threshold are ever reported (resultsTmp.py:303-310).  gfm_graph_score walks the
 * windows of the regions, scores both strands of every walk against the motif and keeps: the score histogram of ALL rows
 * (q-values are computed over all of them, score_sequences.py:194-198), the number of rows, and one 16-byte entry per row
 * with score >= select_cutoff.  gfm_graph_annotate then produces, for those entries only, the columns a TSV row would
 * have carried -- k-mer, start, stop, strand, haplotype count, vg's ref flag, region -- identical to what
 * gfm_graph_plan + gfm_graph_emit + gfm_score_kmers give for the same rows.
 *
 * gfm_graph_score: regions as gfm_graph_plan takes them; the motif's width is the k-mer width.
 *   flags: GFM_GRAPH_FORWARD_ONLY = skip the '-' rows before they are scored or counted (--no-reverse, :281-282).
 *   d_hist uint64 [L] in/out or NULL (+= rows per score); select_cutoff as gfm_score_kmers (GFM_NO_SELECT: no hits);
 *   d_hits gfm_graph_entry_t [hit_capacity], appended from *d_hit_count on (entries beyond the capacity are counted,
 *   not stored; the caller zeroes the counter); *d_n_rows += rows scored; *d_overflow (optional; the caller zeroes it
 *   with its counters, the kernels write it directly) = 1 if a window holds
 *   more than 2^40 walks, or the regions hold more than 2^20 windows of more than 64 walks each (those windows' rows are
 *   left out: score fewer regions at a time); *n_windows (host, optional) = windows of the call.  Entry order is
 *   arbitrary; the records of gfm_graph_annotate sorted by (w, q2) are in the row order of gfm_graph_emit.  Enqueue only (the first call for a
 *   (regions, width) pair builds and uploads its PLAN -- the tile table; the lists of the windows that touch an insertion /
 *   deletion and of the heavy windows follow on the device -- and later calls with the same regions and width reuse it; a handle
 *   keeps the plans of its 32 most recently used pairs: the widths of a motif set alternate, grafimo.py:177-183).
 * gfm_graph_annotate: for entry i < min(*d_hit_count, hit_capacity) the record d_records[i] of the LAST gfm_graph_score
 *   call on this handle; d_cutoff (device, optional): entries with score < *d_cutoff get keep = 0 and no columns (the
 *   p < t candidates of a --qvalueT scan that the q-value cutoff drops: q >= p); d_qtable (device, optional): the
 *   q-value of the entry's score.  Enqueue only.
 * One call at a time per graph handle: the fused calls of a handle share its scratch (overflow word, histogram slabs, the
 * plan's lists, the tile table), so a gfm_graph_score / gfm_graph_annotate on another stream than the handle's previous
 * call waits (an event, on the device) for that call's work; use one handle per stream for calls that should overlap. */
#define GFM_GRAPH_FORWARD_ONLY 1u
typedef struct gfm_graph_entry {   /* opaque to the caller, who only provides the room: */
    int32_t tile;       /* where the walk is among the call's windows (64 consecutive window starts of one region) */
    int32_t score;      /* scaled score */
    int64_t q2k;        /* window of the tile << 56 | (walk of the window * 2 + strand: 0 '+', 1 '-') */
} gfm_graph_entry_t;
typedef struct gfm_graph_hit {
    int64_t start, stop, freq, q2;
    double qvalue;
    int32_t w, score, region;
    uint8_t strand, is_ref, keep, pad;
    uint8_t kmer[GFM_MAX_WIDTH];
} gfm_graph_hit_t;
int gfm_graph_score(gfm_graph_t g, gfm_motif_t m, int32_t n_regions, const int64_t *h_starts, const int64_t *h_stops,
                    uint32_t flags, int32_t select_cutoff, uint64_t *d_hist, void *d_hits, int64_t hit_capacity,
                    uint64_t *d_hit_count, uint64_t *d_n_rows, int32_t *d_overflow, int64_t *n_windows, void *stream);
/* gfm_graph_score for up to THREE motifs of one width over ONE enumeration of the walks -- the `for motif in motif_set`
 * loop of grafimo.findmotif (grafimo.py:177-183; motifs are processed per width, motif_ops.py:303-311) for the motifs of a
 * width: tiles, site records, window classification and the walks' digits are shared, every motif has its own LDS score table,
 * histogram window, hit list and cutoff.  Arrays of n_motifs entries: select_cutoffs, d_hist (NULL or NULL entries: no
 * histogram for that motif), d_hits / hit_capacity / d_hit_count (one list per motif; the caller zeroes the counters).
 * *d_n_rows += rows scored PER MOTIF (every motif scores the same walks).  Results are identical to n_motifs gfm_graph_score
 * calls.  gfm_graph_annotate serves every motif's list in turn (the entries do not depend on the motif). */
int gfm_graph_score_multi(gfm_graph_t g, const gfm_motif_t *motifs, int32_t n_motifs, int32_t n_regions,
                          const int64_t *h_starts, const int64_t *h_stops, uint32_t flags, const int32_t *select_cutoffs,
                          uint64_t *const *d_hist, void *const *d_hits, const int64_t *hit_capacity,
                          uint64_t *const *d_hit_count, uint64_t *d_n_rows, int32_t *d_overflow, int64_t *n_windows,
                          void *stream);
/* Measurement aid (bench.py `extract.roofline`): with on != 0 the next (up to 64) gfm_graph_score[_multi] calls bracket
 * graph_score_kernel ALONE -- the kernel of the plain and one-deletion windows, the one the fused path's time is in -- with
 * a hipEvent pair on the launch stream; gfm_graph_profile_read waits for them and returns the durations in ms, oldest first. */
int gfm_graph_profile_enable(gfm_graph_t g, int on);
int gfm_graph_profile_read(gfm_graph_t g, float *h_ms_out, int capacity, int *n_out);
int gfm_graph_annotate(gfm_graph_t g, const void *d_hits, const uint64_t *d_hit_count, int64_t hit_capacity,
                       const int32_t *d_cutoff, const double *d_qtable, void *d_records, void *stream);

/* ------------------------------------------------------------------ the report's rows from the hit records (host)
 * replaces what ResultTmp.to_df does with the rows that passed the threshold (resultsTmp.py:303-314): the --recomb filter
 * (:309-310: rows with haplotype_frequency == 0 are dropped unless --recomb), the ascending sort by p-value (:312; rows
 * of equal p-value stay in the order of the TSV rows -- chromosome entry, window, walk, strand -- as a stable sort over
 * the files' rows leaves them) and the columns of resultsTmp.py:270-301 as plain arrays.  No device is touched.
 *   h_ptable [table_len]: the motif's tail table (gfm_motif_tables), p-value of a row = h_ptable[score]
 *     (score_sequences.py:390-391); scale, offset, width: score = scaled / scale + width * offset (:393);
 *   h_recs[p] / n_recs[p]: the records gfm_graph_annotate wrote for one graph handle of the call, copied to the host;
 *   h_entry_of[p] (optional): int64 per region of that handle's region list -> the rank of its chromosome entry in the
 *     caller's order (rows are ordered by it first); region_base[p] (optional): added to a record's `region` in o_region;
 *   flags: GFM_HITS_DROP_ZERO_FREQ (no --recomb), GFM_HITS_FIRST_PER_REGION (only the first row of every region in
 *     report order: what --top-graphs walks, res_writer.py:153-157);
 *   outputs (caller-allocated for sum(n_recs) rows, any may be NULL): start, stop, haplotype_frequency, region,
 *     score = log-odds (score_sequences.py:393), p-value, q-value, strand (0 '+', 1 '-'), reference (1 = "ref": vg's
 *     flag AND |stop - start| == width, score_sequences.py:305-307), o_kmers [rows][width + 1] the k-mer and a '\n'
 *     (one decode + split makes the matched_sequence strings).  *n_out = rows written.
 *   A table of 4 096 rows or more is built with the help of up to four of the library's kept host threads when they are idle. */
#define GFM_HITS_DROP_ZERO_FREQ 1u
#define GFM_HITS_FIRST_PER_REGION 2u
int gfm_graph_hit_columns(const double *h_ptable, int32_t table_len, int32_t scale, double offset, int32_t width,
                          int32_t n_parts, const gfm_graph_hit_t *const *h_recs,
                          const int64_t *n_recs, const int64_t *const *h_entry_of, const int64_t *region_base,
                          uint32_t flags, int64_t *n_out, int64_t *o_start, int64_t *o_stop, int64_t *o_freq,
                          int64_t *o_region, double *o_score, double *o_pvalue, double *o_qvalue, uint8_t *o_strand,
                          uint8_t *o_ref, uint8_t *o_kmers);
/* gfm_graph_hit_columns for the motifs of a SET without holding the caller -- grafimo.findmotif builds one table per motif
 * of the set (grafimo.py:177-183), and what a Python caller does with a table's columns next (strings, a DataFrame) holds
 * the interpreter: the jobs -- one per motif, the arguments of gfm_graph_hit_columns as a struct; `status` and `n_out`
 * are written by the library -- are taken by the library's kept host threads, _start returns at once, _wait returns when
 * every job is done (and frees the run): GFM_OK, or the code of the first job that failed with its message in
 * gfm_last_error(); jobs[i].status says which.  The jobs array and every buffer it points to stay untouched by the caller
 * from _start to _wait.  Every run that was started must be waited for. */
typedef struct gfm_hit_columns_job {
    const double *h_ptable;
    const gfm_graph_hit_t *const *h_recs;
    const int64_t *n_recs;
    const int64_t *const *h_entry_of;
    const int64_t *region_base;
    int64_t *o_start, *o_stop, *o_freq, *o_region;
    double *o_score, *o_pvalue, *o_qvalue;
    uint8_t *o_strand, *o_ref, *o_kmers;
    double offset;
    int64_t n_out;      /* out: rows written */
    int32_t table_len, scale, width, n_parts;
    uint32_t flags;
    int32_t status;     /* out: GFM_OK or this job's error code */
} gfm_hit_columns_job_t;
typedef struct gfm_hit_columns_run *gfm_hit_columns_run_t;
int gfm_graph_hit_columns_start(gfm_hit_columns_job_t *jobs, int32_t n_jobs, gfm_hit_columns_run_t *out);
int gfm_graph_hit_columns_wait(gfm_hit_columns_run_t run);
/* The sequence_name strings of regions, "CHROM:START-STOP" (extract_regions.py:165-170: the query string of `vg find -p`,
 * which score_seqs copies from column 1 of the rows), '\n'-terminated, one after the other in h_out.  Returns the bytes
 * written; with capacity too small (0 to ask) nothing is written and the room to come back with is returned. */
int64_t gfm_region_labels(const char *chrom, const int64_t *h_starts, const int64_t *h_stops, int64_t n, char *h_out,
                          int64_t capacity);

/* Phased VCF (plain or gzip/bgzip) -> the site arrays of gfm_graph_create for one chromosome; host
 * threads parse the lines.  The reference hands the VCF to `vg construct` / `vg index -G`
 * (constructVG.py:332,394); here every ALT allele is taken apart: single-base substitutions (one site per
 * position with up to 3 alternates, also across records), insertions, plain deletions, equal-length
 * multi-base substitutions (one substitution per mismatching position) -- after normalising the ALT against REF
 * (common trailing, then leading bases dropped: the alleles of an STR record REF=ATTT ALT=A,AT,ATT become deletions of
 * 3, 2, 1 bases behind its first base; up to 64 ALT alleles per record).  What is then still none of these -- a
 * complex allele, REF=ACG ALT=TC -- becomes the substitutions of its first min(|REF|, |ALT|) bases plus an insertion /
 * deletion of the rest behind the last of them, with the allele's carriers (vg construct decomposes by alignment: same
 * haplotype sequences).  A record with an ALT that is no string of A, C, G, T (symbolic <DEL> / <CN0>, breakends, '*')
 * is left out whole, as `vg construct` without --handle-sv leaves it out; its ALT alleles are counted in *n_skipped.
 * Deletions may overlap; two records that delete the same bases merge their carriers.  Two haplotypes per sample in
 * file order.
 * gfm_vcf_read copies: pos [n], n_alts [n], alt_bases [n][3], del_len [n], alt_bits [n][3][ceil(H/64)]
 * (may be NULL). */
typedef struct gfm_vcf *gfm_vcf_t;
int gfm_vcf_open(const char *path, const char *chrom, int with_haplotypes, int n_threads, gfm_vcf_t *out,
                 int64_t *n_sites, int32_t *n_haplotypes, int64_t *n_skipped);
int gfm_vcf_read(gfm_vcf_t v, int32_t *pos, uint8_t *n_alts, uint8_t *alt_bases, int32_t *del_len,
                 uint64_t *alt_bits);
/* insertion sites: ins_len [n], ins_off [n] into ins_bases [gfm_vcf_ins_bytes()] (ins_bases may be NULL) */
int64_t gfm_vcf_ins_bytes(gfm_vcf_t v);
int gfm_vcf_read_insertions(gfm_vcf_t v, int32_t *ins_len, int32_t *ins_off, uint8_t *ins_bases);
void gfm_vcf_close(gfm_vcf_t v);

#ifdef __cplusplus
}
#endif
#endif /* GRAFIMO_HIP_H */
