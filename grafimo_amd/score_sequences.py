"""compute_results: the scoring step of ``grafimo findmotif`` on the GPU (seam S3).

Drop-in for the reference's score_sequences.py (paths relative to /root/reference/src/grafimo/):

  compute_results(motif, sequence_loc, debug, args_obj=None, testmode=False) -> DataFrame   :44-211
  compute_qvalues(pvalues, debug)                                                           :401-428
  print_scoring_msg(motif, noreverse, debug)                                                :433-464

Same arguments, same printed lines, same DataFrame schema.  What changes is the inside: the
reference forks ``cores`` Python workers that parse the vg TSVs line by line and score every
k-mer with two O(1000*W) sums (:216-328, :331-396); here the TSVs are parsed by the C++ mmap
ingest (``cores`` host threads), the k-mer matrix is scored by the HIP kernels, BH q-values come
from the device score histogram, the threshold is applied on the device and only hit rows come
back.  There is no CPU scoring fallback: without libgrafimo_hip.so or a GPU the call fails.
"""
import ctypes
import glob
import os
import time
from typing import List, Optional

import numpy as np
import pandas as pd

from . import _native as nv
from .device import DeviceMotif
from .motif import Motif, is_motif_like
from .resultsTmp import build_frame
from .utils import die, exception_handler, print_progress_bar
from .workflow import is_findmotif_like


class KmerTable:
    """Columnar view of the vg TSV rows of one motif width (gfm_tsv_* ingest)."""

    def __init__(self, paths: List[str], width: int, skip_reverse: bool, threads: int):
        arr = (ctypes.c_char_p * len(paths))(*[p.encode() for p in paths])
        h = ctypes.c_void_p()
        n = ctypes.c_int64()
        nv.check(nv.lib().gfm_tsv_open(arr, len(paths), int(width), int(bool(skip_reverse)),
                                       int(threads), ctypes.byref(h), ctypes.byref(n)))
        try:
            self.n = int(n.value)
            self.width = int(width)
            self.kmers = np.empty((self.n, width), dtype=np.uint8)
            self.start = np.empty(self.n, dtype=np.int64)
            self.stop = np.empty(self.n, dtype=np.int64)
            self.strand = np.empty(self.n, dtype=np.uint8)
            self.freq = np.empty(self.n, dtype=np.int64)
            self.is_ref = np.empty(self.n, dtype=np.uint8)
            self.name_id = np.empty(self.n, dtype=np.int32)
            nv.check(nv.lib().gfm_tsv_read(h, nv.ptr(self.kmers), nv.ptr(self.start), nv.ptr(self.stop),
                                           nv.ptr(self.strand), nv.ptr(self.freq), nv.ptr(self.is_ref),
                                           None, nv.ptr(self.name_id)))
            k = nv.lib().gfm_tsv_name_count(h)
            nbytes = int(nv.lib().gfm_tsv_names_bytes(h))
            off = np.empty(k + 1, dtype=np.int64)
            buf = np.empty(max(nbytes, 1), dtype=np.uint8)
            nv.check(nv.lib().gfm_tsv_names(h, nv.ptr(off), nv.ptr(buf)))
            raw = buf.tobytes()
            self.names = [raw[off[i]:off[i + 1]].decode() for i in range(k)]
        finally:
            nv.lib().gfm_tsv_close(h)


class ScanHits:
    """The hit rows of one motif of a streamed scan, ascending by row id, each with the columns of its TSV row."""

    def __init__(self, h, j: int, K: int, width: int, want_qvalues: bool):
        self.n_hits = K
        self.rows = np.empty(K, dtype=np.int64)
        self.scaled = np.empty(K, dtype=np.int32)
        self.logodds = np.empty(K, dtype=np.float64)
        self.pvalue = np.empty(K, dtype=np.float64)
        self.qvalue = np.empty(K, dtype=np.float64) if want_qvalues else None
        self.kmers = np.empty((K, width), dtype=np.uint8)
        self.start = np.empty(K, dtype=np.int64)
        self.stop = np.empty(K, dtype=np.int64)
        self.strand = np.empty(K, dtype=np.uint8)
        self.freq = np.empty(K, dtype=np.int64)
        self.is_ref = np.empty(K, dtype=np.uint8)
        self.name_id = np.empty(K, dtype=np.int32)
        nv.check(nv.lib().gfm_scan_hits_of(h, j, nv.ptr(self.rows), nv.ptr(self.scaled), nv.ptr(self.logodds),
                                           nv.ptr(self.pvalue), nv.ptr(self.qvalue), nv.ptr(self.kmers),
                                           nv.ptr(self.start), nv.ptr(self.stop), nv.ptr(self.strand),
                                           nv.ptr(self.freq), nv.ptr(self.is_ref), nv.ptr(self.name_id)))


class ScanRetry(nv.NativeError):
    """a deferred scan's hit list was too short; the library has grown it: run both phases again (on every rank)"""


class StreamScan:
    """gfm_scan_tsv_begin / _finish: the TSV files of one width parsed, uploaded and scored as one pipelined pass
    (parse threads -> pinned chunks -> copy stream -> score kernel per chunk); only the hit rows -- with the
    columns of their TSV rows -- come back.  `dm`: one DeviceMotif, or several of ONE width (they share the pass:
    every chunk is read once per group of up to three motifs).  .n rows were scored; .stats holds the timing split;
    .hits[j] are motif j's rows (one motif: also as attributes of the scan itself).

    `hists`: torch int64 tensors [L] (one per motif, on the motifs' device) that receive the score histograms
    instead of the library's own buffers, and `defer=True` stops after the scoring phase: a sharded caller
    all-reduces the tensors over its process group, synchronises, then calls finish() (distributed.py)."""

    def __init__(self, dm, paths: List[str], skip_reverse: bool, threads: int, threshold: float,
                 on_qvalue: bool, want_qvalues: bool, chunk_rows: int = 0, hists=None, defer: bool = False):
        self.dms = list(dm) if isinstance(dm, (list, tuple)) else [dm]
        self.width = self.dms[0].width
        self.want_qvalues = bool(want_qvalues)
        if hists is not None and len(hists) != len(self.dms):
            raise ValueError("one histogram tensor per motif")
        self._args = (list(paths), int(bool(skip_reverse)), int(threads), float(threshold), int(bool(on_qvalue)),
                      int(bool(want_qvalues)), int(chunk_rows))
        self._hists = hists             # kept alive until finish()
        self._defer = bool(defer)
        self._h = None
        self.hits = None
        self._begin()
        if not defer:
            self.finish()

    def _begin(self):
        paths, skip_reverse, threads, threshold, on_qvalue, want_qvalues, chunk_rows = self._args
        M = len(self.dms)
        arr, _keep = nv.c_paths(paths)
        handles = (ctypes.c_void_p * M)(*[d.handle for d in self.dms])
        hist_ptrs = None
        if self._hists is not None:
            hist_ptrs = (ctypes.c_void_p * M)(*[t.data_ptr() for t in self._hists])
        h = ctypes.c_void_p()
        n = ctypes.c_int64()
        nv.check(nv.lib().gfm_scan_tsv_begin(handles, M, arr, len(paths), skip_reverse, threads, threshold, on_qvalue,
                                             want_qvalues, chunk_rows, hist_ptrs, ctypes.byref(h), ctypes.byref(n)))
        self._h = h
        self.n = int(n.value)

    def finish(self):
        """Second phase: q-tables and cutoffs from the histograms as they are now, selection, hits back.
        The scan stores no scores: a hit list that turns out too short (more than one row in sixteen passes the threshold)
        is grown by the library, which then asks for the scan again (GFM_ERR_OVERFLOW) -- done here, once, for a scan that
        runs both phases itself; a deferred scan (a sharded caller all-reduces the histograms between the phases) gets
        ScanRetry and repeats both phases on every rank together."""
        if self._h is None:
            raise RuntimeError("the scan is closed")
        M = len(self.dms)
        try:
            counts = (ctypes.c_int64 * M)()
            rc = nv.lib().gfm_scan_tsv_finish(self._h, counts)
            if rc == nv.GFM_ERR_OVERFLOW:
                msg = nv.lib().gfm_last_error().decode("utf-8", "replace")
                if self._defer:
                    raise ScanRetry(rc, msg)
                self.close_handle()
                self._begin()
                rc = nv.lib().gfm_scan_tsv_finish(self._h, counts)
            nv.check(rc)
            self.hits = [ScanHits(self._h, j, int(counts[j]), self.width, self.want_qvalues) for j in range(M)]
            self.stats = nv.ScanStats()
            nv.check(nv.lib().gfm_scan_stats(self._h, ctypes.byref(self.stats)))
            t = nv.lib().gfm_scan_table(self._h)
            cnt = nv.lib().gfm_tsv_name_count(t)
            nbytes = int(nv.lib().gfm_tsv_names_bytes(t))
            off = np.empty(cnt + 1, dtype=np.int64)
            buf = np.empty(max(nbytes, 1), dtype=np.uint8)
            nv.check(nv.lib().gfm_tsv_names(t, nv.ptr(off), nv.ptr(buf)))
            raw = buf.tobytes()
            if raw.isascii():               # (byte offsets are character offsets: one decode, then slices of the str)
                txt, o = raw.decode("ascii"), off.tolist()
                self.names = [txt[o[i]:o[i + 1]] for i in range(cnt)]
            else:
                self.names = [raw[off[i]:off[i + 1]].decode() for i in range(cnt)]
        finally:
            self.close()
        first = self.hits[0]                # one motif: the hit columns as attributes of the scan, as before
        for k, v in vars(first).items():
            setattr(self, k, v)
        return self

    def close_handle(self):
        if getattr(self, "_h", None) is not None:
            nv.lib().gfm_scan_close(self._h)
            self._h = None

    def close(self):
        if getattr(self, "_h", None) is not None:
            nv.lib().gfm_scan_close(self._h)
            self._h = None
            self._hists = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def print_scoring_msg(motif: Motif, noreverse: bool, debug: bool) -> None:
    """'Scoring hits for motif +ID.' / '-ID.' (score_sequences.py:433-464)."""
    if not is_motif_like(motif):
        exception_handler(TypeError, f"Expected Motif, got {type(motif).__name__}.\n", debug)
    if not isinstance(noreverse, bool):
        exception_handler(TypeError, f"Expected bool, got {type(noreverse).__name__}.\n", debug)
    msg = "Scoring hits for motif {}."
    print(msg.format("+" + motif.motif_id))
    if not noreverse:
        print(msg.format("-" + motif.motif_id), end="\n\n")


def compute_qvalues(pvalues: List[np.double], debug: bool) -> List[np.double]:
    """Benjamini-Hochberg q-values of a list of p-values (score_sequences.py:401-428).
    Host-side equivalent kept for API compatibility (sort, p/(rank/n), reverse cumulative
    minimum, clip at 1); compute_results itself derives q-values from the device histogram."""
    if not isinstance(pvalues, list):
        exception_handler(TypeError, f"Expected list, got {type(pvalues).__name__}.\n", debug)
    print("\nComputing q-values...\n")
    p = np.asarray(pvalues, dtype=np.float64)
    n = len(p)
    order = np.argsort(p, kind="stable")
    raw = p[order] / (np.arange(1, n + 1) / float(n))
    corr = np.minimum.accumulate(raw[::-1])[::-1]
    corr[corr > 1] = 1
    out = np.empty(n, dtype=np.float64)
    out[order] = corr
    return list(out)


def compute_results(motif: Motif, sequence_loc: str, debug: bool, args_obj=None,
                    testmode: Optional[bool] = False) -> pd.DataFrame:
    """Score the k-mers extracted from the genome variation graph and build the report table.

    ``args_obj`` is the reference's ``Findmotif`` (or grafimo_amd.workflow.Findmotif): .cores,
    .threshold, .noqvalue, .qvalueT, .noreverse, .recomb, .verbose are read.  ``testmode``
    hard-codes the reference's test settings (score_sequences.py:100-107)."""
    if not is_motif_like(motif):
        exception_handler(TypeError, f"Expected Motif, got {type(motif).__name__}.\n", debug)
    if not isinstance(sequence_loc, str):
        exception_handler(TypeError, f"Expected str, got {type(sequence_loc).__name__}.\n", debug)
    if not os.path.isdir(sequence_loc):
        exception_handler(FileNotFoundError, f"Unable to locate {sequence_loc}.\n", debug)
    if not testmode:
        if not is_findmotif_like(args_obj):
            exception_handler(TypeError, f"Expected Findmotif, got {type(args_obj).__name__}.\n", debug)
        cores, threshold = args_obj.cores, args_obj.threshold
        no_qvalue, qval_t = args_obj.noqvalue, args_obj.qvalueT
        no_reverse, recomb, verbose = args_obj.noreverse, args_obj.recomb, args_obj.verbose
    else:
        cores, threshold, recomb = 1, float(1), True
        no_qvalue = qval_t = no_reverse = verbose = False
    assert threshold > 0 and threshold <= 1
    assert cores >= 1
    # What grafimo_amd's scan_graph leaves when THIS function is its consumer (extract_regions.scan_graph): a manifest of
    # graphs and regions instead of rows -- scored where they are enumerated, the same table as from the files.
    from .extract_regions import compute_results_from_manifest, read_manifest
    manifest = read_manifest(sequence_loc)
    if manifest is not None:
        if testmode:
            from .workflow import Findmotif
            args_obj = Findmotif(cores=1, threshold=1.0, recomb=True)
        print_progress_bar(0, 1, prefix="Progress:", suffix="Complete", length=50)
        try:
            df = compute_results_from_manifest(motif, manifest, debug, args_obj)
        except nv.NativeError as e:
            exception_handler(RuntimeError, e.msg + "\n", debug)
        except KeyboardInterrupt:
            print("\nCaught SIGINT. GRAFIMO will exit")
            die(2)
        print_progress_bar(1, 1, prefix="Progress:", suffix="Complete", length=50)
        return df
    print_scoring_msg(motif, no_reverse, debug)

    width = motif.width
    files = sorted(glob.glob(os.path.join(sequence_loc, f"width_{width}", "*.tsv")))
    start_s = time.time()
    print_progress_bar(0, 1, prefix="Progress:", suffix="Complete", length=50)
    if not files:      # nothing was extracted for this width (score_sequences.py:189-192)
        errmsg = "No result retrieved. Unable to proceed.\n"
        errmsg += "\nAre you using the correct VGs and searching on the right chromosomes?\n"
        exception_handler(ValueError, errmsg, debug)
    dm = DeviceMotif.lease(motif)            # a kept handle when this motif was scored before (device.py)
    try:
        scan = StreamScan(dm, files, no_reverse, cores, float(threshold), bool(qval_t), not no_qvalue)
    except nv.NativeError as e:
        exception_handler(ValueError if e.code == nv.GFM_ERR_IO else RuntimeError, e.msg + "\n", debug)
    except KeyboardInterrupt:
        print("\nCaught SIGINT. GRAFIMO will exit")
        die(2)
    finally:
        dm.release()
    if scan.n == 0:
        errmsg = "No result retrieved. Unable to proceed.\n"
        errmsg += "\nAre you using the correct VGs and searching on the right chromosomes?\n"
        exception_handler(ValueError, errmsg, debug)
    if not no_qvalue:
        print("\nComputing q-values...\n")
    print_progress_bar(1, 1, prefix="Progress:", suffix="Complete", length=50)
    if verbose:
        st = scan.stats
        print("Sequences scored in %.2fs (parsing %.2fs on %d threads, %d chunk(s), %.1f MB to the GPU)"
              % (time.time() - start_s, st.parse_s, st.parse_threads, st.n_chunks, st.h2d_bytes / 1e6))
    print(f"Scanned sequences:\t{scan.n}")
    print(f"Scanned nucleotides:\t{scan.n * width}")

    start_df = time.time()
    df = _frame_from_scan(motif, scan.hits[0], np.array(scan.names, dtype=object), no_qvalue, recomb)
    if verbose:
        print("\nResults summary built in %.2fs" % (time.time() - start_df))
    return df


def _frame_from_scan(motif: Motif, hits: "ScanHits", names, no_qvalue: bool, recomb: bool) -> pd.DataFrame:
    """The report table of one motif from the hit rows of a streamed scan (names: the scan's REGION strings)."""
    return build_frame(
        motif,
        seqnames=list(names[hits.name_id]) if hits.n_hits else [],
        starts=hits.start, stops=hits.stop,
        strands=[chr(c) for c in hits.strand],
        scores=hits.logodds, pvalues=hits.pvalue,
        qvalues=None if no_qvalue else hits.qvalue,
        seqs=[bytes(k).decode() for k in hits.kmers],
        frequencies=hits.freq,
        references=["ref" if r else "non.ref" for r in hits.is_ref],
        threshold=None, recomb=bool(recomb),
    )


def compute_results_many(motifs: List[Motif], sequence_loc: str, debug: bool, args_obj) -> List[pd.DataFrame]:
    """compute_results for a whole motif set (the `for motif in motif_set` loop of grafimo.findmotif,
    grafimo.py:177-183) without repeating the shared work: the TSV files of a width go through ONE streamed pass for
    all motifs of that width (gfm_scan_tsv_begin / _finish: parsed and uploaded once, every chunk scored by
    gfm_score_kmers_multi -- up to three motifs share each read of the k-mers on the device).  Returns the tables
    in the order of `motifs`; every table equals compute_results(motif, ...) and the same lines are printed per
    motif."""
    if not is_findmotif_like(args_obj):
        exception_handler(TypeError, f"Expected Findmotif, got {type(args_obj).__name__}.\n", debug)
    threshold = float(args_obj.threshold)
    no_qvalue, qval_t = bool(args_obj.noqvalue), bool(args_obj.qvalueT)
    no_reverse, recomb = bool(args_obj.noreverse), bool(args_obj.recomb)
    assert 0 < threshold <= 1
    out: List[Optional[pd.DataFrame]] = [None] * len(motifs)
    by_width = {}
    for i, m in enumerate(motifs):
        if not is_motif_like(m):
            exception_handler(TypeError, f"Expected Motif, got {type(m).__name__}.\n", debug)
        by_width.setdefault(m.width, []).append(i)
    # scan_graph's manifest instead of rows (extract_regions.scan_graph): the set goes through the graph, the motifs of a
    # width sharing the enumeration of the walks
    from .extract_regions import compute_results_many_from_manifest, read_manifest
    manifest = read_manifest(sequence_loc)
    if manifest is not None:
        try:
            return compute_results_many_from_manifest(motifs, manifest, debug, args_obj)
        except nv.NativeError as e:
            exception_handler(RuntimeError, e.msg + "\n", debug)
    for width, idxs in by_width.items():
        files = sorted(glob.glob(os.path.join(sequence_loc, f"width_{width}", "*.tsv")))
        no_rows = "No result retrieved. Unable to proceed.\n"
        no_rows += "\nAre you using the correct VGs and searching on the right chromosomes?\n"
        if not files:
            exception_handler(ValueError, no_rows, debug)
        dms = [DeviceMotif.from_motif(motifs[i]) for i in idxs]
        try:
            try:
                scan = StreamScan(dms, files, no_reverse, max(1, int(args_obj.cores)), threshold, qval_t, not no_qvalue)
            except nv.NativeError as e:
                exception_handler(ValueError if e.code == nv.GFM_ERR_IO else RuntimeError, e.msg + "\n", debug)
            if scan.n == 0:
                exception_handler(ValueError, no_rows, debug)
            names = np.array(scan.names, dtype=object)
            for i, hits in zip(idxs, scan.hits):
                print_scoring_msg(motifs[i], no_reverse, debug)
                if not no_qvalue:
                    print("\nComputing q-values...\n")
                print(f"Scanned sequences:\t{scan.n}")
                print(f"Scanned nucleotides:\t{scan.n * width}")
                out[i] = _frame_from_scan(motifs[i], hits, names, no_qvalue, recomb)
        finally:
            for dm in dms:
                dm.close()
    return out
