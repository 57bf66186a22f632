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


class StreamScan:
    """gfm_scan_tsv: the TSV files of one width parsed, uploaded and scored as one pipelined pass (parse
    threads -> pinned chunks -> copy stream -> score kernel per chunk); only the hit rows -- with the
    columns of their TSV rows -- come back.  .n rows were scored; .stats holds the timing split."""

    def __init__(self, dm: DeviceMotif, paths: List[str], skip_reverse: bool, threads: int, threshold: float,
                 on_qvalue: bool, want_qvalues: bool, chunk_rows: int = 0):
        arr = (ctypes.c_char_p * len(paths))(*[p.encode() for p in paths])
        h = ctypes.c_void_p()
        n, k = ctypes.c_int64(), ctypes.c_int64()
        nv.check(nv.lib().gfm_scan_tsv(dm.handle, arr, len(paths), int(bool(skip_reverse)), int(threads),
                                       float(threshold), int(bool(on_qvalue)), int(bool(want_qvalues)),
                                       int(chunk_rows), ctypes.byref(h), ctypes.byref(n), ctypes.byref(k)))
        try:
            self.n, self.n_hits, self.width = int(n.value), int(k.value), dm.width
            K = self.n_hits
            self.rows = np.empty(K, dtype=np.int64)
            self.scaled = np.empty(K, dtype=np.int32)
            self.logodds = np.empty(K, dtype=np.float64)
            self.pvalue = np.empty(K, dtype=np.float64)
            self.qvalue = np.empty(K, dtype=np.float64) if want_qvalues else None
            self.kmers = np.empty((K, dm.width), dtype=np.uint8)
            self.start = np.empty(K, dtype=np.int64)
            self.stop = np.empty(K, dtype=np.int64)
            self.strand = np.empty(K, dtype=np.uint8)
            self.freq = np.empty(K, dtype=np.int64)
            self.is_ref = np.empty(K, dtype=np.uint8)
            self.name_id = np.empty(K, dtype=np.int32)
            nv.check(nv.lib().gfm_scan_hits(h, nv.ptr(self.rows), nv.ptr(self.scaled), nv.ptr(self.logodds),
                                            nv.ptr(self.pvalue), nv.ptr(self.qvalue), nv.ptr(self.kmers),
                                            nv.ptr(self.start), nv.ptr(self.stop), nv.ptr(self.strand),
                                            nv.ptr(self.freq), nv.ptr(self.is_ref), nv.ptr(self.name_id)))
            self.stats = nv.ScanStats()
            nv.check(nv.lib().gfm_scan_stats(h, ctypes.byref(self.stats)))
            t = nv.lib().gfm_scan_table(h)
            cnt = nv.lib().gfm_tsv_name_count(t)
            nbytes = int(nv.lib().gfm_tsv_names_bytes(t))
            off = np.empty(cnt + 1, dtype=np.int64)
            buf = np.empty(max(nbytes, 1), dtype=np.uint8)
            nv.check(nv.lib().gfm_tsv_names(t, nv.ptr(off), nv.ptr(buf)))
            raw = buf.tobytes()
            self.names = [raw[off[i]:off[i + 1]].decode() for i in range(cnt)]
        finally:
            nv.lib().gfm_scan_close(h)


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
    print_scoring_msg(motif, no_reverse, debug)

    width = motif.width
    files = sorted(glob.glob(os.path.join(sequence_loc, f"width_{width}", "*.tsv")))
    start_s = time.time()
    print_progress_bar(0, 1, prefix="Progress:", suffix="Complete", length=50)
    if not files:      # nothing was extracted for this width (score_sequences.py:189-192)
        errmsg = "No result retrieved. Unable to proceed.\n"
        errmsg += "\nAre you using the correct VGs and searching on the right chromosomes?\n"
        exception_handler(ValueError, errmsg, debug)
    dm = DeviceMotif.from_motif(motif)
    try:
        scan = StreamScan(dm, files, no_reverse, cores, float(threshold), bool(qval_t), not no_qvalue)
    except nv.NativeError as e:
        exception_handler(ValueError if e.code == nv.GFM_ERR_IO else RuntimeError, e.msg + "\n", debug)
    except KeyboardInterrupt:
        print("\nCaught SIGINT. GRAFIMO will exit")
        die(2)
    finally:
        dm.close()
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
    names = np.array(scan.names, dtype=object)
    df = build_frame(
        motif,
        seqnames=list(names[scan.name_id]) if scan.n_hits else [],
        starts=scan.start,
        stops=scan.stop,
        strands=[chr(c) for c in scan.strand],
        scores=scan.logodds,
        pvalues=scan.pvalue,
        qvalues=None if no_qvalue else scan.qvalue,
        seqs=[bytes(k).decode() for k in scan.kmers],
        frequencies=scan.freq,
        references=["ref" if r else "non.ref" for r in scan.is_ref],
        threshold=None, recomb=bool(recomb),
    )
    if verbose:
        print("\nResults summary built in %.2fs" % (time.time() - start_df))
    return df


def _frame_from_hits(motif: Motif, table: "KmerTable", rows, logodds, pvalue, qvalue, recomb: bool) -> pd.DataFrame:
    names = np.array(table.names, dtype=object)
    return build_frame(
        motif,
        seqnames=list(names[table.name_id[rows]]),
        starts=table.start[rows], stops=table.stop[rows],
        strands=[chr(c) for c in table.strand[rows]],
        scores=logodds, pvalues=pvalue, qvalues=qvalue,
        seqs=[bytes(k).decode() for k in table.kmers[rows]],
        frequencies=table.freq[rows],
        references=["ref" if r else "non.ref" for r in table.is_ref[rows]],
        threshold=None, recomb=bool(recomb),
    )


def compute_results_many(motifs: List[Motif], sequence_loc: str, debug: bool, args_obj) -> List[pd.DataFrame]:
    """compute_results for a whole motif set (the `for motif in motif_set` loop of grafimo.findmotif,
    grafimo.py:177-183) without repeating the shared work: the TSV files of a width are parsed and
    uploaded ONCE for all motifs of that width, and up to three motifs share each read of the k-mers
    on the device (gfm_score_kmers_multi).  Returns the tables in the order of `motifs`; every table
    equals compute_results(motif, ...) and the same lines are printed per motif."""
    import torch
    from .scan import scan_same_width
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
    for width, idxs in by_width.items():
        files = sorted(glob.glob(os.path.join(sequence_loc, f"width_{width}", "*.tsv")))
        try:
            table = KmerTable(files, width, no_reverse, max(1, int(args_obj.cores)))
        except nv.NativeError as e:
            exception_handler(ValueError if e.code == nv.GFM_ERR_IO else RuntimeError, e.msg + "\n", debug)
        if table.n == 0:
            errmsg = "No result retrieved. Unable to proceed.\n"
            errmsg += "\nAre you using the correct VGs and searching on the right chromosomes?\n"
            exception_handler(ValueError, errmsg, debug)
        d_kmers = torch.from_numpy(table.kmers).cuda()
        dms = [DeviceMotif.from_motif(motifs[i]) for i in idxs]
        try:
            results = scan_same_width(dms, d_kmers, threshold, on_qvalue=qval_t, want_qvalues=not no_qvalue)
            for i, dm, res in zip(idxs, dms, results):
                print_scoring_msg(motifs[i], no_reverse, debug)
                if not no_qvalue:
                    print("\nComputing q-values...\n")
                lo, pv = dm.annotate(res["scaled"])
                print(f"Scanned sequences:\t{table.n}")
                print(f"Scanned nucleotides:\t{table.n * width}")
                out[i] = _frame_from_hits(motifs[i], table, res["rows"], lo, pv,
                                          None if no_qvalue else res["qtable"][res["scaled"]], recomb)
        finally:
            for dm in dms:
                dm.close()
    return out
