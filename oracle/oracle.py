"""ctypes binding of the CPU oracle + the reference's row handling restated.

TEST INFRASTRUCTURE ONLY (see grafimo_oracle.c).  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from
grafimo_amd/.  Parity status: pinned (tests/test_oracle.py).

Citations are relative to /root/reference/src/grafimo/.
"""
import ctypes
import glob
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgrafimo_oracle.so")

RANGE = 1000  # utils.py:26
PSEUDOBG = np.double(0.0000005)  # utils.py:24

_c_dp = ctypes.POINTER(ctypes.c_double)
_c_i64p = ctypes.POINTER(ctypes.c_int64)
_c_i32p = ctypes.POINTER(ctypes.c_int32)
_c_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    src = os.path.join(_HERE, "grafimo_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.orc_compute_log_odds.argtypes = [_c_dp, ctypes.c_int, _c_dp, _c_dp]
        L.orc_scale_pwm.argtypes = [_c_dp, ctypes.c_int, _c_i64p,
                                    ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                    ctypes.POINTER(ctypes.c_int), _c_dp]
        L.orc_comp_pval_mat.argtypes = [_c_i64p, ctypes.c_int, _c_dp, _c_dp]
        L.orc_np_sum.argtypes = [_c_dp, ctypes.c_long]
        L.orc_np_sum.restype = ctypes.c_double
        L.orc_seq_sum.argtypes = [_c_dp, ctypes.c_long]
        L.orc_seq_sum.restype = ctypes.c_double
        L.orc_score_kmers.argtypes = [_c_u8p, ctypes.c_long, _c_i64p, _c_dp, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                      ctypes.c_int, _c_i32p, _c_dp, _c_dp]
        L.orc_score_kmers_table.argtypes = [_c_u8p, ctypes.c_long, _c_i64p, _c_dp,
                                            ctypes.c_int, ctypes.c_int, _c_i32p, _c_dp]
        L.orc_fdr_bh.argtypes = [_c_dp, ctypes.c_long, _c_dp]
        L.orc_score_tsv_text.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_int, _c_i64p, _c_dp, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, _c_dp]
        L.orc_score_tsv_text.restype = ctypes.c_long
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


# ---------------------------------------------------------------- motif side
def compute_log_odds(probs, bg):
    """motif_processing.pyx:444-507.  probs f64[4,W] rows A,C,G,T; bg f64[4]."""
    probs = np.ascontiguousarray(probs, dtype=np.float64)
    bg = np.ascontiguousarray(bg, dtype=np.float64)
    W = probs.shape[1]
    out = np.empty((4, W), dtype=np.float64)
    rc = lib().orc_compute_log_odds(_p(probs, _c_dp), W, _p(bg, _c_dp), _p(out, _c_dp))
    if rc:
        raise AssertionError("reference assertion would fire in compute_log_odds")
    return out


def scale_pwm(logodds):
    """motif_ops.py:1090-1111 -> (sm int64[4,W], min_val, max_val, scale:int, offset:f64)."""
    lo = np.ascontiguousarray(logodds, dtype=np.float64)
    W = lo.shape[1]
    sm = np.empty((4, W), dtype=np.int64)
    mn, mx, sc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    off = ctypes.c_double()
    lib().orc_scale_pwm(_p(lo, _c_dp), W, _p(sm, _c_i64p), ctypes.byref(mn),
                        ctypes.byref(mx), ctypes.byref(sc), ctypes.byref(off))
    return sm, mn.value, mx.value, sc.value, np.double(off.value)


def comp_pval_mat(sm, bg):
    """motif_processing.pyx:552-603 -> pmf f64[1000*W+1] (un-normalised last DP row)."""
    sm = np.ascontiguousarray(sm, dtype=np.int64)
    bg = np.ascontiguousarray(bg, dtype=np.float64)
    W = sm.shape[1]
    out = np.empty(RANGE * W + 1, dtype=np.float64)
    rc = lib().orc_comp_pval_mat(_p(sm, _c_i64p), W, _p(bg, _c_dp), _p(out, _c_dp))
    if rc:
        raise RuntimeError(f"orc_comp_pval_mat failed ({rc})")
    return out


def p_table(pmf):
    """O(1) replacement for the per-row tail sum: suffix_sum(pmf)/total.
    (sequential from the top; used by the 'table' CPU variant.)"""
    suf = np.cumsum(pmf[::-1])[::-1]
    return suf / suf[0]


def pseudo_bg(bg_items, no_reverse):
    """motif_ops.py:1189-1302: average with reverse complement unless
    --no-reverse (average_bg_with_rc :1233-1263), then norm_bg (:1268-1302):
    (bg + 5e-7) / (sum(bg) + 4*5e-7).

    bg_items: [(nuc, prob), ...] in the order the reference's dict holds them
    (alphabet order for the uniform background, file order for a bg file) --
    the f64 total is accumulated in dict order, so order is part of the input.
    Returns f64[4] in A,C,G,T order."""
    rev = {"A": "T", "C": "G", "G": "C", "T": "A"}
    bgs = {k: np.double(v) for k, v in bg_items}
    if not no_reverse:
        avg = {}
        for nuc in bgs:
            rc = rev[nuc]
            if ord(nuc) < ord(rc):
                a = np.double((bgs[nuc] + bgs[rc]) / np.double(2))
                avg[nuc] = a
                avg[rc] = a
        bgs = avg
    tot = np.double(len(bgs) * PSEUDOBG)
    for nuc in bgs:
        tot += np.double(bgs[nuc])
    d = {k: np.double((v + PSEUDOBG) / tot) for k, v in bgs.items()}
    return np.array([d[k] for k in "ACGT"], dtype=np.float64)


# ---------------------------------------------------------------- scoring side
def score_kmers(kmers, sm, pmf, min_val, scale, offset, sum_mode=0):
    """score_sequences.py:331-396 over a dense uint8 [N,W] matrix.
    -> (scaled int32[N], logodds f64[N], pvalue f64[N])."""
    kmers = np.ascontiguousarray(kmers, dtype=np.uint8)
    N, W = kmers.shape
    sm = np.ascontiguousarray(sm, dtype=np.int64)
    pmf = np.ascontiguousarray(pmf, dtype=np.float64)
    sc = np.empty(N, dtype=np.int32)
    lo = np.empty(N, dtype=np.float64)
    pv = np.empty(N, dtype=np.float64)
    rc = lib().orc_score_kmers(_p(kmers, _c_u8p), N, _p(sm, _c_i64p), _p(pmf, _c_dp),
                               int(min_val), int(scale), W, float(offset), sum_mode,
                               _p(sc, _c_i32p), _p(lo, _c_dp), _p(pv, _c_dp))
    if rc:
        raise ValueError("k-mer holds a byte outside ACGTacgtN (undefined in the reference)")
    return sc, lo, pv


def score_kmers_table(kmers, sm, ptab, min_val):
    kmers = np.ascontiguousarray(kmers, dtype=np.uint8)
    N, W = kmers.shape
    sm = np.ascontiguousarray(sm, dtype=np.int64)
    ptab = np.ascontiguousarray(ptab, dtype=np.float64)
    sc = np.empty(N, dtype=np.int32)
    pv = np.empty(N, dtype=np.float64)
    rc = lib().orc_score_kmers_table(_p(kmers, _c_u8p), N, _p(sm, _c_i64p), _p(ptab, _c_dp),
                                     int(min_val), W, _p(sc, _c_i32p), _p(pv, _c_dp))
    if rc:
        raise ValueError("k-mer holds a byte outside ACGTacgtN")
    return sc, pv


def score_tsv_text(text: bytes, W, sm, tab, min_val, scale, offset, table=False, no_reverse=False):
    """score_seqs' loop over TSV text (score_sequences.py:273-321): parse every line, score its k-mer
    with the per-row O(L) sums (`tab` = pmf) or one lookup (`table`, `tab` = p_table).
    -> (rows scored, checksum)."""
    sm = np.ascontiguousarray(sm, dtype=np.int64)
    tab = np.ascontiguousarray(tab, dtype=np.float64)
    chk = ctypes.c_double(0.0)
    n = lib().orc_score_tsv_text(text, len(text), int(W), _p(sm, _c_i64p), _p(tab, _c_dp), int(min_val), int(scale),
                                 float(offset), int(bool(table)), int(bool(no_reverse)), ctypes.byref(chk))
    if n < 0:
        raise ValueError("malformed TSV row")
    return int(n), chk.value


def fdr_bh(pvalues):
    """score_sequences.py:401-428 (statsmodels fdr_bh)."""
    p = np.ascontiguousarray(pvalues, dtype=np.float64)
    q = np.empty_like(p)
    if len(p):
        lib().orc_fdr_bh(_p(p, _c_dp), len(p), _p(q, _c_dp))
    return q


def parse_tsv_rows(paths, no_reverse=False):
    """Row handling of score_seqs (score_sequences.py:273-321), pure Python.
    Returns a dict of parallel lists + the uint8 k-mer matrix."""
    cols = {k: [] for k in ("seqname", "seq", "chrom", "start", "stop", "strand", "freq", "ref")}
    for path in paths:
        with open(path) as fh:
            for line in fh:
                data = line.strip().split()
                strand = data[2][-1]
                if no_reverse and strand == "-":
                    continue
                cols["seqname"].append(data[0])
                cols["seq"].append(data[1])
                cols["chrom"].append(data[0].split(":")[0])
                cols["start"].append(int(data[2].split(":")[1][:-1]))
                cols["stop"].append(int(data[3].split(":")[1][:-1]))
                cols["strand"].append(strand)
                cols["freq"].append(int(data[4]))
                cols["ref"].append(data[5])
    return cols


def compute_results(motif, sequence_loc, threshold=1.0, no_qvalue=False, qval_t=False,
                    no_reverse=False, recomb=True, sum_mode=0):
    """compute_results (score_sequences.py:44-211) + ResultTmp.to_df
    (resultsTmp.py:241-314) restated on numpy; `motif` is a dict with
    score_matrix, pmf, min_val, scale, offset, width, motif_id, motif_name.
    Returns a dict of columns (already thresholded, filtered and stably sorted
    by p-value)."""
    W = motif["width"]
    files = sorted(glob.glob(os.path.join(sequence_loc, f"width_{W}", "*.tsv")))
    cols = parse_tsv_rows(files, no_reverse)
    n = len(cols["seq"])
    if n == 0:
        raise ValueError("No result retrieved. Unable to proceed.")
    kmers = np.frombuffer("".join(cols["seq"]).encode(), dtype=np.uint8).reshape(n, W)
    sc, lo, pv = score_kmers(kmers, motif["score_matrix"], motif["pmf"], motif["min_val"],
                             motif["scale"], motif["offset"], sum_mode)
    start = np.array(cols["start"], dtype=np.int64)
    stop = np.array(cols["stop"], dtype=np.int64)
    ref = np.array(cols["ref"], dtype=object)
    # indel reference fix (score_sequences.py:305-307)
    ref[(ref == "ref") & (np.abs(stop - start) != W)] = "non.ref"
    out = {
        "motif_id": np.array([motif["motif_id"]] * n, dtype=object),
        "motif_alt_id": np.array([motif["motif_name"]] * n, dtype=object),
        "sequence_name": np.array(cols["seqname"], dtype=object),
        "start": start, "stop": stop,
        "strand": np.array(cols["strand"], dtype=object),
        "score": lo, "p-value": pv,
    }
    if not no_qvalue:
        out["q-value"] = fdr_bh(pv)
    out["matched_sequence"] = np.array(cols["seq"], dtype=object)
    out["haplotype_frequency"] = np.array(cols["freq"], dtype=np.int64)
    out["reference"] = ref
    keep = (out["q-value"] < threshold) if qval_t else (pv < threshold)
    if not recomb:
        keep &= out["haplotype_frequency"] > 0
    idx = np.nonzero(keep)[0]
    idx = idx[np.argsort(pv[idx], kind="stable")]
    res = {k: v[idx] for k, v in out.items()}
    res["_scaled_score"] = sc[idx]
    res["_scanned"] = n
    return res
