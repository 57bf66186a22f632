"""Drop-in for the reference's native ``motif_processing`` module (seam S1 of SURVEY.md section 8b).

The reference builds ``motif_processing.pyx`` as a top-level extension module and imports six
callables from it (motif_ops.py:29-35).  This module exports the same six names with the same
signatures and error behaviour; the two numeric ones are backed by libgrafimo_hip.so:

  compute_log_odds -> gfm_compute_log_odds (host f64, libm log)      pyx:512-548 / :444-507
  comp_pval_mat    -> gfm_comp_pval_mat    (Staden DP ON THE GPU)    pyx:608-632 / :552-603

The other four are tiny host-side table preparations kept in Python exactly as the
reference has them (file parsing and 4xW elementwise arithmetic).
"""
from typing import Dict, List

import numpy as np

from .grafimo_errors import BGFileError, MotifProcessingError
from .utils import DNA_ALPHABET, exception_handler, isListEqual
from . import _native as nv
from .motif import dense_bg, dense_score_matrix


def read_bg_file(bg_file: str, debug: bool) -> Dict[str, float]:
    """0-order Markov background file -> {nuc: prob} in file order (pyx:40-99).
    Lines starting with '#' are skipped; reading stops once A,C,G,T were all seen."""
    bg_dict: Dict[str, float] = {}
    try:
        with open(bg_file, mode="r") as ifstream:
            for line in ifstream:
                if line[0] == "#":
                    continue
                if line[0].upper() in DNA_ALPHABET:
                    nuc, prob_str = line.split()
                    prob = float(prob_str)
                    assert prob > 0
                    if nuc.upper() in bg_dict:
                        exception_handler(BGFileError, "Found two times %s.\n" % nuc, debug)
                    bg_dict[nuc.upper()] = prob
                else:
                    errmsg = "Found symbol not part of the DNA alphabet: %s\n" % line[0]
                    exception_handler(ValueError, errmsg, debug)
                if len(bg_dict) == len(DNA_ALPHABET):
                    break
    except Exception:
        exception_handler(BGFileError, "An error occurred while parsing %s" % bg_file, debug)
    return bg_dict


def get_uniform_bg(alphabet: List[str], debug: bool) -> Dict[str, float]:
    """Uniform background over the alphabet (pyx:134-161)."""
    unifp = 1.0 / float(len(alphabet))
    return {nuc: unifp for nuc in alphabet}


def _check_common(width, alphabet, pseudocount, debug):
    if not isListEqual(alphabet, DNA_ALPHABET):
        exception_handler(ValueError, "The motif is not built on DNA alphabet.\n", debug)
    if pseudocount is not None and pseudocount <= 0:
        exception_handler(ValueError, "Pseudocount values must be > 0.\n", debug)
    if width <= 0:
        exception_handler(ValueError, "Forbidden motif width.\n", debug)


def apply_pseudocount_jaspar_transfac_pfm(counts_matrix, probs_matrix, pseudocount, bgs, width,
                                          alphabet, nucsmap, debug) -> np.ndarray:
    """Count formats (pyx:192-261): per column site_counts = int(sum(counts[:, j])) (C int
    truncation), p' = (p * site_counts + pseudo * bg) / (site_counts + pseudo)."""
    counts_matrix = np.asarray(counts_matrix)
    probs_matrix = np.asarray(probs_matrix)
    if counts_matrix.size == 0 or sum(sum(counts_matrix)) == 0:
        exception_handler(ValueError, "Motif counts matrix is empty.\n", debug)
    if probs_matrix.size == 0 or sum(sum(probs_matrix)) == 0:
        exception_handler(ValueError, "Motif probability matrix is empty.\n", debug)
    _check_common(width, alphabet, pseudocount, debug)
    pseudo = float(pseudocount)
    out = np.zeros(counts_matrix.shape, dtype=np.double)
    for j in range(width):
        site_counts = int(sum(counts_matrix[:, j]))
        total_counts = float(site_counts) + pseudo
        for nuc in alphabet:
            i = nucsmap[nuc]
            bg = float(bgs[nuc])
            assert bg > 0
            out[i, j] = ((float(probs_matrix[i, j]) * float(site_counts)) + (pseudo * bg)) / total_counts
    assert out.size != 0 and sum(sum(out)) != 0
    return out


def apply_pseudocount_meme(probs_matrix, pseudocount, site_counts, width, bgs, alphabet, nucsmap,
                           debug) -> np.ndarray:
    """MEME (pyx:313-384): p' = (p * nsites + pseudo * bg) / (nsites + pseudo)."""
    probs_matrix = np.asarray(probs_matrix)
    if probs_matrix.size == 0 or sum(sum(probs_matrix)) == 0:
        exception_handler(ValueError, "The probability matrix is empty.\n", debug)
    if pseudocount <= 0:
        exception_handler(ValueError, "The pseudocount must be > 0.", debug)
    if site_counts <= 0:
        exception_handler(ValueError, "The site counts must be > 0.\n", debug)
    _check_common(width, alphabet, None, debug)
    pseudo = float(pseudocount)
    nsites = int(site_counts)
    total_counts = float(nsites) + pseudo
    out = np.zeros(probs_matrix.shape, dtype=np.double)
    for j in range(width):
        for nuc in alphabet:
            i = nucsmap[nuc]
            bg = float(bgs[nuc])
            assert bg > 0
            out[i, j] = ((float(probs_matrix[i, j]) * nsites) + (pseudo * bg)) / total_counts
    return out


def _dense(matrix, nucsmap):
    """rows of `matrix` (nucsmap order) -> contiguous f64 [4, W] in A,C,G,T order"""
    m = np.asarray(matrix, dtype=np.float64)
    return np.ascontiguousarray(m[[nucsmap[n] for n in DNA_ALPHABET]])


def compute_log_odds(probs_matrix, width: int, bgs: Dict, alphabet: List[str], nucsmap: dict,
                     debug: bool) -> np.ndarray:
    """lo[n, j] = ln(p[n, j] / bg[n]) * 1.44269504 (pyx:444-507), rows in nucsmap order."""
    probs_matrix = np.asarray(probs_matrix)
    if probs_matrix.size == 0 or sum(sum(probs_matrix)) == 0:
        exception_handler(ValueError, "The motif probability matrix is empty.\n", debug)
    _check_common(width, alphabet, None, debug)
    dense = _dense(probs_matrix, nucsmap)
    bg = np.array([float(bgs[n]) for n in DNA_ALPHABET], dtype=np.float64)
    out = np.empty_like(dense)
    rc = nv.lib().gfm_compute_log_odds(nv.ptr(dense), int(width), nv.ptr(bg), nv.ptr(out))
    if rc == nv.GFM_ERR_ASSERT:
        raise AssertionError(nv.lib().gfm_last_error().decode())
    nv.check(rc)
    res = np.zeros(probs_matrix.shape, dtype=np.double)
    for k, n in enumerate(DNA_ALPHABET):
        res[nucsmap[n]] = out[k]
    return res


def comp_pval_mat(motif, debug: bool) -> np.ndarray:
    """Score-distribution DP (Staden 1994; pyx:552-603) on the GPU -> f64 [1000*W + 1],
    the last DP row, un-normalised, bit-identical to the reference's."""
    if not motif.is_scaled:
        exception_handler(MotifProcessingError,
                          "The motif score matrix has not been scaled yet.\n", debug)
    sm = dense_score_matrix(motif)   # any object with the reference Motif's members (motif.py:323-457)
    bg = dense_bg(motif)
    out = np.empty(nv.RANGE * motif.width + 1, dtype=np.float64)
    rc = nv.lib().gfm_comp_pval_mat(nv.ptr(sm), int(motif.width), nv.ptr(bg), nv.ptr(out))
    if rc == nv.GFM_ERR_ASSERT:
        raise AssertionError(nv.lib().gfm_last_error().decode())
    nv.check(rc)
    return out
