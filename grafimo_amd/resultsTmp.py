"""Column store of per-k-mer results (mirror of the reference's ResultTmp, resultsTmp.py:15-451).

Same public methods (``append_list``, ``add_qvalues``, ``to_df``, ``isempty``, ``size``) and the
same filtering rules in ``to_df`` (resultsTmp.py:241-314): strict ``<`` on the q-value
(--qvalueT) or p-value, rows with haplotype_frequency == 0 dropped unless --recomb, ascending
sort by p-value.  The GPU path thresholds on the device and only ever materialises hit rows
here; the class is kept so that code written against the reference keeps working.
The sort is stable (ties keep input order); the reference's quicksort leaves tie order
unspecified.
"""
from typing import List, Optional

import numpy as np
import pandas as pd

from .motif import Motif, is_motif_like

_COLUMNS = ("seqnames", "seqs", "chroms", "starts", "stops", "strands", "scores", "pvalues",
            "frequencies", "references")


class ResultTmp(object):
    def __init__(self):
        for c in _COLUMNS:
            setattr(self, "_" + c, [])
        self._qvalues: List[float] = []

    def size(self) -> int:
        n = len(self._seqnames)
        assert all(len(getattr(self, "_" + c)) == n for c in _COLUMNS)
        return n

    def append_list(self, seqnames, seqs, chroms, starts, stops, strands, scores, pvalues,
                    frequencies, references) -> None:
        args = (seqnames, seqs, chroms, starts, stops, strands, scores, pvalues, frequencies, references)
        for a in args:
            if not isinstance(a, list):
                raise TypeError(f"\n\nERROR: Expected list, got {type(a).__name__}.\n")
        for c, a in zip(_COLUMNS, args):
            getattr(self, "_" + c).extend(a)

    def add_qvalues(self, qvalues: List[float]) -> None:
        if not isinstance(qvalues, list):
            raise TypeError(f"\n\nERROR: Expected list, got {type(qvalues).__name__}.\n")
        self._qvalues = qvalues

    def isempty(self) -> bool:
        return any(len(getattr(self, "_" + c)) == 0 for c in _COLUMNS)

    seqnames = property(lambda self: self._seqnames)
    seqs = property(lambda self: self._seqs)
    chroms = property(lambda self: self._chroms)
    starts = property(lambda self: self._starts)
    stops = property(lambda self: self._stops)
    strands = property(lambda self: self._strands)
    scores = property(lambda self: self._scores)
    pvalues = property(lambda self: self._pvalues)
    qvalues = property(lambda self: self._qvalues)
    frequencies = property(lambda self: self._frequencies)
    references = property(lambda self: self._references)

    def to_df(self, motif: Motif, threshold: float, qvalt: bool, recomb: bool,
              ignore_qvals: Optional[bool] = False) -> pd.DataFrame:
        if not is_motif_like(motif):
            raise TypeError(f"\n\nERROR: Expected Motif, got {type(motif).__name__}.\n")
        if not isinstance(threshold, float):
            raise TypeError(f"\n\nERROR: Expected float, got {type(threshold).__name__}.\n")
        if threshold <= 0 or threshold > 1:
            raise ValueError("\n\nERROR: The threshold must be between 0 and 1.\n")
        if not isinstance(qvalt, bool) or not isinstance(ignore_qvals, bool):
            raise ValueError("\n\nERROR: Expected bool.\n")
        if not isinstance(recomb, bool):
            raise TypeError(f"Expected bool, got {type(recomb).__name__}.\n")
        if qvalt:
            assert bool(self._qvalues) and not ignore_qvals
        return build_frame(motif, self._seqnames, self._starts, self._stops, self._strands,
                           self._scores, self._pvalues, None if ignore_qvals else self._qvalues,
                           self._seqs, self._frequencies, self._references,
                           threshold=threshold, qvalt=qvalt, recomb=recomb)


def build_frame(motif, seqnames, starts, stops, strands, scores, pvalues, qvalues, seqs,
                frequencies, references, threshold=None, qvalt=False, recomb=True) -> pd.DataFrame:
    """The report table (column names and order of resultsTmp.py:270-301).  ``threshold=None``
    means the rows were already thresholded (device path)."""
    n = len(seqnames)
    data = {
        "motif_id": [motif.motif_id] * n,
        "motif_alt_id": [motif.motif_name] * n,
        "sequence_name": seqnames,
        "start": starts,
        "stop": stops,
        "strand": strands,
        "score": scores,
        "p-value": pvalues,
    }
    if qvalues is not None:
        data["q-value"] = qvalues
    data["matched_sequence"] = seqs
    data["haplotype_frequency"] = frequencies
    data["reference"] = references
    df = pd.DataFrame(data)
    if threshold is not None:
        df = df[df["q-value"] < threshold] if qvalt else df[df["p-value"] < threshold]
    if not recomb:
        df = df[df["haplotype_frequency"] > 0]
    df = df.sort_values(["p-value"], ascending=True, kind="stable")
    df.reset_index(drop=True, inplace=True)
    return df


def build_frame_sorted(motif, seqnames, starts, stops, strands, scores, pvalues, qvalues, seqs, frequencies, references,
                       recomb=True) -> pd.DataFrame:
    """build_frame for rows that are numpy arrays already and already thresholded (the device paths): the --recomb filter
    (resultsTmp.py:309-310) and the stable sort by p-value (:312) are done on the arrays, and the table is made once from
    the final columns -- the same table as build_frame(threshold=None), at a fraction of pandas' per-call costs (a
    boolean row filter and sort_values each rebuild every column)."""
    pvalues = np.asarray(pvalues, dtype=np.float64)
    frequencies = np.asarray(frequencies)
    idx = np.arange(len(pvalues)) if recomb else np.nonzero(frequencies > 0)[0]
    idx = idx[np.argsort(pvalues[idx], kind="stable")]
    n = len(idx)
    take = lambda a: np.asarray(a)[idx]     # noqa: E731
    data = {
        "motif_id": np.full(n, motif.motif_id, dtype=object),
        "motif_alt_id": np.full(n, motif.motif_name, dtype=object),
        "sequence_name": take(seqnames),
        "start": take(starts),
        "stop": take(stops),
        "strand": take(strands),
        "score": take(scores),
        "p-value": pvalues[idx],
    }
    if qvalues is not None:
        data["q-value"] = take(qvalues)
    data["matched_sequence"] = take(seqs)
    data["haplotype_frequency"] = frequencies[idx]
    data["reference"] = take(references)
    return pd.DataFrame(data, copy=False)
