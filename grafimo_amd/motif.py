"""Motif carrier type read by the scoring path.

Mirrors the public surface of the reference's ``Motif`` (motif.py:18-483): same
constructor arguments, ``set_*`` methods and read-only properties, same type checks
(``scale`` must be ``int``, ``offset`` must be ``numpy.double`` -- motif.py:242-256), so
code written against the reference object works unchanged.  The numeric members the GPU
path uploads are ``score_matrix`` (int 4xW, rows in ``nucsmap`` order), ``bg``,
``min_val``, ``scale``, ``offset``, ``width`` and ``pval_matrix``.
"""
from typing import Dict, List

import numpy as np
import pandas as pd

from .grafimo_errors import NotValidMotifMatrixError
from .utils import DNA_ALPHABET, isListEqual


def _expect(value, typ, what=None):
    if not isinstance(value, typ):
        name = typ.__name__ if isinstance(typ, type) else str(typ)
        raise TypeError(f"\n\nERROR: Expected {name}, got {type(value).__name__}.\n")


class Motif(object):
    """A DNA motif: probability matrix, scaled scoring matrix, score-distribution
    (``pval_matrix``), scaling parameters, background, identifiers."""

    def __init__(self, count_matrix, width, alphabet, motif_id, motif_name, nucsmap):
        _expect(count_matrix, np.ndarray)
        if count_matrix.size == 0 or sum(sum(count_matrix)) == 0:
            raise NotValidMotifMatrixError("\n\nERROR: Empty motif count matrix.\n")
        _expect(width, int)
        if width <= 0:
            raise ValueError(f"\n\nERROR: Forbidden motif width ({width}).\n")
        _expect(motif_id, str)
        if not motif_id:
            raise ValueError("\n\nERROR: Not valid motif ID.\n")
        _expect(motif_name, str)
        if not motif_name:
            raise ValueError("\n\nERROR: Not valid motif name.\n")
        _expect(alphabet, list)
        if not isListEqual(alphabet, DNA_ALPHABET):
            raise ValueError("\n\nERROR: The motif is not built on DNA alphabet.\n")
        _expect(nucsmap, dict)
        self._count_matrix = count_matrix
        self._score_matrix = None
        self._pval_matrix = None
        self._min_val = -np.inf
        self._max_val = np.inf
        self._scale = -1
        self._offset = 0
        self._bg = None
        self._width = width
        self._motif_id = motif_id
        self._motif_name = motif_name
        self._alphabet = alphabet
        self._nucsmap = nucsmap
        self._is_scaled = False

    # ------------------------------------------------------------------ setters
    def set_motif_matrix(self, motif_matrix):
        _expect(motif_matrix, pd.DataFrame)
        if motif_matrix.empty:
            raise ValueError("\n\nERROR: Empty motif matrix.\n")
        self._count_matrix = motif_matrix

    def set_motif_score_matrix(self, score_matrix):
        if not isinstance(score_matrix, (np.ndarray, pd.DataFrame)):
            raise TypeError(
                f"\n\nERROR: Expected ndarray, got {type(score_matrix).__name__}.\n")
        if score_matrix.size == 0:
            raise ValueError("\n\nERROR: Empty motif score matrix.\n")
        self._score_matrix = score_matrix

    def set_motif_pval_matrix(self, pval_mat):
        _expect(pval_mat, np.ndarray)
        if len(pval_mat) == 0:
            raise ValueError("\n\nERROR: Empty motif p-value matrix.\n")
        if sum(pval_mat[:]) <= 0:
            raise ValueError("\n\nERROR: Not valid motif p-value matrix.\n")
        self._pval_matrix = pval_mat

    def set_min_val(self, min_val):
        _expect(min_val, int)
        if min_val <= -np.inf:
            raise ValueError(f"\n\nERROR: Forbidden value {min_val}.\n")
        self._min_val = min_val

    def set_max_val(self, max_val):
        _expect(max_val, int)
        if max_val >= np.inf:
            raise ValueError(f"\n\nERROR: Forbidden value ({max_val}).\n")
        self._max_val = max_val

    def set_scale(self, scale):
        _expect(scale, int)
        if scale <= 0:
            raise ValueError("\n\nERROR: Scaling factor must be positive integer number.\n")
        self._scale = scale

    def set_offset(self, offset):
        _expect(offset, np.double)
        self._offset = offset

    def set_bg(self, bgs):
        _expect(bgs, dict)
        self._bg = bgs

    def set_width(self, width):
        _expect(width, int)
        if width <= 0:
            raise ValueError("\n\nERROR: Not valid motif width.\n")
        self._width = width

    def set_motif_id(self, motif_id):
        _expect(motif_id, str)
        if not motif_id:
            raise ValueError("\n\nERROR: Not valid motif ID.\n")
        self._motif_id = motif_id

    def set_motif_name(self, motif_name):
        _expect(motif_name, str)
        if not motif_name:
            raise ValueError("\n\nERROR: Not valid motif name.\n")
        self._motif_name = motif_name

    def set_alphabet(self, alphabet):
        _expect(alphabet, list)
        if len(alphabet) == 0:
            raise ValueError("\n\nERROR: Empty motif alphabet.\n")
        if not isListEqual(alphabet, DNA_ALPHABET):
            raise ValueError("\n\nERROR: The motif is not built on DNA alphabet.\n")
        self._alphabet = alphabet

    def set_is_scaled(self):
        if self._is_scaled:
            raise AssertionError("\n\nERROR: The motif matrix has already been scaled.\n")
        self._is_scaled = True

    # ------------------------------------------------------------------ getters
    def _need(self, value, name):
        if value is None or (hasattr(value, "__len__") and len(value) == 0):
            raise AttributeError(f"\n\nERROR: \"self.{name}\" is empty.\n")
        return value

    @property
    def count_matrix(self) -> np.ndarray:
        return self._need(self._count_matrix, "_count_matrix")

    @property
    def score_matrix(self) -> np.ndarray:
        return self._need(self._score_matrix, "_score_matrix")

    @property
    def pval_matrix(self) -> np.ndarray:
        return self._need(self._pval_matrix, "_pval_matrix")

    @property
    def min_val(self) -> int:
        return self._min_val

    @property
    def max_val(self) -> int:
        return self._max_val

    @property
    def scale(self) -> int:
        return self._scale

    @property
    def nucsmap(self) -> Dict[str, int]:
        return self._need(self._nucsmap, "_nucsmap")

    @property
    def offset(self):
        return self._offset

    @property
    def bg(self) -> Dict[str, float]:
        return self._need(self._bg, "_bg")

    @property
    def width(self) -> int:
        return self._width

    @property
    def motif_id(self) -> str:
        return self._need(self._motif_id, "_motif_id")

    @property
    def motif_name(self) -> str:
        return self._need(self._motif_name, "_motif_name")

    @property
    def alphabet(self) -> List[str]:
        return self._need(self._alphabet, "_alphabet")

    @property
    def is_scaled(self) -> bool:
        return self._is_scaled

    def compute_min_value(self) -> None:
        self._min_val = int(np.asarray(self.score_matrix).min())

    def print(self, matrix: str) -> None:
        """Print one of the motif matrices (names as in the reference)."""
        _expect(matrix, str)
        if not matrix:
            raise ValueError("\n\nERROR: Unable to guess what should be printed.\n")
        table = {
            "raw_counts": lambda: self._count_matrix,
            "score_matrix": lambda: self._score_matrix,
            "pval_matrix": lambda: self._pval_matrix,
        }
        if matrix not in table:
            raise ValueError("\n\nERROR: Unable to print the requested matrix.\n")
        print(table[matrix]())

    # ------------------------------------------------------------------ helpers (ours)
    def dense_score_matrix(self) -> np.ndarray:
        return dense_score_matrix(self)

    def dense_bg(self) -> np.ndarray:
        return dense_bg(self)


# ---------------------------------------------------------------------- duck-typed boundary
# What the scoring path reads from a motif (score_sequences.py:294, motif_processing.pyx:577-592).  The
# reference's own ``grafimo.motif.Motif`` carries exactly these, so it can be handed to every entry
# point here without conversion; nothing below needs a method the reference class does not have.
MOTIF_FIELDS = ("score_matrix", "nucsmap", "bg", "min_val", "scale", "offset", "width", "motif_id",
                "motif_name")


def is_motif_like(obj) -> bool:
    """True for this package's Motif, the reference's Motif, or any object with the same members."""
    if isinstance(obj, Motif):
        return True
    return all(hasattr(type(obj), f) or f in getattr(obj, "__dict__", ()) for f in MOTIF_FIELDS)


def dense_score_matrix(motif) -> np.ndarray:
    """motif.score_matrix (rows in ``motif.nucsmap`` order, ndarray or DataFrame) as int64 [4, W] with
    rows A,C,G,T -- the layout of the C ABI."""
    sm = np.asarray(motif.score_matrix)
    idx = [int(motif.nucsmap[n]) for n in DNA_ALPHABET]
    return np.ascontiguousarray(sm[idx], dtype=np.int64)


def dense_bg(motif) -> np.ndarray:
    """motif.bg ({nucleotide: probability}) as f64 [4] in A,C,G,T order."""
    return np.array([float(motif.bg[n]) for n in DNA_ALPHABET], dtype=np.float64)
