"""Motif file parsing and preprocessing: PWM file -> ``Motif`` with scaled scoring matrix and
score distribution.

Same public names, argument meaning and error behaviour as the reference's motif_ops.py
(paths relative to /root/reference/src/grafimo/):

  build_motif_jaspar / _meme / _transfac / _pfm     motif_ops.py:51, :237, :640, :809
  process_motif_for_logodds                         motif_ops.py:971-1022
  scale_pwm                                         motif_ops.py:1027-1111
  get_motif_pwm                                     motif_ops.py:1116-1184
  pseudo_bg / average_bg_with_rc / norm_bg          motif_ops.py:1189-1302
  norm_motif                                        motif_ops.py:1307-1362

Structure is our own: each format has a small reader returning a ``_RawMotif`` and one
shared ``_finish_motif`` applies background, normalisation and pseudocounts.  The numeric
tail (log-odds, integer scaling, p-value DP) goes through libgrafimo_hip.so; the DP runs on
the GPU.  ``pvalue_matrix=False`` (an extra keyword the reference does not have) skips the DP
so that a motif can be parsed on a GPU-less host and get its distribution when it is uploaded.
"""
import os
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import pandas as pd

from . import _native as nv
from .grafimo_errors import BGFileError, MotifFileFormatError, MotifFileReadError
from .motif import Motif, is_motif_like
from .motif_processing import (apply_pseudocount_jaspar_transfac_pfm, apply_pseudocount_meme,
                               comp_pval_mat, compute_log_odds, get_uniform_bg, read_bg_file)
from .utils import (DNA_ALPHABET, PSEUDOBG, RANGE, REV_COMPL, UNIF, almost_equal,
                    exception_handler, isListEqual)


# ------------------------------------------------------------------------------ sniffers
def _is_numeric(s: str) -> bool:
    try:
        float(s)
        return True
    except ValueError:
        return False


def _check_file(motif_file, debug):
    if not isinstance(motif_file, str):
        exception_handler(TypeError, f"Expected str, got {type(motif_file).__name__}.\n", debug)
    if not os.path.isfile(motif_file):
        exception_handler(FileNotFoundError, f"Unable to locate {motif_file}.\n", debug)
    if os.stat(motif_file).st_size == 0:
        exception_handler(EOFError, f"{motif_file} seems to be empty.\n", debug)


def is_jaspar(motif_file: str, debug: bool) -> bool:
    """utils.py:212-260: '.jaspar' suffix, '>' header, rows 'X [ n n ... ]'."""
    _check_file(motif_file, debug)
    if motif_file.split(".")[-1] != "jaspar":
        return False
    with open(motif_file) as fh:
        if not fh.readline().strip().startswith(">"):
            return False
        for line in fh:
            tok = line.strip().split()
            if not tok:
                return False
            if len(tok) < 3 or tok[1] != "[" or tok[-1] != "]":
                return False
            if not all(_is_numeric(c) for c in tok[2:-1]):
                return False
    return True


def is_meme(motif_file: str, debug: bool) -> bool:
    """utils.py:262-298: some line starts with 'MEME version'."""
    _check_file(motif_file, debug)
    with open(motif_file) as fh:
        return any(line.startswith("MEME version") for line in fh)


def is_transfac(motif_file: str, debug: bool) -> bool:
    """utils.py:300-365: two-letter field codes, AC + ID + PO/P0 present, positions 1..W."""
    _check_file(motif_file, debug)
    seen = {"AC": False, "ID": False, "PO": False}
    width = 0
    with open(motif_file) as fh:
        for line in fh:
            line = line.strip()
            if not line:
                continue
            parts = line.split(None, 1)
            field = parts[0].strip()
            if len(field) != 2:
                return False
            if len(parts) != 2:
                continue
            value = parts[1].strip()
            if field in seen:
                if not value:
                    return False
                if field in ("P0", "PO") and value.split()[:4] != DNA_ALPHABET:
                    return False
                seen[field] = True
            try:
                position = int(field)
            except ValueError:
                continue
            if width == 0 and position == 0:
                return False
            width += 1
            if width != position:
                return False
    return sum(seen.values()) == 3


def is_pfm(motif_file: str, debug: bool) -> bool:
    """utils.py:367-405: every non-header token is numeric."""
    _check_file(motif_file, debug)
    with open(motif_file) as fh:
        for line in fh:
            if line.startswith(">"):
                continue
            if not all(_is_numeric(c) for c in line.strip().split()):
                return False
    return True


# ------------------------------------------------------------------------------ background
def average_bg_with_rc(bgs: Dict, debug: bool) -> Dict[str, float]:
    """bg[A]=bg[T]=(A+T)/2, bg[C]=bg[G]=(C+G)/2; insertion order follows the input dict
    (motif_ops.py:1233-1263)."""
    if not isinstance(bgs, dict):
        exception_handler(TypeError, f"Expected dict, got {type(bgs).__name__}.\n", debug)
    out: Dict[str, float] = {}
    for nuc in bgs.keys():
        rc = REV_COMPL[nuc.upper()]
        if REV_COMPL[rc] == nuc and ord(nuc) < ord(rc):
            avg = np.double((bgs[nuc] + bgs[rc]) / np.double(2))
            out[nuc] = avg
            out[rc] = avg
    return out


def norm_bg(bgs: Dict, debug: bool) -> Dict[str, float]:
    """(bg + 5e-7) / (sum(bg) + 4 * 5e-7), total accumulated in dict order
    (motif_ops.py:1268-1302)."""
    if not isinstance(bgs, dict):
        exception_handler(TypeError, f"Expected dict, got {type(bgs).__name__}.\n", debug)
    tot = np.double(len(bgs) * PSEUDOBG)
    for nuc in bgs.keys():
        tot += np.double(bgs[nuc])
    assert tot > 0
    return {nuc: np.double((bgs[nuc] + PSEUDOBG) / tot) for nuc in bgs.keys()}


def pseudo_bg(bgs: Dict, no_reverse: bool, debug: bool) -> Dict[str, float]:
    """motif_ops.py:1189-1228."""
    if not isinstance(bgs, dict):
        exception_handler(TypeError, f"Expected dict, got {type(bgs).__name__}.\n", debug)
    if not isinstance(no_reverse, bool):
        exception_handler(TypeError, f"Expected bool, got {type(no_reverse).__name__}.\n", debug)
    return norm_bg(bgs if no_reverse else average_bg_with_rc(bgs, debug), debug)


def norm_motif(motif_probs: pd.DataFrame, motif_width: int, alphabet: List[str],
               debug: bool) -> pd.DataFrame:
    """Re-normalise columns whose sum differs from 1 by more than 1e-5 (motif_ops.py:1307-1362)."""
    if not isinstance(motif_probs, pd.DataFrame):
        exception_handler(TypeError, f"Expected DataFrame, got {type(motif_probs).__name__}.\n", debug)
    if not isinstance(motif_width, int):
        exception_handler(TypeError, f"Expected int, got {type(motif_width).__name__}.\n", debug)
    if motif_width <= 0:
        exception_handler(ValueError, "Forbidden motif width.\n", debug)
    if any(nuc not in DNA_ALPHABET for nuc in alphabet):
        exception_handler(ValueError, "The motif is not built on DNA alphabet.\n", debug)
    values = motif_probs.to_numpy(dtype=np.float64, copy=True)
    rows = [list(motif_probs.index).index(nuc) for nuc in alphabet]
    for j in range(motif_width):
        tot = np.double(0)
        for r in rows:
            tot += values[r, j]
        assert tot != 0
        if not almost_equal(1, tot, 0.00001):
            for r in rows:
                values[r, j] = np.double(values[r, j] / tot)
    return pd.DataFrame(values, index=motif_probs.index, columns=motif_probs.columns)


def _load_bg(bg_file: str, alphabet: List[str], no_reverse: bool, debug: bool) -> Dict[str, float]:
    if bg_file == UNIF:
        bgs = get_uniform_bg(alphabet, debug)
    elif os.path.isfile(bg_file):
        bgs = read_bg_file(bg_file, debug)
    else:
        exception_handler(BGFileError, f"Unable to parse {bg_file}.\n", debug)
    return pseudo_bg(bgs, no_reverse, debug)


# ------------------------------------------------------------------------------ numeric tail
def scale_pwm(motif_matrix: np.ndarray, alphabet: List[str], motif_width: int, nucsmap: dict,
              debug: bool) -> Tuple[np.ndarray, int, int, int, np.double]:
    """Integer scaling of the log-odds matrix to [0, 1000] (motif_ops.py:1027-1111) through
    gfm_scale_pwm.  Returns (scaled int matrix in the input's row order, min, max, scale, offset)."""
    if not isinstance(motif_matrix, np.ndarray):
        exception_handler(TypeError, f"Expected ndarray, got {type(motif_matrix).__name__}.\n", debug)
    if motif_matrix.size == 0 or sum(sum(motif_matrix)) == 0:
        exception_handler(ValueError, "The motif log-odds natrix is empty.\n", debug)
    if not isinstance(alphabet, list):
        exception_handler(TypeError, f"Expected list, got {type(alphabet).__name__}.\n", debug)
    if not isListEqual(alphabet, DNA_ALPHABET):
        exception_handler(ValueError, "The motif is not built on DNA alphabet.\n", debug)
    if not isinstance(motif_width, int):
        exception_handler(TypeError, f"Expected int, got {type(motif_width).__name__}.\n", debug)
    if motif_width <= 0:
        exception_handler(ValueError, "Forbidden motif width.\n", debug)
    if not isinstance(nucsmap, dict):
        exception_handler(TypeError, f"Expected dict, got {type(nucsmap).__name__}.\n", debug)
    import ctypes
    lo = np.ascontiguousarray(motif_matrix, dtype=np.float64)
    sm = np.empty(lo.shape, dtype=np.int64)
    mn, mx, sc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    off = ctypes.c_double()
    nv.check(nv.lib().gfm_scale_pwm(nv.ptr(lo), int(motif_width), nv.ptr(sm), ctypes.byref(mn),
                                    ctypes.byref(mx), ctypes.byref(sc), ctypes.byref(off)))
    return sm.astype(int), int(mn.value), int(mx.value), int(sc.value), np.double(off.value)


def process_motif_for_logodds(motif: Motif, debug: bool, pvalue_matrix: bool = True) -> Motif:
    """log-odds -> integer scaling -> p-value DP (motif_ops.py:971-1022)."""
    if not is_motif_like(motif):
        exception_handler(TypeError, f"Expected Motif, got {type(motif).__name__}.\n", debug)
    lo = compute_log_odds(motif.count_matrix, motif.width, motif.bg, motif.alphabet,
                          motif.nucsmap, debug)
    motif.set_motif_score_matrix(lo)
    scaled, min_val, max_val, scale, offset = scale_pwm(
        motif.score_matrix, motif.alphabet, motif.width, motif.nucsmap, debug)
    motif.set_motif_score_matrix(scaled)
    motif.set_is_scaled()
    motif.set_scale(scale)
    motif.set_min_val(min_val)
    motif.set_max_val(max_val)
    motif.set_offset(offset)
    if pvalue_matrix:
        motif.set_motif_pval_matrix(comp_pval_mat(motif, debug))
    return motif


# ------------------------------------------------------------------------------ readers
@dataclass
class _RawMotif:
    motif_id: str
    motif_name: str
    rows: List[str]            # nucleotide of each matrix row, file order
    values: np.ndarray         # counts (count formats) or probabilities (MEME), [4, W]
    nsites: Optional[int] = None


def _finish_count_motif(raw: _RawMotif, bg_file, pseudocount, no_reverse, debug) -> Motif:
    """Shared tail of the JASPAR / TRANSFAC / PFM readers."""
    nucsmap = {nuc: i for i, nuc in enumerate(raw.rows)}
    alphabet = sorted(raw.rows)
    counts = pd.DataFrame(raw.values, index=raw.rows)
    width = int(counts.shape[1])
    bgs = _load_bg(bg_file, alphabet, no_reverse, debug)
    probs = norm_motif(counts / counts.sum(0), width, alphabet, debug)
    probs = apply_pseudocount_jaspar_transfac_pfm(
        counts.to_numpy(), probs.to_numpy(), pseudocount, bgs, width, alphabet, nucsmap, debug)
    motif = Motif(probs, width, alphabet, raw.motif_id, raw.motif_name, nucsmap)
    motif.set_bg(bgs)
    return motif


def _read_jaspar(motif_file: str, debug: bool) -> _RawMotif:
    """'>ID<TAB>NAME' then one 'X [ c c ... ]' row per nucleotide (motif_ops.py:126-232)."""
    rows, counts = [], []
    try:
        with open(motif_file) as fh:
            header = fh.readline().strip()[1:]
            if not header:
                exception_handler(IOError, f"{motif_file} seems to empty.\n", debug)
            motif_id, motif_name = header.split("\t")[0:2]
            for line in fh:
                line = line.strip()
                if not line:
                    break
                rows.append(line[:1].upper())
                counts.append([float(x) for x in line[1:].split()[1:][:-1]])
        if not rows:
            exception_handler(IOError, f"{motif_file} seems to be empty.\n", debug)
    except Exception:
        exception_handler(MotifFileReadError, f"An error occurred while reading {motif_file}.\n", debug)
    if any(len(c) != len(counts[0]) for c in counts):
        exception_handler(ValueError, "Motif counts width mismatch.\n", debug)
    return _RawMotif(motif_id, motif_name, rows, np.array(counts, dtype=np.float64))


def _read_transfac(motif_file: str, debug: bool) -> _RawMotif:
    """AC = id, ID = name, PO/P0 header then 'NN a c g t' rows (motif_ops.py:701-804)."""
    motif_id = motif_name = ""
    rows: List[str] = []
    cols: Dict[str, List[float]] = {}
    try:
        with open(motif_file) as fh:
            it = iter(fh)
            for line in it:
                line = line.strip()
                if not line:
                    continue
                parts = line.split(None, 1)
                field = parts[0].strip()
                if field == "AC":
                    motif_id = parts[1].strip()
                elif field == "ID":
                    motif_name = parts[1].strip()
                elif field in ("P0", "PO"):
                    rows = parts[1].strip().split()[:4]
                    assert rows == DNA_ALPHABET
                    cols = {nt: [] for nt in rows}
                    width = 0
                    for cline in it:
                        cparts = cline.strip().split(None, 1)
                        try:
                            position = int(cparts[0].strip())
                        except ValueError:
                            break
                        if len(cparts) != 2:
                            exception_handler(ValueError, f"Invalid count line seen in {motif_file}", debug)
                        width += 1
                        if position != width:
                            exception_handler(ValueError, "Mismatching motif width and position.", debug)
                        count = cparts[1].strip().split()[:4]
                        if len(count) != 4:
                            exception_handler(ValueError, "Perhaps the input motif is not a DNA motif", debug)
                        for nt, c in zip(rows, count):
                            cols[nt].append(float(c))
    except Exception:
        exception_handler(OSError, f"An error occurred while parsing {motif_file}.", debug)
    if any(len(cols[DNA_ALPHABET[0]]) != len(cols[nt]) for nt in cols):
        exception_handler(ValueError, "Motif width mismatch in counts.", debug)
    return _RawMotif(motif_id, motif_name, rows, np.array([cols[nt] for nt in rows], dtype=np.float64))


def _read_pfm(motif_file: str, debug: bool) -> _RawMotif:
    """Optional '>ID NAME' header, then four rows of counts A,C,G,T (motif_ops.py:871-968)."""
    motif_id = motif_name = ""
    counts = []
    try:
        with open(motif_file) as fh:
            for line in fh:
                line = line.strip()
                if not line:
                    exception_handler(ValueError, f"{motif_file} seems empty.", debug)
                if line.startswith(">"):
                    motif_id, motif_name = line[1:].split()
                    continue
                counts.append([float(x) for x in line.split()])
        if len(counts) < 2:
            exception_handler(IOError, f"{motif_file} seems to be empty or that it has missing data.", debug)
    except Exception:
        exception_handler(OSError, f"An error occurred while parsing {motif_file}.", debug)
    assert len(counts) == 4
    if any(len(c) != len(counts[0]) for c in counts):
        exception_handler(ValueError, "Mismatch in counts length.", debug)
    if not motif_name and not motif_id:
        motif_id = motif_name = os.path.basename(motif_file)
    return _RawMotif(motif_id, motif_name, list(DNA_ALPHABET), np.array(counts, dtype=np.float64))


def _read_meme(motif_file: str, debug: bool) -> Tuple[List[str], List[_RawMotif]]:
    """MEME text format, any number of motifs (motif_ops.py:364-637): 'ALPHABET= ACGT',
    per motif 'MOTIF id [name]', 'letter-probability matrix: ... w= W nsites= N E= x',
    then W rows of four probabilities."""
    raws: List[_RawMotif] = []
    alphabet: List[str] = []
    try:
        with open(motif_file) as fh:
            lines = fh.readlines()
        i = 0
        while i < len(lines) and not lines[i].startswith("ALPHABET"):
            i += 1
        if i == len(lines):
            exception_handler(EOFError, f"Unexpected EOF reached, unable to parse {motif_file}.\n", debug)
        if lines[i].strip().replace("ALPHABET= ", "") != "ACGT":
            exception_handler(ValueError, "The motif is not built on DNA alphabet.\n", debug)
        alphabet = sorted("ACGT")
        i += 1
        while True:
            while i < len(lines) and not lines[i].startswith("MOTIF"):
                i += 1
            if i == len(lines):
                break
            ids = lines[i].split()
            if len(ids) == 2:
                motif_id = motif_name = ids[1]
            else:
                motif_id, motif_name = ids[1:3]
            i += 1
            while i < len(lines) and not lines[i].startswith("letter-probability matrix:"):
                i += 1
            stat = lines[i]
            width = int(stat.split("w=")[1].split()[0])
            nsites = int(stat.split("nsites=")[1].split()[0])
            np.double(stat.split("E=")[1].split()[0])  # parsed (and required) like the reference
            i += 1
            cols = [[], [], [], []]
            pos = 0
            while i < len(lines):
                freqs = lines[i].split()
                i += 1
                if len(freqs) != 4:
                    if pos < width:
                        exception_handler(EOFError, "Unexpected end of motif found.\n", debug)
                    break
                for k in range(4):
                    cols[k].append(np.double(freqs[k]))
                pos += 1
            raws.append(_RawMotif(motif_id, motif_name, list(alphabet),
                                  np.array(cols, dtype=np.float64), nsites))
    except Exception:
        exception_handler(MotifFileReadError, f"An error occurred while reading {motif_file}.\n", debug)
    return alphabet, raws


# ------------------------------------------------------------------------------ builders
def _check_build_args(motif_file, bg_file, pseudocount, no_reverse, debug):
    if not isinstance(motif_file, str):
        exception_handler(TypeError, f"Expected str, got {type(motif_file).__name__}.\n", debug)
    if not os.path.isfile(motif_file):
        exception_handler(FileNotFoundError, f"Unable to locate {motif_file}.\n", debug)
    if not isinstance(bg_file, str):
        exception_handler(TypeError, f"Expected str, got {type(bg_file).__name__}.\n", debug)
    if bg_file != UNIF and not os.path.isfile(bg_file):
        exception_handler(FileNotFoundError, f"Unable to locate {bg_file}.\n", debug)
    if pseudocount <= 0:
        exception_handler(ValueError, "Pseudocount value must be positive.\n", debug)
    if not isinstance(no_reverse, bool):
        exception_handler(TypeError, f"Expected bool, got {no_reverse}.\n", debug)


def _timed_process(motif, verbose, debug, pvalue_matrix):
    t = time.time()
    motif = process_motif_for_logodds(motif, debug, pvalue_matrix)
    if verbose:
        print("Motif %s processed in %.2fs" % (motif.motif_id, time.time() - t))
    return motif


def build_motif_jaspar(motif_file: str, bg_file: str, pseudocount: float, no_reverse: bool,
                       verbose: bool, debug: bool, pvalue_matrix: bool = True) -> Motif:
    _check_build_args(motif_file, bg_file, pseudocount, no_reverse, debug)
    raw = _read_jaspar(motif_file, debug)
    motif = _finish_count_motif(raw, bg_file, pseudocount, no_reverse, debug)
    return _timed_process(motif, verbose, debug, pvalue_matrix)


def build_motif_transfac(motif_file: str, bgfile: str, pseudocount: float, no_reverse: bool,
                         verbose: bool, debug: bool, pvalue_matrix: bool = True) -> Motif:
    _check_build_args(motif_file, bgfile, pseudocount, no_reverse, debug)
    raw = _read_transfac(motif_file, debug)
    motif = _finish_count_motif(raw, bgfile, pseudocount, no_reverse, debug)
    return _timed_process(motif, verbose, debug, pvalue_matrix)


def build_motif_pfm(motif_file: str, bgfile: str, pseudocount: float, no_reverse: bool,
                    verbose: bool, debug: bool, pvalue_matrix: bool = True) -> Motif:
    _check_build_args(motif_file, bgfile, pseudocount, no_reverse, debug)
    raw = _read_pfm(motif_file, debug)
    motif = _finish_count_motif(raw, bgfile, pseudocount, no_reverse, debug)
    return _timed_process(motif, verbose, debug, pvalue_matrix)


def build_motif_meme(motif_file: str, bg_file: str, pseudocount: float, no_reverse: bool,
                     cores: int, verbose: bool, debug: bool,
                     pvalue_matrix: bool = True) -> List[Motif]:
    """All motifs of a MEME file.  ``cores`` is accepted for signature compatibility: the
    reference fans motif preprocessing out over a process pool (motif_ops.py:303-311); here the
    expensive part (the DP) runs on the GPU, in the calling process."""
    _check_build_args(motif_file, bg_file, pseudocount, no_reverse, debug)
    if not isinstance(pseudocount, float):
        exception_handler(TypeError, f"Expected float, got {type(pseudocount).__name__}.\n", debug)
    alphabet, raws = _read_meme(motif_file, debug)
    nucsmap = {nuc: i for i, nuc in enumerate(alphabet)}
    bgs = _load_bg(bg_file, alphabet, no_reverse, debug)
    print(f"\nRead {len(raws)} motifs in {motif_file}")
    print("\nProcessing motifs\n")
    motifs = []
    for raw in raws:
        width = int(raw.values.shape[1])
        probs = norm_motif(pd.DataFrame(raw.values, index=alphabet), width, alphabet, debug)
        probs = apply_pseudocount_meme(probs.to_numpy(), pseudocount, raw.nsites, width, bgs,
                                       alphabet, nucsmap, debug)
        motif = Motif(probs, width, alphabet, raw.motif_id, raw.motif_name, nucsmap)
        motif.set_bg(bgs)
        motifs.append(_timed_process(motif, verbose, debug, pvalue_matrix))
    return motifs


def get_motif_pwm(motif_file: str, workflow, cores: int, debug: bool,
                  pvalue_matrix: bool = True) -> List[Motif]:
    """Format dispatch (motif_ops.py:1116-1184); ``workflow`` needs .bgfile, .pseudo,
    .noreverse, .verbose like the reference's Findmotif."""
    if not isinstance(motif_file, str):
        exception_handler(TypeError, f"Expected str, got {type(motif_file).__name__}.\n", debug)
    if not os.path.isfile(motif_file):
        exception_handler(FileNotFoundError, f"Unable to locate {motif_file}.\n", debug)
    a = (workflow.bgfile, workflow.pseudo, workflow.noreverse)
    if is_jaspar(motif_file, debug):
        motif = build_motif_jaspar(motif_file, *a, workflow.verbose, debug, pvalue_matrix)
    elif is_meme(motif_file, debug):
        motif = build_motif_meme(motif_file, *a, cores, workflow.verbose, debug, pvalue_matrix)
    elif is_transfac(motif_file, debug):
        motif = build_motif_transfac(motif_file, *a, workflow.verbose, debug, pvalue_matrix)
    elif is_pfm(motif_file, debug):
        motif = build_motif_pfm(motif_file, *a, workflow.verbose, debug, pvalue_matrix)
    else:
        exception_handler(MotifFileFormatError,
                          "GRAFIMO accepts motifs in JASPAR, MEME, TRANSFAC, or PFM formats.", debug)
    return motif if isinstance(motif, list) else [motif]


def build_motif_meme_host(motif_file, bg_file, pseudocount, no_reverse):
    """MEME file -> motifs without their score distribution (no GPU touched); the DP runs on
    the device when the motif is uploaded (DeviceMotif)."""
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return build_motif_meme(motif_file, bg_file, float(pseudocount), bool(no_reverse), 1, False,
                                True, pvalue_matrix=False)
