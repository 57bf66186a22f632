"""Report writers for the hit table (mirror of res_writer.py:41-303, :415-437).

``write_results`` writes ``<prefix>.tsv`` (``DataFrame.to_csv(sep="\t")`` with the unnamed index
column), ``<prefix>.html`` (``to_html``) and ``<prefix>.gff`` (GFF3) into the output directory with
the reference's naming rules (``grafimo_out_<PID>_<MOTIFID>`` by default, per-motif prefix when
several motifs share a user-given directory).  ``writeGFF3`` keeps the reference's literal quirks:
score rounded to one decimal, '-' strand rows written with start/stop swapped,
``np.format_float_scientific(exp_digits=2)``, and the doubled '=' of ``"=".join(["pvalue=", ...])``
and ``"=".join(["sequence=", seq, ";\n"])`` (res_writer.py:287-289).
Rendering PNGs of the top regions (``--top-graphs``) shells out to ``vg`` and ``dot`` in the
reference and is out of scope here; asking for it raises.
"""
import os
import sys
import time

import numpy as np
import pandas as pd

from .motif import Motif, is_motif_like
from .utils import PHASE, SOURCE, TP, exception_handler

DEFAULT_OUTDIR = "default_out_dir_name"   # utils.py:28


def _columns(data: pd.DataFrame, no_qvalue: bool, debug: bool):
    """The column lists writeGFF3 walks (utils.dftolist, utils.py:496-560)."""
    if not isinstance(data, pd.DataFrame):
        exception_handler(TypeError, f"Expected DataFrame, got {type(data).__name__}.\n", debug)
    if len(data) == 0:
        exception_handler(ValueError, "Empty DataFrames cannot be converted to lists of values.\n", debug)
    if len(data.columns) > 12 or len(data.columns) < 11:
        exception_handler(ValueError, "Not enough values to extract from the DataFrame.\n", debug)
    names = ["motif_id", "motif_alt_id", "sequence_name", "start", "stop", "strand", "score",
             "p-value", "matched_sequence", "haplotype_frequency", "reference"]
    if not no_qvalue:
        names.append("q-value")
    return [data[c].tolist() for c in names]


def writeGFF3(prefix: str, data: pd.DataFrame, no_qvalue: bool, debug: bool) -> None:
    """GFF3 annotation of the motif occurrence candidates (res_writer.py:213-303)."""
    if not isinstance(prefix, str):
        exception_handler(TypeError, f"Expected str, got {type(prefix).__name__}.\n", debug)
    if not isinstance(no_qvalue, bool):
        exception_handler(TypeError, f"Expected bool, got {type(no_qvalue).__name__}.\n", debug)
    if isinstance(data, pd.DataFrame) and not no_qvalue and "q-value" not in data.columns:
        exception_handler(ValueError, "Q-values columns seems to be missing.\n", debug)
    cols = _columns(data, no_qvalue, debug)
    mids, mnames, seqnames, starts, stops, strands, scores, pvals, seqs, _freqs, refs = cols[:11]
    qvals = cols[11] if not no_qvalue else None
    gfffn = ".".join([prefix, "gff"])
    lines = ["##gff-version 3\n"]
    for i in range(len(seqnames)):
        seqname = seqnames[i]
        strand = strands[i]
        lo, hi = (stops[i], starts[i]) if strand == "-" else (starts[i], stops[i])
        atts = [
            "".join(["Name=", mids[i], "_", seqname, strand, ":", refs[i]]),
            "=".join(["Alias", mnames[i]]),
            "=".join(["ID", mids[i], "-", mnames[i], "-", seqname]),
            "=".join(["pvalue=", str(np.format_float_scientific(pvals[i], exp_digits=2))]),
        ]
        if qvals is not None:
            atts.append("=".join(["qvalue", str(np.format_float_scientific(qvals[i], exp_digits=2))]))
        atts.append("=".join(["sequence=", seqs[i], ";\n"]))
        lines.append("\t".join(
            [seqname.split(":")[0], SOURCE, TP, str(lo), str(hi), str(round(scores[i], 1)), strand,
             PHASE, ";".join(atts)]))
    try:
        with open(gfffn, mode="w+") as ofstream:
            ofstream.write("".join(lines))
    except OSError:
        exception_handler(OSError, f"An error ocurred while writing {gfffn}.\n", debug)


def write_results(results: pd.DataFrame, motif: Motif, motif_num: int, args_obj, debug: bool) -> None:
    """TSV + HTML + GFF3 reports in the output directory (res_writer.py:41-208).
    ``args_obj`` needs .outdir, .noqvalue, .top_graphs, .verbose."""
    if not isinstance(results, pd.DataFrame):
        exception_handler(TypeError, f"Expected DataFrame, got {type(results).__name__}.\n", debug)
    if len(results) == 0:
        exception_handler(ValueError, "No potential motif occurrence retreived.\n", debug)
    if not is_motif_like(motif):
        exception_handler(TypeError, f"Expected Motif, got {type(motif).__name__}.\n", debug)
    if not isinstance(motif_num, int):
        exception_handler(TypeError, f"Expected int, got {type(motif_num).__name__}.\n", debug)
    if motif_num <= 0:
        exception_handler(ValueError, "No motif searched. Probably something went wrong.\n", debug)
    outdir = getattr(args_obj, "outdir", DEFAULT_OUTDIR)
    no_qvalue = bool(args_obj.noqvalue)
    verbose = bool(getattr(args_obj, "verbose", False))
    if int(getattr(args_obj, "top_graphs", 0) or 0) > 0:
        exception_handler(NotImplementedError,
                          "--top-graphs needs the external vg and dot binaries (out of scope).\n", debug)
    dirname_default = outdir == DEFAULT_OUTDIR
    if dirname_default:
        outdir = "_".join(["grafimo_out", str(os.getpid()), motif.motif_id])
    os.makedirs(outdir, exist_ok=True)
    print(f"\nWriting results in {outdir}.\n")
    prefix = "grafimo_out"
    if not dirname_default and motif_num > 1:
        prefix = "_".join(["grafimo_out", motif.motif_id])
    path = os.path.join(outdir, prefix)
    t = time.time()
    results.to_csv(".".join([path, "tsv"]), sep="\t", encoding="utf-8")
    if verbose:
        print("%s.tsv written in %.2fs" % (prefix, time.time() - t))
    t = time.time()
    results.to_html(".".join([path, "html"]))
    if verbose:
        print("%s.html written in %.2fs" % (prefix, time.time() - t))
    t = time.time()
    writeGFF3(path, results, no_qvalue, debug)
    if verbose:
        print("%s.gff written in %.2fs" % (prefix, time.time() - t))


def print_results(results: pd.DataFrame, debug: bool) -> None:
    """--text-only: the table on stdout, all columns shown (res_writer.py:415-437)."""
    if not isinstance(results, pd.DataFrame):
        exception_handler(TypeError, f"Expected DataFrame, got {type(results).__name__}.\n", debug)
    pd.set_option("display.max_columns", None)
    print()
    print(results)
    pd.reset_option("display.max_rows")
