"""Argument container for the scoring step.

The reference passes its ``Findmotif`` workflow object (workflow.py:233-634) into
``compute_results`` and ``get_motif_pwm``; only a handful of read-only properties are used on
the hot path.  ``Findmotif`` here carries exactly those (same property names and defaults as
the CLI: __main__.py:119-415), so either this object or the reference's own instance can be
passed to ``grafimo_amd.score_sequences.compute_results``.
"""
from .utils import UNIF


ALL_CHROMS = "use_all_chroms"   # utils.py:22: the default of --chroms-find


class Findmotif(object):
    """cores .. pseudo are what compute_results / get_motif_pwm read; graph_genome .. namemap are what
    scan_graph reads (workflow.py:428-505, :621-630), for the GPU replacement of that step."""

    def __init__(self, cores=1, threshold=1e-4, no_qvalue=False, qval_t=False, no_reverse=False,
                 recomb=False, verbose=False, bgfile=UNIF, pseudo=0.1, graph_genome="", graph_genome_dir="",
                 bedfile="", chroms=None, chroms_prefix="", namemap=None):
        self._graph_genome = graph_genome
        self._graph_genome_dir = graph_genome_dir
        self._bedfile = bedfile
        self._chroms = list(chroms) if chroms else [ALL_CHROMS]
        self._chroms_prefix = chroms_prefix
        self._namemap = dict(namemap) if namemap else {}
        self._cores = int(cores)
        self._thresh = float(threshold)
        self._no_qvalue = bool(no_qvalue)
        self._qvalueT = bool(qval_t)
        self._no_rev = bool(no_reverse)
        self._recomb = bool(recomb)
        self._verbose = bool(verbose)
        self._bgfile = bgfile
        self._pseudo = float(pseudo)

    cores = property(lambda self: self._cores)
    threshold = property(lambda self: self._thresh)
    noqvalue = property(lambda self: self._no_qvalue)
    qvalueT = property(lambda self: self._qvalueT)
    noreverse = property(lambda self: self._no_rev)
    recomb = property(lambda self: self._recomb)
    verbose = property(lambda self: self._verbose)
    bgfile = property(lambda self: self._bgfile)
    pseudo = property(lambda self: self._pseudo)
    graph_genome = property(lambda self: self._graph_genome)
    graph_genome_dir = property(lambda self: self._graph_genome_dir)
    bedfile = property(lambda self: self._bedfile)
    chroms = property(lambda self: self._chroms)
    chroms_num = property(lambda self: len(self._chroms))
    chroms_prefix = property(lambda self: self._chroms_prefix)
    namemap = property(lambda self: self._namemap)

    def has_graphgenome(self) -> bool:
        return bool(self._graph_genome)

    def has_graphgenome_dir(self) -> bool:
        return bool(self._graph_genome_dir)


REQUIRED_FLAGS = ("cores", "threshold", "noqvalue", "qvalueT", "noreverse", "recomb", "verbose")


def is_findmotif_like(obj) -> bool:
    return all(hasattr(obj, k) for k in REQUIRED_FLAGS)


SCAN_FLAGS = ("bedfile", "chroms", "chroms_num", "chroms_prefix", "namemap", "cores", "verbose",
              "has_graphgenome", "has_graphgenome_dir")


def is_scan_args_like(obj) -> bool:
    return all(hasattr(obj, k) for k in SCAN_FLAGS)
