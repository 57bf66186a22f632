"""Scoring stage of ``grafimo findmotif`` on the GPU, for k-mers that were already extracted.

The reference's ``findmotif`` runs ``get_motif_pwm`` -> ``scan_graph`` (external ``vg find``) ->
``compute_results`` -> ``write_results`` (grafimo.py:80-190).  ``vg`` is not part of this build, so
this entry point starts after ``scan_graph``: ``--sequences`` is the directory it would have
produced (``width_W/REGION.tsv``).  Flags keep the names, defaults and meaning of the reference CLI
(__main__.py:119-415): -m/--motif, -k/--bgfile, -p/--pseudo, -t/--threshold, -q/--no-qvalue,
-r/--no-reverse, -f/--text-only, --recomb, --qvalueT, -j/--cores, -o/--out, --verbose, --debug.

    python -m grafimo_amd -m MA0139.1.meme -s /tmp/grafimo_XXXX -t 1e-4 -o out_dir

Without vg, the k-mers can also come from the extraction kernel: give the inputs of ``grafimo buildvg``
(-l/--linear-genome FASTA, -v/--vcf phased VCF) and the -b/--bedfile of ``findmotif`` instead of -s;
substitutions, insertions and deletions of the VCF are part of that graph (grafimo_amd/extract_regions.py).

    python -m grafimo_amd -m MA0139.1.meme -l chr22.fa -v chr22.vcf.gz -b peaks.bed -o out_dir
"""
import argparse
import sys
import time

from .motif_ops import get_motif_pwm
from .res_writer import DEFAULT_OUTDIR, print_results, write_results
from .score_sequences import compute_results, compute_results_many
from .utils import UNIF
from .workflow import Findmotif


class _Workflow(Findmotif):
    def __init__(self, a):
        super().__init__(cores=a.cores, threshold=a.threshold, no_qvalue=a.no_qvalue, qval_t=a.qval_t,
                         no_reverse=a.no_reverse, recomb=a.recomb, verbose=a.verbose, bgfile=a.bgfile,
                         pseudo=a.pseudo)
        self.outdir = a.out
        self.top_graphs = 0
        self.text_only = a.text_only


def get_parser():
    p = argparse.ArgumentParser(prog="python -m grafimo_amd", description=__doc__,
                                formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("-m", "--motif", nargs="+", required=True, metavar="MOTIF-FILE")
    p.add_argument("-s", "--sequences", metavar="DIR",
                   help="directory holding width_W/*.tsv as written by vg find -K W -E")
    p.add_argument("-l", "--linear-genome", dest="linear_genome", metavar="FASTA",
                   help="reference FASTA (with -v and -b: extract the k-mers on the GPU instead of -s)")
    p.add_argument("-v", "--vcf", metavar="VCF", help="phased VCF (.vcf or .vcf.gz): substitutions, insertions, deletions")
    p.add_argument("-b", "--bedfile", metavar="BED", help="regions to scan (UCSC BED: lines starting with chr)")
    p.add_argument("--chroms-prefix-find", dest="chroms_prefix", default="", metavar="PREFIX",
                   help="chromosome names in the FASTA / VCF = PREFIX + the BED name without its leading chr")
    p.add_argument("--strict-variants", action="store_true", dest="strict_variants",
                   help="fail on VCF records with a symbolic ALT (<DEL>, <CN0>, breakends, '*') instead of leaving them "
                        "out with a warning, which is what vg construct does without --handle-sv")
    p.add_argument("--skip-unmodelled-variants", action="store_true", dest="skip_unmodelled",
                   help="accepted for compatibility (it is the default now: complex alleles are modelled, records with "
                        "symbolic ALTs are left out with a warning)")
    p.add_argument("-k", "--bgfile", default=UNIF)
    p.add_argument("-p", "--pseudo", type=float, default=0.1)
    p.add_argument("-t", "--threshold", type=float, default=1e-4)
    p.add_argument("-q", "--no-qvalue", action="store_true", dest="no_qvalue")
    p.add_argument("-r", "--no-reverse", action="store_true", dest="no_reverse")
    p.add_argument("-f", "--text-only", action="store_true", dest="text_only")
    p.add_argument("--recomb", action="store_true")
    p.add_argument("--qvalueT", action="store_true", dest="qval_t")
    p.add_argument("-j", "--cores", type=int, default=0, help="host threads for TSV ingest (0 = all)")
    p.add_argument("-o", "--out", default=DEFAULT_OUTDIR)
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--debug", action="store_true")
    return p


def main(argv=None):
    a = get_parser().parse_args(argv)
    if a.threshold <= 0 or a.threshold > 1:
        sys.exit("ERROR: the threshold must be in (0, 1]")
    if a.qval_t and a.no_qvalue:
        sys.exit("ERROR: --qvalueT needs q-values (drop -q)")
    from_graph = bool(a.linear_genome or a.vcf or a.bedfile)
    if from_graph == bool(a.sequences) or (from_graph and not (a.linear_genome and a.vcf and a.bedfile)):
        sys.exit("ERROR: give either -s DIR or all of -l FASTA -v VCF -b BED")
    if a.cores <= 0:
        import os
        a.cores = os.cpu_count() or 1
    wf = _Workflow(a)
    start = time.time()
    motifs = []
    for mfile in a.motif:
        # the Motif objects come back without their score distribution: the DP runs on the GPU
        # when compute_results uploads the motif
        motifs += get_motif_pwm(mfile, wf, a.cores, a.debug, pvalue_matrix=False)
    graphs, region_lists = [], []
    if from_graph:
        from .extract_regions import DeviceGraph, GraphIndex, compute_results_from_graph, read_bed_regions
        for bed_chrom, regs in read_bed_regions(a.bedfile, a.debug).items():
            chrom = a.chroms_prefix + bed_chrom.split("chr")[1]      # extract_regions.py:122,137
            index = GraphIndex.from_fasta_vcf(a.linear_genome, a.vcf, chrom, allow_skipped=not a.strict_variants)
            if a.verbose:
                print(f"{chrom}: {len(index.pos)} variant sites ({int((index.ins_len > 0).sum())} insertions, "
                      f"{int((index.del_len > 0).sum())} deletions), {index.n_haplotypes} haplotypes, "
                      f"{index.skipped} ALT alleles left out")
            graphs.append(DeviceGraph(index))
            region_lists.append(regs)
    shared = None if from_graph or len(motifs) < 2 else compute_results_many(motifs, a.sequences, a.debug, wf)
    for k, motif in enumerate(motifs):
        if from_graph:
            res = compute_results_from_graph(motif, graphs, region_lists, a.debug, wf)
        elif shared is not None:
            res = shared[k]              # one ingest / upload per width, batched launches
        else:
            res = compute_results(motif, a.sequences, a.debug, wf)
        if a.text_only:
            print_results(res, a.debug)
        else:
            write_results(res, motif, len(motifs), wf, a.debug)
    print("Elapsed time %.2fs" % (time.time() - start))


if __name__ == "__main__":
    main()
