"""Debug aid for the fused extraction -> scoring path: fused vs materialised table on a small rich graph, printing the first
rows that differ (run on the GPU box)."""
import contextlib, io, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, pandas as pd, torch
from extract_helpers import make_graph_files
from grafimo_amd.extract_regions import DeviceGraph, GraphIndex, compute_results_from_graph
from grafimo_amd.motif_ops import build_motif_meme_host
from grafimo_amd.workflow import Findmotif

pd.set_option("display.width", 250); pd.set_option("display.max_columns", 30)
tmp = tempfile.mkdtemp()
motif = build_motif_meme_host(os.path.join(ROOT, "tests/golden/ref_data/MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
for rich in (False, True):
    fasta, vcf = make_graph_files(tmp, chrom="7", length=1200, n_sites=120, n_samples=30, seed=3, rich=rich)
    with contextlib.redirect_stderr(io.StringIO()):
        g = DeviceGraph(GraphIndex.from_fasta_vcf(fasta, vcf, "7"))
    for kw in (dict(threshold=1.0, recomb=True), dict(threshold=0.05)):
        with contextlib.redirect_stdout(io.StringIO()):
            a = compute_results_from_graph(motif, g, [(0, 500), (600, 1200)], True, Findmotif(**kw))
            b = compute_results_from_graph(motif, g, [(0, 500), (600, 1200)], True, Findmotif(**kw), fused=False)
        key = ["sequence_name", "start", "stop", "strand", "matched_sequence"]
        m = a.merge(b, on=key, how="outer", suffixes=("_f", "_m"), indicator=True)
        bad = m[(m["_merge"] != "both") | (m["haplotype_frequency_f"] != m["haplotype_frequency_m"]) |
                (m["reference_f"] != m["reference_m"]) | (m["p-value_f"] != m["p-value_m"])]
        print(f"rich={rich} {kw}: fused {len(a)} rows, materialised {len(b)} rows, differing {len(bad)}; order equal: "
              f"{len(a) == len(b) and (a[key].astype(str).values == b[key].astype(str).values).all()}")
        if len(bad):
            print(bad.head(20).to_string())
    g.close()
