"""TEST INFRASTRUCTURE (imports oracle/): one seed of the report-table fuzz -- the product's top-level calls over a graph,
compute_results_from_graph_many (the widths out of step, the native columns on the library's host threads, frames on the
calling thread) and compute_results_from_graph (one motif, everything inline), on a random rich graph (SNPs, multi-allelic
sites, insertions, deletions, multi-base substitutions), random regions, a random motif set (widths 1..40, several motifs of
one width, sometimes the same motif twice) and random flags, against the CPU oracle end to end (tests/extract_helpers.py
oracle_table: the walk enumerator's rows as TSV files -> oracle.compute_results) -- table for table, every column.
`pytest -m gpu` runs a bounded seed set (tests/test_gpu_fused.py); scripts/tables_fuzz.py runs seeds for a fixed time."""
import contextlib
import io
import os
import shutil

import numpy as np

from extract_fuzz_core import SynMotif
from extract_helpers import assert_table_equals_oracle, make_graph_files, oracle_table, variants_from_index


def _same(a, b, what):
    assert list(a.columns) == list(b.columns), what
    assert len(a) == len(b), (what, len(a), len(b))
    for c in b.columns:
        if b[c].dtype.kind == "f":
            assert np.array_equal(a[c].to_numpy(float), b[c].to_numpy(float)), (what, c)
        else:
            assert (a[c].astype(str).to_numpy() == b[c].astype(str).to_numpy()).all(), (what, c)


def _approx_rows(idx, regions, W):
    """upper estimate of the rows of the regions at width W (allele product per window, two ways per indel, two strands)"""
    tot = 0.0
    for s, e in regions:
        for p in range(s, max(s, e - W + 1) + 1):
            i0, i1 = np.searchsorted(idx.pos, p), np.searchsorted(idx.pos, p + W)
            w = 2.0
            for i in range(i0, i1):
                w *= (1 + int(idx.n_alts[i])) if (idx.del_len[i] == 0 and idx.ins_len[i] == 0) else 2
            tot += w
    return tot


def fuzz_seed(seed, tmp, stats):
    from grafimo_amd.extract_regions import (DeviceGraph, GraphIndex, compute_results_from_graph, compute_results_from_graph_many)
    from grafimo_amd.workflow import Findmotif
    rng = np.random.default_rng(90_000 + seed)
    d = os.path.join(str(tmp), f"s{seed}")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    length = int(rng.integers(700, 2200))
    fasta, vcf = make_graph_files(d, chrom="7", length=length, n_sites=int(length * rng.uniform(0.04, 0.12)),
                                  n_samples=int(rng.integers(4, 40)), seed=seed, rich=True)
    idx = GraphIndex.from_fasta_vcf(fasta, vcf, "7")
    g = DeviceGraph(idx)
    try:
        ref, v = idx.ref.tobytes(), variants_from_index(idx)
        regions = []
        for _ in range(int(rng.integers(1, 5))):
            s = int(rng.integers(0, length - 60))
            regions.append((s, min(length - 1, s + int(rng.integers(45, 420)))))
        widths = [int(w) for w in rng.choice(np.arange(1, 41), size=int(rng.integers(1, 4)), replace=False)]
        widths = [w for w in widths if any(e - s + 1 >= w for s, e in regions)] or [8]
        # the enumerator is a Python loop and a threshold of 1 reports every row: keep a seed's rows in the tens of thousands
        while max(_approx_rows(idx, regions, w) for w in widths) > 60_000:
            if len(regions) > 1:
                regions.pop(int(np.argmax([e - s for s, e in regions])))
            else:
                s, e = regions[0]
                regions[0] = (s, s + max(45, (e - s) // 2))
                if e - s <= 46:
                    widths = [min(widths)] if min(widths) < max(widths) else [max(1, widths[0] // 2)]
        if rng.random() < 0.3:
            regions.append((5, 5 + int(rng.integers(1, 8))))                   # a region shorter than most widths: no window
        motifs = [SynMotif(int(rng.choice(widths)), seed=int(rng.integers(0, 1 << 20))) for _ in range(int(rng.integers(2, 8)))]
        for i, m in enumerate(motifs):
            m.motif_id, m.motif_name = f"M{i}_{m.width}", f"m{i}"
        if rng.random() < 0.3:
            motifs.append(motifs[0])                                            # the same numbers twice in one set
        kw = dict(threshold=float(rng.choice([1.0, 0.5, 0.2, 0.05])), recomb=bool(rng.integers(0, 2)),
                  no_reverse=bool(rng.random() < 0.25))
        r = rng.random()
        if r < 0.25:
            kw["qval_t"] = True
            kw["threshold"] = max(kw["threshold"], 0.3)
        elif r < 0.4:
            kw["no_qvalue"] = True
        wf = Findmotif(**kw)
        with contextlib.redirect_stdout(io.StringIO()) as out:
            tabs = compute_results_from_graph_many(motifs, g, regions, True, wf)
            singles = [compute_results_from_graph(m, g, regions, True, wf) for m in motifs]
        assert out.getvalue().count("Scanned sequences:") == 2 * len(motifs)
        done = set()
        for i, (m, many, one) in enumerate(zip(motifs, tabs, singles)):
            what = (seed, i, m.width, kw)
            _same(many, one, what)
            exp, scanned = oracle_table(os.path.join(d, "oracle"), "7", ref, v, regions, m, reuse_rows=m.width in done, **kw)
            done = {m.width}                       # (the directory holds the rows of ONE width at a time)
            assert_table_equals_oracle(many, exp, what)
            stats["rows_scanned"] += scanned
            stats["rows_reported"] += len(many)
            stats["tables"] += 1
        stats["graphs"] += 1
    finally:
        g.close()
        shutil.rmtree(d, ignore_errors=True)
