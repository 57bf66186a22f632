"""gfm_region_best / gfm_locus_max / gfm_region_ids (csrc/region_reduce.hip, through the C ABI) against numpy group-bys
over ORACLE scores -- the strand-max / per-region best-hit reductions BASELINE.json's north_star names.  The reference
holds no such reduction (score_sequences.py:279-321 keeps every strand as a row); what it consumes is the order of the
regions by their best reported hit (--top-graphs, res_writer.py:153-157), which these must reproduce."""
import glob
import os

import numpy as np
import pytest

from conftest import REF_DATA

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ctcf(golden_motifs):
    _, flat = golden_motifs
    return flat["ctcf_meme_unif#0"]


def _oracle_scores(m, kmers):
    from oracle import oracle as orc
    sc, _ = orc.score_kmers_table(kmers, m["score_matrix"], orc.p_table(m["pmf"]), m["min_val"])
    return sc.astype(np.int32)


def _np_region_best(scores, region, n_regions, row_base=0, keep=None):
    """(best score, its lowest row) per region by sorting: region, then score descending, then row ascending."""
    idx = np.arange(len(scores)) if keep is None else np.nonzero(keep)[0]
    idx = idx[(region[idx] >= 0) & (region[idx] < n_regions)]
    best_s = np.full(n_regions, -1, dtype=np.int64)
    best_r = np.full(n_regions, -1, dtype=np.int64)
    if len(idx):
        order = idx[np.lexsort((idx, -scores[idx].astype(np.int64), region[idx]))]
        first = np.concatenate(([True], region[order][1:] != region[order][:-1]))
        best_s[region[order[first]]] = scores[order[first]]
        best_r[region[order[first]]] = order[first] + row_base
    return best_s, best_r


def _np_locus_max(scores, region, start, stop, keep=None):
    idx = np.arange(len(scores)) if keep is None else np.nonzero(keep)[0]
    out = np.full(len(scores), -1, dtype=np.int64)
    if not len(idx):
        return out
    lo, hi = np.minimum(start[idx], stop[idx]), np.maximum(start[idx], stop[idx])
    order = np.lexsort((hi, lo, region[idx]))
    r_, l_, h_ = region[idx][order], lo[order], hi[order]
    head = np.concatenate(([True], (r_[1:] != r_[:-1]) | (l_[1:] != l_[:-1]) | (h_[1:] != h_[:-1])))
    starts = np.nonzero(head)[0]
    gmax = np.maximum.reduceat(scores[idx][order].astype(np.int64), starts)
    out[idx[order]] = np.repeat(gmax, np.diff(np.concatenate((starts, [len(order)]))))
    return out


def _device_scores(m, kmers, dev):
    from grafimo_amd.device import DeviceMotif
    dm = DeviceMotif(m["score_matrix"], m["bg"], m["min_val"], m["scale"], m["offset"], m["pmf"])
    d_k = torch.from_numpy(kmers).to(dev)
    sc = torch.empty(len(kmers), dtype=torch.int32, device=dev)
    dm.score(d_k, sc)
    torch.cuda.synchronize(dev)
    return dm, sc


def test_config2_batch_ten_thousand_regions(dev, ctcf):
    """BASELINE config 2: 10 000 regions x 2 000 rows.  Per region the best (score, row) and per locus the
    both-strand maximum from the HIP path == numpy group-bys over the C oracle's scores; with and without the
    haplotype-frequency filter and a score cutoff (host value and device-resident)."""
    from grafimo_amd import synth, top_hits as th
    batch = synth.make_batch(10_000, 2_000, 19, ctcf["probs"], synth.seed_for(2))
    exp = _oracle_scores(ctcf, batch.kmers)
    dm, sc = _device_scores(ctcf, batch.kmers, dev)
    assert np.array_equal(sc.cpu().numpy(), exp)
    region = torch.from_numpy(batch.region).to(dev)
    freq = torch.from_numpy(batch.freq).to(dev)
    start, stop = torch.from_numpy(batch.start).to(dev), torch.from_numpy(batch.stop).to(dev)
    cut = dm.pvalue_cutoff(1e-4)
    d_cut = torch.tensor([cut], dtype=torch.int32, device=dev)
    for use_freq, min_score, d_c in [(False, 0, None), (True, 0, None), (False, cut, None), (True, 0, d_cut)]:
        eff = max(min_score, cut if d_c is not None else 0)
        keep = (exp >= eff) & ((batch.freq > 0) if use_freq else True)
        s, r, ok = th.decode_best(th.region_best(sc, region, 10_000, freq=freq if use_freq else None,
                                                 min_score=min_score, cutoff=d_c, row_base=5))
        es, er = _np_region_best(exp, batch.region, 10_000, row_base=5, keep=keep)
        assert np.array_equal(s, es) and np.array_equal(r, er) and np.array_equal(ok, es >= 0)
        lm = th.locus_max(sc, region, 10_000, start, stop, freq=freq if use_freq else None, min_score=min_score,
                          cutoff=d_c).cpu().numpy()
        assert np.array_equal(lm, _np_locus_max(exp, batch.region, batch.start, batch.stop, keep=keep))
    # the region order --top-graphs walks: regions by their best hit == first rows per region of the sorted hit table
    s, r, ok = th.decode_best(th.region_best(sc, region, 10_000, freq=freq, min_score=cut))
    hits = np.nonzero((exp >= cut) & (batch.freq > 0))[0]
    sel = th.best_rows_per_region(batch.region[hits], exp[hits], hits)
    assert np.array_equal(np.sort(hits[sel]), np.sort(r[ok]))
    assert np.array_equal(th.locus_max_of_hits(batch.region[hits], batch.start[hits], batch.stop[hits], exp[hits]),
                          th.locus_max(sc, region, 10_000, start, stop, freq=freq, min_score=cut).cpu().numpy()[hits])
    dm.close()


def test_reference_fixture_704_rows(dev, ctcf):
    """The reference's own scoring fixture (real `vg find` rows: '+' and '-' rows of one region, SNP alleles, a
    deletion): HIP scores -> region best and strand-max per locus == group-bys over oracle scores; the best row is
    the first row of the reference's expected table sorted by p-value."""
    import pandas as pd
    from grafimo_amd import top_hits as th
    from grafimo_amd.score_sequences import KmerTable
    files = sorted(glob.glob(os.path.join(REF_DATA, "width_19", "*.tsv")))
    t = KmerTable(files, 19, False, 1)
    assert t.n == 704
    exp = _oracle_scores(ctcf, t.kmers)
    dm, sc = _device_scores(ctcf, t.kmers, dev)
    assert np.array_equal(sc.cpu().numpy(), exp)
    region = torch.from_numpy(t.name_id).to(dev)
    n_reg = len(t.names)
    s, r, ok = th.decode_best(th.region_best(sc, region, n_reg))
    es, er = _np_region_best(exp, t.name_id, n_reg)
    assert np.array_equal(s, es) and np.array_equal(r, er) and ok.all()
    ref = pd.read_csv(os.path.join(REF_DATA, "scoring_results.tsv"), sep="\t", index_col=0)
    best = int(r[0])
    assert float(ref["score"].max()) == pytest.approx(exp[best] / ctcf["scale"] + 19 * ctcf["offset"], rel=1e-12)
    top = ref[ref["score"] == ref["score"].max()]
    assert t.kmers[best].tobytes().decode() in set(top["matched_sequence"])
    lm = th.locus_max(sc, region, n_reg, torch.from_numpy(t.start).to(dev), torch.from_numpy(t.stop).to(dev)).cpu().numpy()
    want = _np_locus_max(exp, t.name_id, t.start, t.stop)
    assert np.array_equal(lm, want)
    # a '+' row and the '-' row of the same span see the same maximum
    fwd = np.nonzero(t.strand == ord("+"))[0]
    key = {(int(t.start[i]), int(t.stop[i])): lm[i] for i in fwd}
    for i in np.nonzero(t.strand == ord("-"))[0]:
        k = (int(t.stop[i]), int(t.start[i]))
        if k in key:
            assert key[k] == lm[i]
    dm.close()


def test_any_row_order_ragged_sizes_and_accumulation(dev):
    """Region ids in any order (the segmented wave scan is an optimisation for contiguous regions, not a
    requirement), ids outside the table ignored, sizes around the 64-lane strips, keys accumulating over two
    batches through `out`; gfm_region_ids with empty regions."""
    from grafimo_amd import top_hits as th
    rng = np.random.default_rng(11)
    for n in (1, 63, 64, 65, 1000, 100_003):
        n_reg = 37
        scores = rng.integers(0, 64_001, n).astype(np.int32)
        region = rng.integers(-2, n_reg + 2, n).astype(np.int32)
        if n > 500:
            region[100:400] = 5                                  # a long run inside the noise
            scores[150:160] = 64_000                             # equal maxima: the lowest row wins
        start = rng.integers(0, 50, n).astype(np.int64) + 1000 * np.maximum(region, 0)
        stop = start + rng.choice([19, -19, 18], n)
        d = lambda a: torch.from_numpy(a).to(dev)
        s, r, ok = th.decode_best(th.region_best(d(scores), d(region), n_reg))
        es, er = _np_region_best(scores, region, n_reg)
        assert np.array_equal(s, es) and np.array_equal(r, er), n
        valid = (region >= 0) & (region < n_reg)
        lm = th.locus_max(d(scores), d(region), n_reg, d(start), d(stop)).cpu().numpy()
        assert np.array_equal(lm, _np_locus_max(scores, region, start, stop, keep=valid)), n
        # two batches into one table
        h = n // 2
        out = th.region_best(d(scores[:h]), d(region[:h]), n_reg)
        out = th.region_best(d(scores[h:]), d(region[h:]), n_reg, row_base=h, out=out)
        s2, r2, _ = th.decode_best(out)
        assert np.array_equal(s2, es) and np.array_equal(r2, er), n
    off = np.array([0, 0, 10, 10, 10, 300, 1000, 1000], dtype=np.int64)
    ids = th.region_ids(off, 1000, device=dev).cpu().numpy()
    assert np.array_equal(ids, np.searchsorted(off, np.arange(1000), side="right") - 1)
    assert len(th.region_best(torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.int32, device=dev), 3)) == 3


def test_scanner_keeps_the_best_hit_per_region_and_gathers_only_that(ctcf):
    """KmerScanner.set_regions: every step leaves per region the key of its best reported row (threshold on p and on q,
    haplotype-frequency filter) == a numpy group-by over oracle scores; with a process group (RCCL, one rank here) and
    top_only the per-step gather moves the n_regions keys instead of the hit entries."""
    import torch.distributed as dist
    from grafimo_amd import synth, top_hits as th
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.scan import KmerScanner
    from oracle import oracle as orc
    dev = torch.device("cuda:0")
    n_reg = 400
    batch = synth.make_batch(n_reg, 600, 19, ctcf["probs"], synth.seed_for(21))
    exp = _oracle_scores(ctcf, batch.kmers)
    ptab = orc.p_table(ctcf["pmf"])
    q = orc.fdr_bh(ptab[exp])
    dm = DeviceMotif(ctcf["score_matrix"], ctcf["bg"], ctcf["min_val"], ctcf["scale"], ctcf["offset"], ctcf["pmf"])
    d_k, region = torch.from_numpy(batch.kmers).to(dev), torch.from_numpy(batch.region).to(dev)
    freq = torch.from_numpy(batch.freq).to(dev)
    started = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29588")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        started = True
    try:
        for on_q, thr, use_freq, coll in [(False, 1e-3, False, False), (False, 1e-3, True, True), (True, 0.5, True, True)]:
            sc = KmerScanner(dm, len(batch), device=dev, always_collective=coll)
            sc.set_regions(region, n_reg, freq=freq if use_freq else None, top_only=coll)
            slot = sc.enqueue(d_k, thr, on_qvalue=on_q, want_qvalues=True, gather_hits=coll)
            res = sc.collect(slot)
            keep = ((q < thr) if on_q else (ptab[exp] < thr)) & ((batch.freq > 0) if use_freq else True)
            es, er = _np_region_best(exp, batch.region, n_reg, keep=keep)
            s_, r_, ok = th.decode_best(res["best"])
            assert np.array_equal(s_, es) and np.array_equal(r_, er), (on_q, thr, use_freq, coll)
            assert ok.sum() > 5
            if coll:
                assert "rows" not in res and slot.gathered is not None and slot.gathered[0].numel() == n_reg
    finally:
        if started:
            dist.destroy_process_group()
        dm.close()


def test_graph_path_top_graphs_is_the_first_row_per_region_of_the_report(tmp_path):
    """compute_results_from_graph(top_graphs=N) == top_regions_table(full report, N), fused and materialising."""
    import contextlib
    import io
    from extract_helpers import make_graph_files
    from grafimo_amd import top_hits as th
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex, compute_results_from_graph
    from grafimo_amd.motif_ops import build_motif_meme_host
    from grafimo_amd.workflow import Findmotif
    fasta, vcf = make_graph_files(str(tmp_path), chrom="7", length=6000, n_sites=400, seed=31)
    with contextlib.redirect_stderr(io.StringIO()):
        g = DeviceGraph(GraphIndex.from_fasta_vcf(fasta, vcf, "7"))
    motif = build_motif_meme_host(os.path.join(REF_DATA, "MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
    regions = [(i, i + 300) for i in range(0, 5700, 300)]
    for kw in (dict(threshold=0.02), dict(threshold=0.02, recomb=True)):
        with contextlib.redirect_stdout(io.StringIO()):
            full = compute_results_from_graph(motif, g, regions, True, Findmotif(**kw))
            top = compute_results_from_graph(motif, g, regions, True, Findmotif(**kw), top_graphs=5)
            top_m = compute_results_from_graph(motif, g, regions, True, Findmotif(**kw), top_graphs=5, fused=False)
        want = th.top_regions_table(full, 5)
        assert len(want) == 5 and full["sequence_name"].nunique() > 5
        for c in full.columns:
            assert (top[c].astype(str) == want[c].astype(str)).all(), c
            assert (top_m[c].astype(str) == want[c].astype(str)).all(), c
        assert list(top["sequence_name"]) == th.top_regions(full, 5)
    g.close()
