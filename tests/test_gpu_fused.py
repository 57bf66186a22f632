"""Extraction fused into scoring (gfm_graph_score + gfm_graph_annotate: csrc/gfm_graph_fused.hpp) against the
materialising path (gfm_graph_plan + gfm_graph_emit + the score kernel over the rows) -- the same table, row for row and
column for column -- and against the CPU oracle.  What the fused path replaces in the reference: the TSV between
`vg find -K` (extract_regions.py:180,225) and score_seqs (score_sequences.py:273-321)."""
import contextlib
import io
import os

import numpy as np
import pandas as pd
import pytest

from conftest import REF_DATA
from extract_helpers import make_graph_files

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _ctcf():
    from grafimo_amd.motif_ops import build_motif_meme_host
    return build_motif_meme_host(os.path.join(REF_DATA, "MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]


class _SynMotif:
    """the members the scoring path reads (grafimo_amd.motif.MOTIF_FIELDS); no score distribution: the DP runs on the device"""

    def __init__(self, W, seed=0):
        from grafimo_amd import synth
        rec = synth.synthetic_motif(W, np.random.default_rng(1000 + 7 * W + seed), np.array([0.3, 0.2, 0.2, 0.3]))
        self.score_matrix, self.nucsmap = rec["sm"], {n: i for i, n in enumerate("ACGT")}
        self.bg = {n: float(rec["bg"][i]) for i, n in enumerate("ACGT")}
        self.min_val, self.scale, self.offset, self.width = int(rec["min_val"]), int(rec["scale"]), np.double(rec["offset"]), W
        self.motif_id, self.motif_name = f"SYN{W}", f"syn{W}"


def _motif_of_width(W, seed=0):
    return _SynMotif(W, seed)


def _approx_walks(idx, regions, W):
    """upper estimate of the walks of a plan on the host (allele product per window, two ways per indel): the tests
    check it BEFORE a GPU call -- a threshold of 1 reports every row, and rows are host memory"""
    tot = 0.0
    for s, e in regions:
        for p in range(s, max(s, e - 1) + 1):
            i0, i1 = np.searchsorted(idx.pos, p), np.searchsorted(idx.pos, p + W)
            w = 1.0
            for i in range(i0, i1):
                w *= (1 + int(idx.n_alts[i])) if (idx.del_len[i] == 0 and idx.ins_len[i] == 0) else 2
            tot += w
    return tot


def _both(motif, g, regions, **kw):
    from grafimo_amd.extract_regions import compute_results_from_graph
    from grafimo_amd.workflow import Findmotif
    with contextlib.redirect_stdout(io.StringIO()) as o1:
        a = compute_results_from_graph(motif, g, regions, True, Findmotif(**kw))
    with contextlib.redirect_stdout(io.StringIO()) as o2:
        b = compute_results_from_graph(motif, g, regions, True, Findmotif(**kw), fused=False)
    assert o1.getvalue() == o2.getvalue()           # the same "Scanned sequences" / "Scanned nucleotides" lines
    return a, b


def _assert_same(a: pd.DataFrame, b: pd.DataFrame, what=""):
    assert list(a.columns) == list(b.columns), what
    assert len(a) == len(b), (what, len(a), len(b))
    for c in b.columns:
        if b[c].dtype.kind == "f":
            assert np.array_equal(a[c].to_numpy(float), b[c].to_numpy(float)), (what, c)
        else:
            assert (a[c].astype(str).to_numpy() == b[c].astype(str).to_numpy()).all(), (what, c)


@pytest.mark.parametrize("W", [1, 3, 8, 19, 30, 47, 64])
def test_every_row_of_a_rich_graph(tmp_path, W):
    """threshold 1.0 with --recomb reports EVERY row: the fused path's k-mers, coordinates, strands, haplotype counts
    and ref flags for all of them equal the materialised rows' -- SNPs, multi-allelic sites, insertions, deletions,
    multi-base substitutions, complex alleles, several records at one position; regions that start at 0, end at the
    chromosome's end, hold more than one tile of windows, or hold no window at all."""
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex
    n_sites = {30: 200, 47: 120, 64: 90}.get(W, 300)          # wide windows over dense sites are millions of walks
    fasta, vcf = make_graph_files(str(tmp_path), chrom="7", length=2600, n_sites=n_sites, n_samples=40, seed=40 + W, rich=True)
    idx = GraphIndex.from_fasta_vcf(fasta, vcf, "7")
    regions = [(0, 130), (400, 400 + W - 1), (500, 1300), (1290, 1500), (2300, 2600), (2590, 2600)]
    assert _approx_walks(idx, regions, W) < 1e6
    g = DeviceGraph(idx)
    motif = _ctcf() if W == 19 else _motif_of_width(W)
    for kw in (dict(threshold=1.0, recomb=True), dict(threshold=1.0, recomb=True, no_reverse=True)):
        a, b = _both(motif, g, regions, **kw)
        assert len(b) > 1000
        _assert_same(a, b, (W, kw))
    g.close()


def test_flag_settings_on_thresholds_that_select(tmp_path):
    """p- and q-value thresholds, --no-qvalue, --no-reverse, the --recomb filter: fused == materialised == the oracle's
    compute_results over the oracle's own rows."""
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex
    fasta, vcf = make_graph_files(str(tmp_path), chrom="7", length=6000, n_sites=500, seed=78, rich=True)
    g = DeviceGraph(GraphIndex.from_fasta_vcf(fasta, vcf, "7"))
    regions = np.array([(0, 900), (1200, 2500), (3000, 5990)], dtype=np.int64)      # an [n, 2] array is accepted too
    assert _approx_walks(g.index, regions.tolist(), 19) < 2e6
    motif = _ctcf()
    for kw in [dict(threshold=0.05), dict(threshold=0.5, qval_t=True, recomb=True), dict(threshold=0.3, qval_t=True),
               dict(threshold=0.02, no_reverse=True), dict(threshold=0.05, no_qvalue=True, recomb=True),
               dict(threshold=1e-4)]:
        a, b = _both(motif, g, regions, **kw)
        _assert_same(a, b, kw)
        if kw["threshold"] > 1e-3 and not kw.get("qval_t"):
            assert len(a) > 0, kw
    g.close()


def test_tiles_with_more_sites_than_the_lds_stage_and_heavy_windows():
    """A hand-made chromosome with a SNP at EVERY position and, every eight bases over a stretch, twenty one-base
    insertions behind one anchor: a tile's site table (262 SNPs + 240 insertions) overflows the 448 records staged in
    LDS and its later windows read their sites from global memory; every window of W = 10 over SNPs alone holds 1 024
    walks (several phase-2 rounds per tile), a window over an anchor twenty-one layouts.  Fused == materialised on
    every row."""
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex
    rng = np.random.default_rng(9)
    Lr = 700
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = acgt[rng.integers(0, 4, Lr)]
    pos, n_alts, alt, ins_len, ins_off, ins_bases = [], [], [], [], [], []
    anchors = {100 + 8 * k for k in range(12)}
    for x in range(20, 560):
        other = acgt[(np.searchsorted(acgt, ref[x]) + 1) % 4]
        pos.append(x); n_alts.append(1); alt.append([other, 0, 0]); ins_len.append(0); ins_off.append(0)
        if x in anchors:
            for _ in range(20):
                pos.append(x); n_alts.append(1); alt.append([0, 0, 0]); ins_len.append(1)
                ins_off.append(len(ins_bases)); ins_bases.append(int(acgt[rng.integers(0, 4)]))
    H = 70
    bits = rng.integers(0, 2 ** 63, size=(len(pos), 3, 2), dtype=np.uint64)
    bits[:, 1:, :] = 0
    bits[:, :, 1] &= np.uint64((1 << (H - 64)) - 1)
    idx = GraphIndex("c", ref, np.array(pos, np.int32), np.array(n_alts, np.uint8), np.array(alt, np.uint8), bits, H,
                     ins_len=np.array(ins_len, np.int32), ins_off=np.array(ins_off, np.int32),
                     ins_bases=np.array(ins_bases, np.uint8))
    g = DeviceGraph(idx)
    a, b = _both(_motif_of_width(6), g, [(99, 420)], threshold=1.0, recomb=True)
    assert 100_000 < len(b) < 3_000_000
    _assert_same(a, b, "dense tile")
    a, b = _both(_motif_of_width(10), g, [(300, 460), (0, 90)], threshold=1.0, recomb=True)
    assert 250_000 < len(b) < 3_000_000
    _assert_same(a, b, "heavy windows")
    a, b = _both(_motif_of_width(6), g, [(60, 460)], threshold=0.05)
    _assert_same(a, b, "selected")
    g.close()


def test_heavy_windows_go_to_their_own_kernel_and_the_list_is_kept_with_the_plan():
    """Windows of more than 64 walks leave graph_score_kernel for graph_heavy_kernel (plain windows: their walks shared out
    over the grid) or for the deletion kernels (a one-deletion window of that many walks).  A stretch of twelve neighbouring
    sites, some tri- and tetra-allelic, a two-base deletion in its middle and quiet stretches around it: every row equals the
    materialised one -- on the first call of a plan (the heavy list is made), on the next call with ANOTHER motif of the
    width (the list is reused), after the regions changed (made again), forward strand only, and with a threshold that
    selects."""
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex
    rng = np.random.default_rng(17)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = acgt[rng.integers(0, 4, 900)]
    pos, n_alts, alt, del_len = [], [], [], []
    for x in list(range(200, 212)) + [320, 322, 324, 326, 328, 330, 332, 500, 640]:
        others = [c for c in acgt if c != ref[x]]
        k = 1 + (x % 3 if x < 212 else 0)
        pos.append(x); n_alts.append(k); alt.append(others[:k] + [0] * (3 - k)); del_len.append(0)
    pos.append(327); n_alts.append(1); alt.append([0, 0, 0]); del_len.append(2)         # a deletion inside the second stretch
    order = np.argsort(np.array(pos), kind="stable")
    H = 20
    bits = rng.integers(0, 2 ** 20, size=(len(pos), 3, 1), dtype=np.uint64)
    idx = GraphIndex("c", ref, np.array(pos, np.int32)[order], np.array(n_alts, np.uint8)[order], np.array(alt, np.uint8)[order],
                     bits[order], H, del_len=np.array(del_len, np.int32)[order])
    g = DeviceGraph(idx)
    regions = [(150, 400), (480, 700)]
    assert _approx_walks(idx, regions, 16) < 3e6
    a, b = _both(_motif_of_width(16), g, regions, threshold=1.0, recomb=True)
    assert len(b) > 100_000
    _assert_same(a, b, "first call of the plan")
    a, b = _both(_motif_of_width(16, seed=3), g, regions, threshold=1.0, recomb=True)
    _assert_same(a, b, "another motif, the plan's lists reused")
    a, b = _both(_motif_of_width(16), g, [(190, 260)], threshold=1.0, recomb=True, no_reverse=True)
    _assert_same(a, b, "other regions, forward only")
    a, b = _both(_motif_of_width(16), g, regions, threshold=0.01)
    assert 0 < len(a) < 100_000
    _assert_same(a, b, "a threshold that selects")
    g.close()


def test_a_hit_list_beyond_the_host_limit_is_refused(monkeypatch):
    """A threshold of 1 over windows of many variant sites would bring every allele combination back as a report row: the
    call is refused from the device's COUNT (nothing of that size is allocated or copied), naming the limit."""
    from grafimo_amd import _native as nv
    from grafimo_amd import extract_regions as xr
    from grafimo_amd.workflow import Findmotif
    rng = np.random.default_rng(3)
    ref = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 400)]
    pos = np.arange(100, 118, dtype=np.int32)                       # 18 biallelic sites inside one 19-mer: 2^18 walks
    alt = np.zeros((len(pos), 3), np.uint8)
    alt[:, 0] = np.where(ref[pos] == ord("A"), ord("C"), ord("A"))
    g = xr.DeviceGraph(xr.GraphIndex("c", ref, pos, np.ones(len(pos), np.uint8), alt, None, 0))
    monkeypatch.setattr(xr, "MAX_HITS", 100_000)
    with contextlib.redirect_stdout(io.StringIO()):
        with pytest.raises(nv.NativeError) as e:
            xr.compute_results_from_graph(_ctcf(), g, [(90, 130)], True, Findmotif(threshold=1.0, recomb=True))
        assert e.value.code == nv.GFM_ERR_OVERFLOW and "GRAFIMO_MAX_HITS" in str(e.value)
        df = xr.compute_results_from_graph(_ctcf(), g, [(90, 130)], True, Findmotif(threshold=1e-3))
    assert len(df) < 100_000
    g.close()


def test_config2_scale_graph_and_repeated_calls():
    """The bench's synthetic chromosome (1000-Genomes-like site density, deletions, 5 096 haplotypes) at a tenth of its
    size: fused == materialised at p < 1e-2; repeated calls (the tile table is reused while regions and width repeat,
    rebuilt when they change) and a hit list that starts too short (grown from the device's count) give the same table."""
    from grafimo_amd import synth
    from grafimo_amd.extract_regions import DeviceGraph
    idx, regions = synth.make_graph_index(1000, 19)
    g = DeviceGraph(idx)
    motif = _ctcf()
    a, b = _both(motif, g, regions, threshold=1e-2)
    assert len(a) > 500
    _assert_same(a, b)
    a2, _ = _both(motif, g, regions, threshold=1e-2)
    _assert_same(a2, a)
    sub = regions[100:300]
    a3, b3 = _both(motif, g, sub, threshold=1e-2)                   # other regions: new tiles
    _assert_same(a3, b3)
    assert len(a3) < len(a)
    g._fused_buf, g._fused_cap = None, 0
    g.fused_buffers(64)                                             # 64 entries: far too few -> counted, grown, redone
    a4, _ = _both(motif, g, regions, threshold=1e-2, qval_t=False)
    _assert_same(a4, a)
    assert g._fused_cap > 64
    m8 = _motif_of_width(8)
    a5, b5 = _both(m8, g, regions, threshold=1e-3)                  # another width over the same regions
    _assert_same(a5, b5)
    # one graph listed several times (chromosome entries of one call may share a handle), entries without regions
    parts = [regions[:300], regions[300:300], regions[300:]]
    a6, b6 = _both(motif, [g, g, g], parts, threshold=1e-2)
    _assert_same(a6, b6)
    _assert_same(a6, a)
    g.close()
