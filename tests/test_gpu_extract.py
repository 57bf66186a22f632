"""K-mer extraction on the MI355X (gfm_graph_* through grafimo_amd.extract_regions) against the
oracle, the reference's golden file, and end to end into the scoring path."""
import contextlib
import io
import datetime
import os

import numpy as np
import pandas as pd
import pytest

from conftest import REF_DATA
from extract_helpers import make_graph_files, scoring_fixture_graph

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _oracle_rows(fasta, vcf, chrom, regions, W, counts=True):
    from oracle import extract_oracle as xo
    ref = xo.read_fasta(fasta)[chrom]
    sites, dels, _ = xo.read_vcf_graph(vcf, chrom)
    nodes = xo.GraphNodeTable(len(ref), sites, dels)
    rows = []
    for s, e in regions:
        rows += xo.enumerate_region_graph(chrom, ref, sites, dels, s, e, W, with_counts=counts, nodes=nodes)
    return rows


def test_reference_golden_file_through_the_gpu(tmp_path):
    """The reference's test_sequence_extraction (tests/grafimo_run_test.py:49-63) with the extraction
    kernel in vg's place: the written TSV equals expected_seqs.tsv after sorting, like the test does."""
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex, write_region_tsvs
    idx = GraphIndex.from_fasta_vcf(os.path.join(REF_DATA, "test.fa"), os.path.join(REF_DATA, "test.vcf.gz"), "x",
                                    with_haplotypes=False)            # the test runs vg without -H
    g = DeviceGraph(idx)
    rows = g.extract([(0, 20)], 19)
    assert len(rows) == 32
    paths = write_region_tsvs(idx, rows, str(tmp_path))
    assert paths == [os.path.join(str(tmp_path), "width_19", "x_0-20.tsv")]
    result = pd.read_csv(paths[0], sep="\t", header=None).sort_values([1, 2, 3])
    result.index = range(len(result))
    expected = pd.read_csv(os.path.join(REF_DATA, "expected_seqs.tsv"), sep="\t", header=None).sort_values([1, 2, 3])
    expected.index = range(len(expected))
    assert result.equals(expected)
    g.close()


@pytest.mark.parametrize("W", [1, 8, 19, 32, 64])
def test_rows_equal_the_oracle_on_a_synthetic_graph(tmp_path, W):
    """Every column of every row, in order: clustered multi-allelic sites, 130 haplotypes (ragged last
    bitset word), regions that touch both ends of the chromosome, overlap, or are shorter than W."""
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex, write_region_tsvs
    # (a 64-base window over the dense graph holds thousands of walks: keep the oracle's Python loop short)
    fasta, vcf = make_graph_files(str(tmp_path), chrom="7", seed=40 + W, n_sites=260 if W < 64 else 100)
    idx = GraphIndex.from_fasta_vcf(fasta, vcf, "7", allow_skipped=True)
    regions = [(0, 150), (100, 300), (2900, 3000), (1500, 1500 + W - 1), (1000, 1250), (2990, 3050), (-20, 90)]
    g = DeviceGraph(idx)
    rows = g.extract(regions, W)
    exp = _oracle_rows(fasta, vcf, "7", [(max(s, 0), min(e, 3000)) for s, e in regions], W)
    assert len(rows) == len(exp) > 0
    km = rows.kmers.cpu().numpy()
    assert [k.tobytes().decode() for k in km] == [r[1] for r in exp]
    strand = [chr(c) for c in rows.strand.cpu().numpy()]
    assert [f"7:{a}{s}" for a, s in zip(rows.start.cpu().numpy(), strand)] == [r[2] for r in exp]
    assert [f"7:{a}{s}" for a, s in zip(rows.stop.cpu().numpy(), strand)] == [r[3] for r in exp]
    assert rows.freq.cpu().numpy().tolist() == [r[4] for r in exp]
    assert ["ref" if x else "non.ref" for x in rows.is_ref.cpu().numpy()] == [r[5] for r in exp]
    assert max(r[4] for r in exp) == 130 and any(0 < r[4] < 130 for r in exp)
    span = lambda r: abs(int(r[3].split(":")[1][:-1]) - int(r[2].split(":")[1][:-1]))
    assert W == 1 or any(span(r) != W for r in exp)           # walks that jump a deletion are in the set
    # the TSV files: same text as the oracle's rows (labels carry the caller's region bounds)
    paths = write_region_tsvs(idx, rows, str(tmp_path / "out"))
    text = "".join(open(p).read() for p in paths)
    lab = {f"7:{max(s, 0)}-{min(e, 3000)}": f"7:{s}-{e}" for s, e in regions}
    exp_text = "".join("\t".join([lab[r[0]]] + [str(x) for x in r[1:]]) + "\n" for r in exp)
    assert sorted(text.splitlines()) == sorted(exp_text.splitlines())
    # the handle keeps its plan buffers: a smaller and then a larger plan on the same graph
    small = g.extract([(100, 300)], W)
    assert [k.tobytes().decode() for k in small.kmers.cpu().numpy()] == \
        [r[1] for r in _oracle_rows(fasta, vcf, "7", [(100, 300)], W)]
    again = g.extract(regions + [(200, 700)], W)
    exp2 = exp + _oracle_rows(fasta, vcf, "7", [(200, 700)], W)
    assert again.freq.cpu().numpy().tolist() == [r[4] for r in exp2]
    assert [k.tobytes().decode() for k in again.kmers.cpu().numpy()] == [r[1] for r in exp2]
    g.close()


@pytest.mark.parametrize("W", [3, 8, 19, 30])
def test_insertions_and_multibase_substitutions_equal_the_oracle(tmp_path, W):
    """Round 2: graphs with insertions (1..6 bases, several per anchor), multi-base substitutions and merged
    same-position records.  Semantics UNPINNED (no vg output shows them; oracle/extract_oracle.py states what is
    assumed): the kernels must produce exactly the oracle's rows -- k-mers, coordinates, strands, haplotype counts,
    flags, in order -- incl. walks that start or end inside inserted bases and windows beyond E - W."""
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex, write_region_tsvs
    from oracle import extract_oracle as xo
    fasta, vcf = make_graph_files(str(tmp_path), chrom="7", length=2500, n_sites=150, n_samples=65, seed=60 + W, rich=True)
    idx = GraphIndex.from_fasta_vcf(fasta, vcf, "7", allow_skipped=True)
    assert (idx.ins_len > 0).sum() > 5 and (idx.del_len > 0).sum() > 2
    ref = xo.read_fasta(fasta)["7"]
    v = xo.read_vcf_variants(vcf, "7")
    regions = [(0, 200), (150, 420), (2400, 2500), (1000, 1000 + W - 1), (1200, 1500), (2450, 2600), (-10, 60)]
    g = DeviceGraph(idx)
    rows = g.extract(regions, W)
    exp = []
    for s, e in regions:
        exp += xo.enumerate_region_variants("7", ref, v, max(s, 0), min(e, 2500), W, with_counts=True)
    assert len(rows) == len(exp) > 0
    km = rows.kmers.cpu().numpy()
    assert [k.tobytes().decode() for k in km] == [r[1] for r in exp]
    strand = [chr(c) for c in rows.strand.cpu().numpy()]
    assert [f"7:{a}{s}" for a, s in zip(rows.start.cpu().numpy(), strand)] == [r[2] for r in exp]
    assert [f"7:{a}{s}" for a, s in zip(rows.stop.cpu().numpy(), strand)] == [r[3] for r in exp]
    assert rows.freq.cpu().numpy().tolist() == [r[4] for r in exp]
    assert ["ref" if x else "non.ref" for x in rows.is_ref.cpu().numpy()] == [r[5] for r in exp]
    span = lambda r: int(r[3].split(":")[1][:-1]) - int(r[2].split(":")[1][:-1])
    fwd = [r for r in exp if r[2].endswith("+")]
    assert any(span(r) < W for r in fwd) and any(span(r) > W for r in fwd)     # insertions read, deletions jumped
    assert any(0 < r[4] < 130 for r in exp)
    # the TSV writer walks the same enumeration on the host (node paths): one line per row, same k-mers
    paths = write_region_tsvs(idx, rows, str(tmp_path / "out"))
    lines = [ln.split("\t") for p_ in paths for ln in open(p_).read().splitlines()]
    assert len(lines) == len(exp) and sorted(l[1] for l in lines) == sorted(r[1] for r in exp)
    assert all(l[6].endswith(",") and len(l) == 7 for l in lines)
    g.close()


def test_insertion_corner_cases_on_a_hand_made_graph():
    """An insertion longer than the window, two insertions at one anchor, an insertion at the anchor of a
    deletion and behind a SNP, an insertion behind the last base of the region."""
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex
    from oracle import extract_oracle as xo
    ref = b"ACGTTGCAAGGCTTACGATCGATTACA"
    v = xo.Variants()
    H = 6
    v.n_haplotypes = H
    car = lambda *h: np.isin(np.arange(H), h)
    v.add(3, 0, alts=["A", "C"], carriers=[car(0, 1), car(2)])
    v.add(3, 1, seq=b"GGGGGGGG", carriers=[car(0)])            # 8 inserted bases (> W below)
    v.add(3, 1, seq=b"T", carriers=[car(3)])
    v.add(3, 2, length=2, carriers=[car(4)])
    v.add(10, 1, seq=b"CA", carriers=[car(1, 5)])
    v.add(12, 0, alts=["G"], carriers=[car(5)])
    v.add(19, 1, seq=b"TTT", carriers=[car(2, 3)])
    pos = np.array(v.pos, dtype=np.int32)
    n = len(v)
    alt = np.zeros((n, 3), np.uint8)
    bits = np.zeros((n, 3, 1), np.uint64)
    dl, il, io, pool = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32), b""
    for i in range(n):
        for k, a in enumerate(v.alts[i]):
            alt[i, k] = ord(a)
        for k, c in enumerate(v.carriers[i]):
            bits[i, k, 0] = np.packbits(np.concatenate([c, np.zeros(64 - H, bool)]), bitorder="little").view(np.uint64)[0]
        dl[i] = v.length[i]
        if v.kind[i] == 1:
            il[i], io[i] = len(v.seq[i]), len(pool)
            pool += v.seq[i]
    n_alts = np.array([max(1, len(a)) for a in v.alts], np.uint8)
    idx = GraphIndex("c", np.frombuffer(ref, np.uint8), pos, n_alts, alt, bits, H, del_len=dl, ins_len=il, ins_off=io,
                     ins_bases=np.frombuffer(pool, np.uint8))
    g = DeviceGraph(idx)
    for W, regions in ((5, [(0, 27), (4, 20), (2, 11)]), (11, [(0, 27)]), (1, [(0, 27)])):
        rows = g.extract(regions, W)
        exp = []
        for s, e in regions:
            exp += xo.enumerate_region_variants("c", ref, v, s, e, W, with_counts=True)
        assert len(rows) == len(exp) > 0
        strand = [chr(c) for c in rows.strand.cpu().numpy()]
        got = list(zip([k.tobytes().decode() for k in rows.kmers.cpu().numpy()],
                       [f"c:{a}{s}" for a, s in zip(rows.start.cpu().numpy(), strand)],
                       [f"c:{a}{s}" for a, s in zip(rows.stop.cpu().numpy(), strand)],
                       rows.freq.cpu().numpy().tolist(), ["ref" if x else "non.ref" for x in rows.is_ref.cpu().numpy()]))
        assert got == [r[1:6] for r in exp], W
        if W == 5:
            assert ("GGGGG", "c:4+", "c:4+", 1, "non.ref") in got      # never leaves the insertion it starts in
    g.close()


def test_graph_without_sites_and_plan_errors(tmp_path):
    from grafimo_amd import _native as nv
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex
    ref = np.frombuffer(b"ACGTTGCANNACGT" * 10, dtype=np.uint8)
    idx = GraphIndex("c", ref, [], [], np.zeros((0, 3), np.uint8), None, 0)
    g = DeviceGraph(idx)
    rows = g.extract([(0, len(ref))], 6)
    assert len(rows) == 2 * (len(ref) - 6 + 1) and int(rows.is_ref.min()) == 1 and int(rows.freq.max()) == 0
    k = rows.kmers.cpu().numpy()
    assert k[0].tobytes() == b"ACGTTG" and k[1].tobytes() == b"CAACGT"        # reverse complement, N kept
    assert k[2 * 4].tobytes() == b"TGCANN" and k[2 * 4 + 1].tobytes() == b"NNTGCA"
    assert len(g.extract([], 6)) == 0 and len(g.extract([(5, 7)], 6)) == 0
    with pytest.raises(nv.NativeError):
        g.extract([(0, 50)], 65)
    g.close()
    # 25 sites with 3 alternates each inside one window: 4^25 walks -> refused, not attempted
    n = 25
    big = GraphIndex("c", ref, list(range(n)), [3] * n, np.tile(np.frombuffer(b"CGT", np.uint8), (n, 1)), None, 0)
    g = DeviceGraph(big)
    with pytest.raises(nv.NativeError) as ei:
        g.extract([(0, 40)], 30)
    assert ei.value.code == nv.GFM_ERR_OVERFLOW
    g.close()


def test_extraction_feeds_scoring_without_a_tsv(tmp_path):
    """compute_results_from_graph (extraction kernel -> score kernel in HBM) == the oracle's
    compute_results over the oracle's TSV rows, for the flag combinations that change the row set."""
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex, compute_results_from_graph
    from grafimo_amd.motif_ops import build_motif_meme_host
    from grafimo_amd.workflow import Findmotif
    from oracle import oracle as orc
    fasta, vcf = make_graph_files(str(tmp_path), chrom="7", length=6000, n_sites=400, seed=77)
    regions = [(0, 900), (1200, 2500), (3000, 5990)]
    motif = build_motif_meme_host(os.path.join(REF_DATA, "MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
    rows = _oracle_rows(fasta, vcf, "7", regions, 19)
    seqdir = tmp_path / "oracle_tsv" / "width_19"
    seqdir.mkdir(parents=True)
    for s, e in regions:
        with open(seqdir / f"7_{s}-{e}.tsv", "w") as fh:
            for r in rows:
                if r[0] == f"7:{s}-{e}":
                    fh.write("\t".join(str(x) for x in r) + "\n")
    md = dict(score_matrix=motif.dense_score_matrix(),
              pmf=orc.comp_pval_mat(motif.dense_score_matrix(), motif.dense_bg()), min_val=motif.min_val,
              scale=motif.scale, offset=float(motif.offset), width=19, motif_id=motif.motif_id,
              motif_name=motif.motif_name)
    g = DeviceGraph(GraphIndex.from_fasta_vcf(fasta, vcf, "7", allow_skipped=True))
    for kw in [dict(threshold=0.05), dict(threshold=0.5, qval_t=True, recomb=True),
               dict(threshold=0.02, no_reverse=True), dict(threshold=0.05, no_qvalue=True, recomb=True)]:
        with contextlib.redirect_stdout(io.StringIO()) as out:
            df = compute_results_from_graph(motif, g, regions, True, Findmotif(**kw))
        ref = orc.compute_results(md, str(tmp_path / "oracle_tsv"), threshold=kw["threshold"],
                                  qval_t=kw.get("qval_t", False), no_qvalue=kw.get("no_qvalue", False),
                                  no_reverse=kw.get("no_reverse", False), recomb=kw.get("recomb", False))
        exp = pd.DataFrame({c: ref[c] for c in ref if not c.startswith("_")})
        assert f"Scanned sequences:\t{ref['_scanned']}" in out.getvalue()
        assert list(df.columns) == list(exp.columns) and len(df) == len(exp) > 0, kw
        key = ["p-value", "sequence_name", "start", "stop", "strand", "matched_sequence", "haplotype_frequency"]
        a = df.sort_values(key).reset_index(drop=True)
        b = exp.sort_values(key).reset_index(drop=True)
        for c in exp.columns:
            if b[c].dtype.kind == "f":
                np.testing.assert_allclose(a[c].to_numpy(float), b[c].to_numpy(float), rtol=1e-9, atol=0)
            else:
                assert (a[c].astype(str) == b[c].astype(str)).all(), (kw, c)
    g.close()


def test_several_graphs_share_one_scoring_pass_and_cli_mode(tmp_path, capsys):
    """q-values are global per motif: two (graph, regions) entries == one entry with all regions; and the
    command line without vg: -l FASTA -v VCF -b BED writes the same table."""
    from grafimo_amd.__main__ import main
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex, compute_results_from_graph, read_bed_regions
    from grafimo_amd.motif_ops import build_motif_meme_host
    from grafimo_amd.workflow import Findmotif
    fasta, vcf = make_graph_files(str(tmp_path), chrom="7", length=4000, n_sites=300, seed=91, gz=False)
    r1, r2 = [(10, 800), (900, 1500)], [(2000, 3900)]
    bed = tmp_path / "regions.bed"
    # UCSC BED as the reference reads it (extract_regions.py:395-433): only lines starting with "chr" count
    bed.write_text("track name=test\n" + "".join(f"chr7\t{s}\t{e}\tpeak\n" for s, e in r1 + r2) + "7\t5\t50\n")
    assert read_bed_regions(str(bed)) == {"chr7": r1 + r2}
    motif = build_motif_meme_host(os.path.join(REF_DATA, "MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
    g = DeviceGraph(GraphIndex.from_fasta_vcf(fasta, vcf, "7", allow_skipped=True))
    kw = Findmotif(threshold=0.05, recomb=True)
    with contextlib.redirect_stdout(io.StringIO()):
        one = compute_results_from_graph(motif, g, r1 + r2, True, kw)
        two = compute_results_from_graph(motif, [g, g], [r1, r2], True, kw)
    assert len(one) > 0 and one.equals(two)
    g.close()
    out = tmp_path / "cli_out"
    with pytest.raises(Exception) as e:                 # strict: records with symbolic ALTs are an error
        main(["-m", os.path.join(REF_DATA, "MA0139.1.meme"), "-l", fasta, "-v", vcf, "-b", str(bed), "-t", "0.05",
              "--recomb", "-o", str(out), "--debug", "--strict-variants"])
    assert "strict variant handling" in str(e.value)
    main(["-m", os.path.join(REF_DATA, "MA0139.1.meme"), "-l", fasta, "-v", vcf, "-b", str(bed), "-t", "0.05",
          "--recomb", "-o", str(out), "--verbose"])         # default: left out with a warning, as vg construct does
    assert "deletions), 130 haplotypes" in capsys.readouterr().out
    tsv = pd.read_csv(out / "grafimo_out.tsv", sep="\t", index_col=0)
    assert len(tsv) == len(one) and list(tsv["matched_sequence"]) == list(one["matched_sequence"])
    np.testing.assert_allclose(tsv["q-value"].to_numpy(), one["q-value"].to_numpy(), rtol=1e-12)
    with pytest.raises(SystemExit):
        main(["-m", "x.meme", "-l", fasta])            # incomplete graph inputs


def test_scan_graph_adapter_feeds_compute_results(tmp_path, capsys, monkeypatch):
    """scan_graph(widths, args_obj, debug) -> tmpdir with width_W/CHR_S-E.tsv, called the way
    grafimo.findmotif does (grafimo.py:174-179), then compute_results on that directory == the direct
    compute_results_from_graph table.  The graph comes from the index saved next to the XG name.  (The FILES are what is
    tested here: this function holds grafimo_amd's compute_results, which alone would get it the manifest.)"""
    import shutil
    monkeypatch.setenv("GRAFIMO_SCAN_OUTPUT", "tsv")
    from grafimo_amd.extract_regions import (DeviceGraph, GraphIndex, compute_results_from_graph, scan_graph)
    from grafimo_amd.motif_ops import build_motif_meme_host
    from grafimo_amd.score_sequences import compute_results
    from grafimo_amd.workflow import Findmotif
    fasta, vcf = os.path.join(REF_DATA, "test.fa"), os.path.join(REF_DATA, "test.vcf.gz")
    idx = GraphIndex.from_fasta_vcf(fasta, vcf, "x")
    gdir = tmp_path / "graphs"
    gdir.mkdir()
    saved = idx.save(str(gdir / "x"))
    assert saved.endswith("x.gfmidx.npz")
    back = GraphIndex.load(saved)
    assert back.chrom == "x" and np.array_equal(back.ref, idx.ref) and np.array_equal(back.pos, idx.pos)
    assert np.array_equal(back.alt_bits, idx.alt_bits) and back.n_haplotypes == idx.n_haplotypes
    regions = [(0, 400), (380, 1001)]
    bed = tmp_path / "r.bed"
    bed.write_text("".join(f"chrx\t{s}\t{e}\n" for s, e in regions))
    motif = build_motif_meme_host(os.path.join(REF_DATA, "MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
    wf = Findmotif(threshold=0.05, recomb=True, graph_genome_dir=str(gdir), bedfile=str(bed), cores=2)
    loc = scan_graph({19, 12}, wf, True)
    try:
        assert os.path.basename(loc).startswith("grafimo_")
        assert sorted(os.listdir(loc)) == ["width_12", "width_19"]
        assert sorted(os.listdir(os.path.join(loc, "width_19"))) == ["x_0-400.tsv", "x_380-1001.tsv"]
        first = open(os.path.join(loc, "width_19", "x_0-400.tsv")).readline().split("\t")
        assert first[0] == "x:0-400" and len(first[1]) == 19 and first[2].startswith("x:") and len(first) == 7
        with contextlib.redirect_stdout(io.StringIO()):
            via_files = compute_results(motif, loc, True, wf)
            g = DeviceGraph(idx)
            direct = compute_results_from_graph(motif, g, regions, True, wf)
            g.close()
    finally:
        shutil.rmtree(loc)
    assert len(direct) > 0
    key = ["p-value", "sequence_name", "start", "stop", "strand"]
    a = via_files.sort_values(key).reset_index(drop=True)
    b = direct.sort_values(key).reset_index(drop=True)
    assert a.equals(b)
    # a single indexed graph (-g): the index sits beside the XG path the workflow names
    wf1 = Findmotif(threshold=0.05, graph_genome=str(gdir / "x.xg"), bedfile=str(bed), chroms=["x"])
    loc = scan_graph({19}, wf1, True)
    assert sorted(os.listdir(os.path.join(loc, "width_19"))) == ["x_0-400.tsv", "x_380-1001.tsv"]
    shutil.rmtree(loc)
    with pytest.raises(Exception) as e:
        scan_graph({19}, Findmotif(graph_genome_dir=str(tmp_path), bedfile=str(bed)), True)
    assert "Unable to locate" in str(e.value)


def test_graph_pipeline_under_a_process_group(tmp_path):
    """The sharded form of compute_results_from_graph with an RCCL group of one rank: the row-count and
    histogram all-reduces are really issued, the table is unchanged."""
    import socket
    import torch.distributed as dist
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex, compute_results_from_graph
    from grafimo_amd.motif_ops import build_motif_meme_host
    from grafimo_amd.workflow import Findmotif
    fasta, vcf = make_graph_files(str(tmp_path), chrom="7", length=3000, n_sites=200, seed=12)
    regions = [(0, 1000), (1500, 2990)]
    motif = build_motif_meme_host(os.path.join(REF_DATA, "MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
    g = DeviceGraph(GraphIndex.from_fasta_vcf(fasta, vcf, "7", allow_skipped=True))
    kw = Findmotif(threshold=0.05, recomb=True)
    with contextlib.redirect_stdout(io.StringIO()):
        plain = compute_results_from_graph(motif, g, regions, True, kw)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev,
                            timeout=datetime.timedelta(seconds=60))
    try:
        with contextlib.redirect_stdout(io.StringIO()) as out:
            coll = compute_results_from_graph(motif, g, regions, True, kw, always_collective=True)
        assert "Scanned sequences" in out.getvalue()
    finally:
        dist.destroy_process_group()
    assert len(plain) > 0 and plain.equals(coll)
    g.close()


def test_scoring_fixture_rows_through_the_gpu(tmp_path):
    """The 704 rows of real vg output behind the reference's test_scoring, regenerated by the extraction
    kernel from the fixture's local graph (5 SNPs, one deletion, 5096 haplotypes), then scored: the same
    table the reference's expected scoring_results.tsv holds."""
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex, compute_results_from_graph, write_region_tsvs
    from grafimo_amd.motif_ops import build_motif_meme_host
    from grafimo_amd.workflow import Findmotif
    rows, refseq, sites, dels, S, E = scoring_fixture_graph()
    H = sites.n_haplotypes
    recs = [(int(p), 0, i) for i, p in enumerate(sites.pos)] + [(int(a), 1, j) for j, a in enumerate(dels.anchor)]
    recs.sort()
    hw = (H + 63) // 64
    bits = np.zeros((len(recs), 3, hw), dtype=np.uint64)
    alt = np.zeros((len(recs), 3), dtype=np.uint8)
    for k, (_, kind, j) in enumerate(recs):
        carry = np.zeros(hw * 64, dtype=bool)
        carry[:H] = dels.hap[j] if kind else sites.hap[j] == 1
        bits[k, 0] = np.packbits(carry, bitorder="little").view(np.uint64)
        if not kind:
            alt[k, 0] = ord(sites.alts[j][0])
    ref = np.frombuffer(refseq, dtype=np.uint8)
    idx = GraphIndex("22", ref, [r[0] for r in recs], [1] * len(recs), alt, bits, H,
                     del_len=[int(dels.length[j]) if kind else 0 for _, kind, j in recs])
    g = DeviceGraph(idx)
    ext = g.extract([(0, E - S)], 19)
    assert len(ext) == 704
    at = lambda s_: int(s_.split(":")[1][:-1])
    exp = sorted((r[1], at(r[2]) - S, at(r[3]) - S, r[2][-1], int(r[4]), r[5]) for r in rows)
    km = ext.kmers.cpu().numpy()
    got = sorted((km[i].tobytes().decode(), int(ext.start[i]), int(ext.stop[i]), chr(int(ext.strand[i])),
                  int(ext.freq[i]), "ref" if int(ext.is_ref[i]) else "non.ref") for i in range(704))
    assert got == exp
    # the TSV writer: node paths too (ids start at 1 here; vg's 849116.. are the same ids shifted, and its first
    # node ends after one base, which a chromosome starting at the region cannot reproduce: compare the rest)
    path = write_region_tsvs(idx, ext, str(tmp_path))[0]
    mine = {}
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        mine[(f[1], f[2], f[3])] = f[6]
    for r in rows:
        key = (r[1], f"22:{at(r[2]) - S}{r[2][-1]}", f"22:{at(r[3]) - S}{r[3][-1]}")
        ids = [int(x[:-1]) - 849116 for x in r[6].strip(",").split(",")]
        got_ids = [int(x[:-1]) for x in mine[key].strip(",").split(",")]
        if 0 not in ids:                       # walks that do not touch vg's one-base first node
            assert [i - 1 for i in ids] == [i - 1 for i in got_ids] or ids == got_ids, key
    # scoring them reproduces the reference's expected table (test_scoring: threshold 1, recomb, q-values)
    motif = build_motif_meme_host(os.path.join(REF_DATA, "MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
    with contextlib.redirect_stdout(io.StringIO()):
        df = compute_results_from_graph(motif, g, [(0, E - S)], True, Findmotif(threshold=1.0, recomb=True))
    ref_df = pd.read_csv(os.path.join(REF_DATA, "scoring_results.tsv"), sep="\t", index_col=0)
    assert len(df) == len(ref_df) == 704
    key = ["matched_sequence", "strand", "haplotype_frequency"]
    a = df.assign(start=df["start"] + S, stop=df["stop"] + S).sort_values(key + ["start"]).reset_index(drop=True)
    b = ref_df.sort_values(key + ["start"]).reset_index(drop=True)
    for c in ("start", "stop", "strand", "matched_sequence", "haplotype_frequency", "reference"):
        assert (a[c].astype(str) == b[c].astype(str)).all(), c
    for c in ("score", "p-value", "q-value"):
        np.testing.assert_allclose(a[c].to_numpy(float), b[c].to_numpy(float), rtol=1e-9, atol=0)
    g.close()


def test_jumping_walks_must_end_inside_the_region():
    """A walk that jumps a deletion and would end past the region's end is not reported (both ends of a
    walk lie inside the region); the same start still yields the walk that stays on the reference."""
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex
    from oracle import extract_oracle as xo
    rng = np.random.default_rng(4)
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=60)
    idx = GraphIndex("c", ref, [25], [1], np.zeros((1, 3), np.uint8), None, 0, del_len=[5])
    sites = xo.Sites([], [], [], np.zeros((0, 0), np.int8))
    dels = xo.Dels([25], [5], np.zeros((1, 0), bool))
    g = DeviceGraph(idx)
    for region in [(0, 32), (0, 60), (20, 34), (26, 45)]:
        rows = g.extract([region], 10)
        exp = xo.enumerate_region_graph("c", ref.tobytes(), sites, dels, region[0], region[1], 10)
        km = rows.kmers.cpu().numpy()
        got = [(km[i].tobytes().decode(), int(rows.start[i]), int(rows.stop[i])) for i in range(len(rows))]
        assert got == [(r[1], int(r[2].split(":")[1][:-1]), int(r[3].split(":")[1][:-1])) for r in exp], region
    # region (0, 32): start 18 reaches base 25 with two bases to go -- the jump would end at 33 > 32
    rows = g.extract([(0, 32)], 10)
    starts = rows.start.cpu().numpy()[0::2].tolist()
    stops = rows.stop.cpu().numpy()[0::2].tolist()
    assert starts.count(18) == 1 and max(stops) <= 32 and 15 in [b - a for a, b in zip(starts, stops)]
    g.close()


@pytest.mark.parametrize("kinds", ["s", "sd", "si", "sidm", "sD", "sO", "sidmDO", "sc", "sidmDOcS"])
def test_hip_rows_equal_the_per_haplotype_brute_force(tmp_path, kinds):
    """the kernels' rows against an algorithm that enumerates no walks at all -- every haplotype of the
    VCF as a linear sequence, W-windows slid over it (oracle/extract_bruteforce.py): a row's haplotype count is the
    number of haplotypes that hold this k-mer at these coordinates, rows no haplotype carries report 0, and no
    window of any haplotype is missing.  Also the reference's own test graph (test.fa + test.vcf.gz)."""
    from extract_helpers import make_consistent_graph_files
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex
    from oracle import extract_bruteforce as bf
    from oracle import extract_oracle as xo

    def hip_rows(rows):
        km = rows.kmers.cpu().numpy()
        st, sp = rows.start.cpu().numpy(), rows.stop.cpu().numpy()
        sd, fr, rf = rows.strand.cpu().numpy(), rows.freq.cpu().numpy(), rows.is_ref.cpu().numpy()
        for i in range(len(rows)):
            yield (km[i].tobytes(), int(st[i]), int(sp[i]), chr(sd[i]), int(fr[i]), "ref" if rf[i] else "non.ref")

    cases = []
    for seed in range(2):
        fasta, vcf = make_consistent_graph_files(str(tmp_path), chrom="c", length=420, n_samples=20,
                                                 seed=700 + 10 * len(kinds) + seed, kinds=kinds)
        cases.append((fasta, vcf, "c", [((0, 90), 19), ((100, 260), 8), ((250, 420), 30), ((30, 200), 3), ((395, 420), 12)]))
    if kinds == "s":
        cases.append((os.path.join(REF_DATA, "test.fa"), os.path.join(REF_DATA, "test.vcf.gz"), "x",
                      [((0, 20), 19), ((0, 50), 19), ((5, 45), 8), ((0, 50), 30)]))
    if kinds == "si":     # records with more than three ALT alleles (an STR site: four insertion lengths and a substitution)
        ref = "ACGTTGCAATCGGATCCATGCAAGTCTAGGCTTAACG"
        head = "##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ta\tb\tc\n"
        (tmp_path / "m.vcf").write_text(head + "s\t9\t.\tA\tG,AT,ATT,ATTT,ATTTT\t.\t.\t.\tGT\t1|2\t3|4\t5|0\n"
                                               "s\t20\t.\tG\tGA,GAC,T,C,A,GACA\t.\t.\t.\tGT\t6|3\t4|5\t1|2\n")
        (tmp_path / "m.fa").write_text(">s\n" + ref + "\n")
        cases.append((str(tmp_path / "m.fa"), str(tmp_path / "m.vcf"), "s", [((0, len(ref)), 3), ((0, len(ref)), 8), ((0, len(ref)), 19)]))
    total = 0
    for fasta, vcf, chrom, plans in cases:
        ref = xo.read_fasta(fasta)[chrom]
        recs, H = bf.read_vcf_records(vcf, chrom)
        assert bf.consistent(ref, recs, H)
        idx = GraphIndex.from_fasta_vcf(fasta, vcf, chrom)
        # every allele is part of the graph, except those of records with a symbolic ALT (kind S: in no graph)
        assert (idx.skipped > 0) == ("S" in kinds) and idx.n_haplotypes == H
        g = DeviceGraph(idx)
        for (S, E), W in plans:
            freq, flags = bf.window_counts(ref, recs, H, S, E, W)
            carried, n = bf.check_rows(hip_rows(g.extract([(S, E)], W)), freq, flags)
            assert n >= 2 * carried
            total += carried
        g.close()
    assert total > 500


@pytest.mark.gpu
def test_deferred_and_in_place_haplotype_counts_agree(tmp_path, monkeypatch):
    """The emit kernels leave the haplotype counts that need the bitsets (four or more sites in a window, constraints on
    sites that are no neighbours) to graph_count_jobs_kernel; a deletion walk that finds the job lists full counts in
    place.  Both ways against the oracle on a dense graph, and against each other (GRAFIMO_EXTRACT_DEL_POOL=0 leaves
    no room for deletion jobs)."""
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex
    from oracle import extract_oracle as xo
    W = 24
    fasta, vcf = make_graph_files(str(tmp_path), chrom="7", length=1500, n_sites=200, n_samples=65, seed=91, rich=True)
    idx = GraphIndex.from_fasta_vcf(fasta, vcf, "7")
    ref = xo.read_fasta(fasta)["7"]
    v = xo.read_vcf_variants(vcf, "7")
    regions = [(0, 110), (420, 520), (1400, 1500)]       # (sized for the Python enumerator: a fraction of a second)
    exp = []
    for s, e in regions:
        exp += xo.enumerate_region_variants("7", ref, v, s, e, W, with_counts=True)
    want = [r[4] for r in exp]
    assert len(want) > 3000 and sum(1 for c in want if c > 0) > 500           # windows of four and more sites among them
    g = DeviceGraph(idx)
    got = {}
    for pool in ("", "0", "37"):
        if pool:
            monkeypatch.setenv("GRAFIMO_EXTRACT_DEL_POOL", pool)
        rows = g.extract(regions, W)
        got[pool] = rows.freq.cpu().numpy().tolist()
        assert [k.tobytes().decode() for k in rows.kmers.cpu().numpy()] == [r[1] for r in exp]
    g.close()
    assert got[""] == want and got["0"] == want and got["37"] == want


@pytest.mark.gpu
def test_windows_of_millions_of_walks_are_planned_not_refused():
    """Twenty-four neighbouring biallelic sites inside one 30-mer are 2^24 walks per window: `vg find -K 30 -E` lists them
    all, so the plan is NOT refused (round 3 refused windows beyond 2^20 walks).  Three such windows: the row
    count is the allele product, the k-mers of a window are all different, sampled rows are the reference with the walk's
    mixed-radix alleles put in, and the FUSED path's score histogram equals the bincount of the scores of these rows."""
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex
    rng = np.random.default_rng(3)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = acgt[rng.integers(0, 4, 600)]
    pos = np.arange(300, 324, dtype=np.int32)
    alt = np.zeros((len(pos), 3), np.uint8)
    alt[:, 0] = np.where(ref[pos] == ord("A"), ord("C"), ord("A"))
    idx = GraphIndex("c", ref, pos, np.ones(len(pos), np.uint8), alt, None, 0)
    g = DeviceGraph(idx)
    W, S = 30, 294                                    # windows 294, 295, 296 hold all 24 sites
    rows = g.extract([(0, 200), (S, S + 2 + W)], W)
    n_plain = 2 * 171
    assert len(rows) == n_plain + 2 * 3 * (1 << 24)
    first = rows.kmers[n_plain:n_plain + 2 * (1 << 24):2]                      # forward rows of window 294
    code = (first >> 1) & 3
    key = torch.zeros(first.shape[0], dtype=torch.int64, device=first.device)
    for j in range(W):
        key = key * 4 + code[:, j].to(torch.int64)
    assert int(torch.unique(key).numel()) == 1 << 24
    take = rng.integers(n_plain, len(rows), 300)
    km, st, sp = rows.kmers[take].cpu().numpy(), rows.start[take].cpu().numpy(), rows.stop[take].cpu().numpy()
    sd, wk, rg = rows.strand[take].cpu().numpy(), rows.walk[take].cpu().numpy(), rows.region[take].cpu().numpy()
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    for i in range(len(take)):
        p = int(st[i]) if sd[i] == ord("+") else int(sp[i])
        _, alleles = idx.walk_alleles(p, W, int(wk[i]))
        want = ref[p:p + W].copy()
        for k, a in enumerate(alleles):
            if a:
                want[pos[k] - p] = alt[k, a - 1]
        if sd[i] == ord("-"):
            want = np.array([comp[c] for c in want[::-1]], np.uint8)
        assert km[i].tobytes() == want.tobytes() and rg[i] == 1 and abs(int(sp[i]) - int(st[i])) == W
    # the fused path walks the same windows (heavy tiles: 2^24 walks per window) and books the same scores
    rec = __import__("grafimo_amd.synth", fromlist=["x"]).synthetic_motif(W, np.random.default_rng(5), np.full(4, 0.25))
    dm = DeviceMotif(rec["sm"], rec["bg"], rec["min_val"], rec["scale"], rec["offset"])
    sc = torch.empty(len(rows), dtype=torch.int32, device=rows.kmers.device)
    dm.score(rows.kmers, sc)
    hist = torch.zeros(dm.L, dtype=torch.int64, device=sc.device)
    reg = np.array([(0, 200), (S, S + 2 + W)], dtype=np.int64)
    g.score(dm, np.ascontiguousarray(reg[:, 0]), np.ascontiguousarray(reg[:, 1]), dm.pvalue_cutoff(1e-7), hist=hist)
    count, n_rows, over, recs = g.fused_results()
    assert n_rows == len(rows) and not over
    assert torch.equal(hist, torch.bincount(sc.long(), minlength=dm.L))
    dm.close()
    g.close()


@pytest.mark.gpu
def test_regions_beyond_one_plan_are_cut_into_pieces(tmp_path):
    """A plan holds at most 2^31 rows (GRAFIMO_PLAN_MAX_WALKS lowers that for this test; tests/plan_pieces_probe.py runs in
    its own process): extract() then plans the regions in pieces -- the region list halved, then a region's range of window
    starts -- and joins them: the same rows, the same region indices, as one plan gives.  Only a single window beyond the
    cap has no rows to give; the message names it."""
    import subprocess
    import sys
    from conftest import ROOT
    probe = os.path.join(ROOT, "tests", "plan_pieces_probe.py")
    outs, pieces = [], []
    for cap in (1 << 17, 0x3fffffff):
        out = str(tmp_path / f"rows_{cap}.npz")
        r = subprocess.run([sys.executable, probe, "rows", out, str(tmp_path / f"tsv_{cap}")], capture_output=True, text=True,
                           env=dict(os.environ, GRAFIMO_PLAN_MAX_WALKS=str(cap)), timeout=600)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
        pieces.append(int(r.stdout.split("pieces")[1].split()[0]))
        outs.append(np.load(out))
    assert pieces[0] > 3 and pieces[1] == 1
    assert len(outs[0]["st"]) > 1_500_000 and (np.diff(outs[0]["rg"]) >= 0).all() and set(outs[0]["rg"].tolist()) == {0, 1, 2, 3, 4}
    for k in ("km", "st", "sp", "rg", "wk", "fr", "sd", "rf"):
        assert np.array_equal(outs[0][k], outs[1][k]), k
    # the TSV files written piece by piece (a region cut into several plans is appended to) == written from one plan
    da, db = tmp_path / f"tsv_{1 << 17}" / "width_24", tmp_path / f"tsv_{0x3fffffff}" / "width_24"
    names = sorted(os.listdir(db))
    assert names == sorted(os.listdir(da)) == ["c_0-120.tsv", "c_100-130.tsv", "c_280-340.tsv", "c_590-640.tsv", "c_700-900.tsv"]
    for f in names:
        assert open(da / f, "rb").read() == open(db / f, "rb").read(), f
    assert sum(open(db / f, "rb").read().count(b"\n") for f in names) == len(outs[1]["st"])
    r = subprocess.run([sys.executable, probe, "single"], capture_output=True, text=True,
                       env=dict(os.environ, GRAFIMO_PLAN_MAX_WALKS=str(1 << 15)), timeout=600)
    assert "REFUSED -7" in r.stdout and "c:280-340" in r.stdout and "starts at c:" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
