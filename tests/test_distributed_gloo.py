"""The N > 1 path on CPU: world_size-2 gloo runs of the sharded orchestration
(grafimo_amd/distributed.py) with a stand-in scoring backend built on the CPU oracle.
What is under test is the sharding, the histogram all-reduce, global row ids, the q-table
from the GLOBAL histogram and the gather/merge of hit tables -- not the kernels."""
import contextlib
import io
import os
import socket
import sys

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN, REF_DATA, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _golden_ctcf():
    import json
    with open(os.path.join(GOLDEN, "motifs.json")) as fh:
        g = json.load(fh)["ctcf_meme_unif"]["motifs"][0]
    g["pmf"] = np.load(os.path.join(GOLDEN, "pmf.npz"))[g["pmf_key"]]
    return g


def _make_backend(g):
    import torch
    from grafimo_amd.distributed import ScanBackend
    from oracle import oracle as orc

    class OracleBackend(ScanBackend):
        """test stand-in: oracle arithmetic, torch CPU tensors for the collectives"""
        device = torch.device("cpu")
        L = 19001

        def __init__(self):
            self.ptab = orc.p_table(g["pmf"])
            self.sm = np.array(g["score_matrix"], dtype=np.int64)

        def score(self, kmers):
            if len(kmers) == 0:
                return np.empty(0, np.int32), torch.zeros(self.L, dtype=torch.int64)
            sc, _ = orc.score_kmers_table(kmers, self.sm, self.ptab, g["min_val"])
            return sc, torch.from_numpy(np.bincount(sc, minlength=self.L).astype(np.int64))

        def tables(self, hist, threshold, on_qvalue):
            h = hist.numpy()
            scores = np.repeat(np.arange(self.L), h)
            q_rows = orc.fdr_bh(self.ptab[scores])
            q = np.ones(self.L)
            q[scores] = q_rows                      # all rows of one score share q
            val = q if on_qvalue else self.ptab
            occ = np.nonzero(h)[0]
            ok = occ[val[occ] < threshold]
            # cutoff semantics of gfm_qvalue_table: smallest score whose value is < threshold
            cutoff = int(ok.min()) if len(ok) else self.L
            return q, cutoff, int(h.sum())

        def pvalue_cutoff(self, threshold):
            idx = np.nonzero(self.ptab < threshold)[0]
            return int(idx.min()) if len(idx) else self.L

        def select_host(self, scaled_all, cutoff, row_base):
            rows = np.nonzero(scaled_all >= cutoff)[0]
            return rows + row_base, scaled_all[rows]

        def annotate(self, scaled):
            lo = scaled / g["scale"] + 19 * g["offset"]
            return lo.astype(np.float64), self.ptab[scaled]

        def begin(self, motifs, files, width, no_reverse, threads, threshold, on_qvalue, want_qvalues):
            """the file-level seam: the product's own TSV ingest (host C++), oracle scoring, tables in finish()
            from the histogram as it is THEN (the orchestration all-reduces it in between)"""
            from types import SimpleNamespace
            from grafimo_amd.score_sequences import KmerTable
            assert len(motifs) == 1 and width == 19
            table = KmerTable(files, width, no_reverse, threads)
            scaled, hist = self.score(table.kmers)
            be = self

            class Shard:
                device = be.device
                n = table.n

                def __init__(self):
                    self.hist = hist.reshape(1, -1).clone()

                def finish(self):
                    q = None
                    cutoff = be.pvalue_cutoff(threshold)
                    if want_qvalues:
                        q, cutoff_q, _ = be.tables(self.hist[0], threshold, on_qvalue)
                        cutoff = cutoff_q if on_qvalue else cutoff
                    rows = np.nonzero(scaled >= cutoff)[0]
                    lo, pv = be.annotate(scaled[rows])
                    h = SimpleNamespace(rows=rows, scaled=scaled[rows], logodds=lo, pvalue=pv,
                                        qvalue=q[scaled[rows]] if want_qvalues else None,
                                        kmers=table.kmers[rows], start=table.start[rows], stop=table.stop[rows],
                                        strand=table.strand[rows], freq=table.freq[rows], is_ref=table.is_ref[rows],
                                        name_id=table.name_id[rows])
                    return [h], table.names

            return Shard()

    return OracleBackend()


def _worker(rank, world, port, seqdir, kw, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from grafimo_amd.distributed import compute_results_sharded
    from grafimo_amd.motif import Motif
    from grafimo_amd.workflow import Findmotif
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = _golden_ctcf()
        m = Motif(np.array(g["probs"]), 19, ["A", "C", "G", "T"], g["motif_id"], g["motif_name"],
                  {n: i for i, n in enumerate("ACGT")})
        with contextlib.redirect_stdout(io.StringIO()) as buf:
            df = compute_results_sharded(m, seqdir, True, Findmotif(**kw), backend=_make_backend(g))
        if rank == 0:
            df.to_pickle(os.path.join(outdir, "df.pkl"))
            with open(os.path.join(outdir, "stdout.txt"), "w") as fh:
                fh.write(buf.getvalue())
        else:
            assert df is None
    finally:
        dist.destroy_process_group()


def _run(world, seqdir, kw, outdir):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), seqdir, kw, outdir), nprocs=world, join=True)
    with open(os.path.join(outdir, "stdout.txt")) as fh:
        return pd.read_pickle(os.path.join(outdir, "df.pkl")), fh.read()


@pytest.fixture(scope="module")
def tsv_dir(tmp_path_factory):
    from grafimo_amd import synth
    g = _golden_ctcf()
    d = tmp_path_factory.mktemp("seqs")
    batch = synth.make_batch(7, 400, 19, np.array(g["probs"]), synth.seed_for(1))   # 7 files: ragged split
    synth.write_tsv_dir(batch, str(d))
    return str(d)


def _oracle_table(seqdir, kw):
    from oracle import oracle as orc
    g = _golden_ctcf()
    md = dict(score_matrix=np.array(g["score_matrix"]), pmf=g["pmf"], min_val=g["min_val"],
              scale=g["scale"], offset=g["offset"], width=19, motif_id=g["motif_id"],
              motif_name=g["motif_name"])
    ref = orc.compute_results(md, seqdir, threshold=kw["threshold"], qval_t=kw.get("qval_t", False),
                              no_qvalue=kw.get("no_qvalue", False), no_reverse=kw.get("no_reverse", False),
                              recomb=kw.get("recomb", False))
    return pd.DataFrame({c: ref[c] for c in ref if not c.startswith("_")}), ref["_scanned"]


@pytest.mark.parametrize("kw", [dict(threshold=1e-3), dict(threshold=0.3, qval_t=True),
                                dict(threshold=1e-2, no_qvalue=True, recomb=True),
                                dict(threshold=5e-2, no_reverse=True)])
def test_two_ranks_equal_the_single_process_table(tsv_dir, tmp_path, kw):
    df, out = _run(2, tsv_dir, kw, str(tmp_path))
    exp, scanned = _oracle_table(tsv_dir, kw)
    assert f"Scanned sequences:\t{scanned}" in out
    assert list(df.columns) == list(exp.columns) and len(df) == len(exp) and len(df) > 0
    key = ["p-value", "start", "stop", "strand", "matched_sequence"]
    a = df.sort_values(key).reset_index(drop=True)
    b = exp.sort_values(key).reset_index(drop=True)
    for c in exp.columns:
        if b[c].dtype.kind == "f":
            np.testing.assert_allclose(a[c].to_numpy(float), b[c].to_numpy(float), rtol=1e-12, atol=0)
        else:
            assert (a[c].astype(str) == b[c].astype(str)).all(), c
    assert (np.diff(df["p-value"].to_numpy()) >= 0).all()


def test_world_of_one_and_empty_shard(tsv_dir, tmp_path):
    """world_size 3 over 7 files of which a rank may get few; and the reference's 1-file fixture
    over 2 ranks leaves rank 1 with an EMPTY shard -- its histogram still joins the all-reduce."""
    kw = dict(threshold=0.5, qval_t=True, recomb=True)
    d3 = tmp_path / "w3"; d3.mkdir()
    df3, _ = _run(3, tsv_dir, kw, str(d3))
    exp, _ = _oracle_table(tsv_dir, kw)
    assert len(df3) == len(exp)
    d2 = tmp_path / "w2"; d2.mkdir()
    df, out = _run(2, REF_DATA, dict(threshold=1.0, recomb=True), str(d2))
    assert len(df) == 704 and "Scanned sequences:\t704" in out


@pytest.mark.parametrize("world", [4, 8])
def test_four_and_eight_ranks_with_ragged_and_empty_shards(tsv_dir, tmp_path, world):
    """7 files over 4 and over 8 ranks (at 8 at least one rank holds nothing): the same table as one
    process, for a p- and a q-value threshold; the hit rows reach rank 0 through gather_columns."""
    for j, kw in enumerate([dict(threshold=2e-3), dict(threshold=0.4, qval_t=True, recomb=True)]):
        d = tmp_path / f"w{world}_{j}"
        d.mkdir()
        df, out = _run(world, tsv_dir, kw, str(d))
        exp, scanned = _oracle_table(tsv_dir, kw)
        assert f"Scanned sequences:\t{scanned}" in out
        assert list(df.columns) == list(exp.columns) and len(df) == len(exp) and len(df) > 0
        key = ["p-value", "start", "stop", "strand", "matched_sequence"]
        a, b = df.sort_values(key).reset_index(drop=True), exp.sort_values(key).reset_index(drop=True)
        for c in exp.columns:
            if b[c].dtype.kind == "f":
                np.testing.assert_allclose(a[c].to_numpy(float), b[c].to_numpy(float), rtol=1e-12, atol=0)
            else:
                assert (a[c].astype(str) == b[c].astype(str)).all(), c


def _gather_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import pickle
    import torch
    import torch.distributed as dist
    from grafimo_amd.distributed import gather_columns, gather_names
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(100 + rank)
        k = [0, 5, 0, 1300][rank % 4] if rank < 4 else rank       # ragged, two empty ranks, one long list
        cols = dict(rows=np.arange(k, dtype=np.int64) + 10_000 * rank, sc=rng.integers(0, 19000, k).astype(np.int32),
                    p=rng.random(k), kmers=rng.integers(65, 85, (k, 19)).astype(np.uint8), flag=(np.arange(k) % 2).astype(np.uint8))
        got = gather_columns(cols, torch.device("cpu"))
        names = gather_names([f"chr{rank}:{i}-{i + 200}" for i in range(rank)], torch.device("cpu"))
        none_everywhere = gather_columns({"x": np.empty(0, np.int64)}, torch.device("cpu"))   # nobody has rows
        with open(os.path.join(outdir, f"r{rank}.pkl"), "wb") as fh:
            pickle.dump(dict(cols=cols, got=got, names=names, empty=none_everywhere), fh)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_gather_columns_moves_ragged_row_sets_to_rank_zero(tmp_path, world):
    import pickle
    import torch.multiprocessing as mp
    mp.spawn(_gather_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [pickle.load(open(tmp_path / f"r{r}.pkl", "rb")) for r in range(world)]
    got = parts[0]["got"]
    for name in ("rows", "sc", "p", "kmers", "flag"):
        want = np.concatenate([p["cols"][name] for p in parts], axis=0)
        assert got[name].dtype == want.dtype and np.array_equal(got[name], want), name
    assert all(p["got"] is None for p in parts[1:])
    assert parts[0]["names"] == [[f"chr{r}:{i}-{i + 200}" for i in range(r)] for r in range(world)]
    assert all(p["names"] is None for p in parts[1:])
    assert len(parts[0]["empty"]["x"]) == 0 and all(p["empty"] is None for p in parts[1:])


def test_shard_helpers(tmp_path):
    from grafimo_amd.distributed import shard_bounds, shard_files
    for n in [0, 1, 7, 8, 1000]:
        for w in [1, 2, 3, 8]:
            parts = [shard_bounds(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
    files = []
    for i, sz in enumerate([10, 10, 10, 500, 10, 10, 10, 10]):
        p = tmp_path / f"f{i:02d}.tsv"
        p.write_bytes(b"x" * sz)
        files.append(str(p))
    for w in [1, 2, 4, 8]:
        shards = [shard_files(files, w, r) for r in range(w)]
        assert sum(shards, []) == files                    # contiguous cover, order kept
    assert shard_files([], 4, 2) == []


def _golden(key):
    import json
    with open(os.path.join(GOLDEN, "motifs.json")) as fh:
        g = json.load(fh)[key]["motifs"][0]
    g["pmf"] = np.load(os.path.join(GOLDEN, "pmf.npz"))[g["pmf_key"]]
    return g


_SAME_WIDTH = ["ctcf_meme_unif", "ctcf_meme_bgnt", "ctcf_jaspar_bgnt_p1"]   # three W=19 motifs


def _same_width_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import pickle
    import torch.distributed as dist
    from grafimo_amd import synth
    from grafimo_amd.distributed import shard_bounds, sharded_scan, sharded_scan_same_width
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gs = [_golden(k) for k in _SAME_WIDTH]
        batch = synth.make_batch(5, 600, 19, np.array(gs[0]["probs"]), synth.seed_for(7))
        lo, hi = shard_bounds(len(batch), world, rank)
        mine = batch.kmers[lo:hi]
        backends = [_make_backend(g) for g in gs]
        calls = {"all_reduce": 0}
        real = dist.all_reduce

        def counting(t, *a, **k):
            calls["all_reduce"] += 1
            return real(t, *a, **k)
        dist.all_reduce = counting
        try:
            many = sharded_scan_same_width(backends, mine, 0.2, True, True)
        finally:
            dist.all_reduce = real
        assert calls["all_reduce"] == 2                       # row counts + ONE [M, L] histogram exchange
        for j, b in enumerate(backends):
            one = sharded_scan(b, mine, 0.2, True, True, select=b.select_host)
            for key in ("rows", "scaled", "logodds", "pvalue", "qvalue"):
                assert np.array_equal(many[j][key], one[key]), (j, key)
            assert many[j]["n_scored"] == one["n_scored"] == len(batch) and many[j]["row_base"] == lo
        bucket = [None] * world if rank == 0 else None
        dist.gather_object([(r["rows"], r["qvalue"]) for r in many], bucket, dst=0)
        if rank == 0:
            with open(os.path.join(outdir, "many.pkl"), "wb") as fh:
                pickle.dump(bucket, fh)
    finally:
        dist.destroy_process_group()


def test_same_width_set_shares_one_histogram_exchange(tmp_path):
    """Config 5 on N ranks: three W=19 motifs over sharded rows -- one [M, L] all-reduce for the
    set, results identical to per-motif sharded scans and to a single-process oracle run."""
    import pickle
    import torch.multiprocessing as mp
    from grafimo_amd import synth
    from oracle import oracle as orc
    mp.spawn(_same_width_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    with open(os.path.join(str(tmp_path), "many.pkl"), "rb") as fh:
        bucket = pickle.load(fh)
    gs = [_golden(k) for k in _SAME_WIDTH]
    batch = synth.make_batch(5, 600, 19, np.array(gs[0]["probs"]), synth.seed_for(7))
    for j, g in enumerate(gs):
        ptab = orc.p_table(g["pmf"])
        sc, pv = orc.score_kmers_table(batch.kmers, np.array(g["score_matrix"], dtype=np.int64), ptab, g["min_val"])
        q = orc.fdr_bh(pv)
        exp_rows = np.nonzero(q < 0.2)[0]
        rows = np.concatenate([bucket[r][j][0] for r in range(2)])
        qs = np.concatenate([bucket[r][j][1] for r in range(2)])
        assert np.array_equal(rows, exp_rows) and len(rows) > 0
        np.testing.assert_allclose(qs, q[exp_rows], rtol=1e-12, atol=0)


# ---------------------------------------------------------------------------------------------------------------------
# the top-hit-only gather (north_star: "RCCL ... only for the final top-hit gather"; grafimo_amd/top_hits.py)
def _top_worker(rank, world, port, seqdir, kw, top_n, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import pickle
    import torch.distributed as dist
    from grafimo_amd import distributed as D
    from grafimo_amd.motif import Motif
    from grafimo_amd.top_hits import compute_top_regions_sharded
    from grafimo_amd.workflow import Findmotif
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = _golden_ctcf()
        m = Motif(np.array(g["probs"]), 19, ["A", "C", "G", "T"], g["motif_id"], g["motif_name"],
                  {n: i for i, n in enumerate("ACGT")})
        stats, moved = {}, []
        real = D.gather_columns

        def counting(cols, device, group=None):
            if "rows" in cols:
                moved.append(len(cols["rows"]))
            return real(cols, device, group)
        D.gather_columns = counting
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                top = compute_top_regions_sharded(m, seqdir, True, Findmotif(**kw), top_graphs=top_n,
                                                  backend=_make_backend(g), stats=stats)
                full = D.compute_results_sharded(m, seqdir, True, Findmotif(**kw), backend=_make_backend(g))
        finally:
            D.gather_columns = real
        with open(os.path.join(outdir, f"top{rank}.pkl"), "wb") as fh:
            pickle.dump(dict(top=top, full=full, stats=stats, moved=moved), fh)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kw", [dict(threshold=5e-2), dict(threshold=0.5, qval_t=True, recomb=True)])
def test_two_ranks_gather_one_hit_per_region_for_the_top_regions(tsv_dir, tmp_path, kw):
    """Every rank keeps the best reported hit of each of its regions and only those travel: the table rank 0 builds
    equals the first-row-per-region of the full sharded report (what --top-graphs walks, res_writer.py:153-157), and
    the rows gathered are at most one per region per rank instead of every hit."""
    import pickle
    import torch.multiprocessing as mp
    from grafimo_amd.top_hits import top_regions, top_regions_table
    top_n = 4
    mp.spawn(_top_worker, args=(2, _free_port(), tsv_dir, kw, top_n, str(tmp_path)), nprocs=2, join=True)
    parts = [pickle.load(open(tmp_path / f"top{r}.pkl", "rb")) for r in range(2)]
    top, full = parts[0]["top"], parts[0]["full"]
    assert parts[1]["top"] is None and parts[1]["full"] is None
    want = top_regions_table(full, top_n)
    assert len(top) == len(want) == top_n and list(top.columns) == list(full.columns)
    for c in full.columns:       # ties in p-value can only come from equal scores: compare as sets of rows per column
        assert (top[c].astype(str) == want[c].astype(str)).all(), c
    assert list(top["sequence_name"]) == top_regions(full, top_n)
    for p in parts:
        # first gather = the reduced one (one row per region of this rank), second = the full report's
        assert p["moved"][0] == p["stats"]["sent"] <= 7 and p["moved"][1] == p["stats"]["hits"]
        assert p["stats"]["sent"] < p["stats"]["hits"]
    assert sum(p["stats"]["sent"] for p in parts) == full["sequence_name"].nunique()


# ------------------------------------------------------------------------------------------------ sharded GRAPH
def _graph_shard_worker(rank, world, port, workdir):
    """every rank: its contiguous share of the regions, the SHARD of the graph those regions can meet (shard_index), the
    oracle's rows over that shard, their score histogram -> all-reduce; rank 0 compares with one process over the whole graph"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from extract_helpers import variants_from_index
    from grafimo_amd.distributed import shard_bounds
    from grafimo_amd.extract_regions import GraphIndex, shard_index
    from oracle import extract_oracle as xo
    from oracle import oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = _golden_ctcf()
    sm, ptab = np.array(g["score_matrix"], dtype=np.int64), orc.p_table(g["pmf"])
    idx = GraphIndex.load(os.path.join(workdir, "g.gfmidx.npz"))
    regions = np.load(os.path.join(workdir, "regions.npy"))
    lo, hi = shard_bounds(len(regions), world, rank)
    mine = regions[lo:hi]
    sub = shard_index(idx, mine[:, 0], mine[:, 1]) if len(mine) else None
    ref = idx.ref.tobytes()

    def rows_of(index, regs):
        v = variants_from_index(index)
        out = []
        for s, e in regs.tolist():
            out += xo.enumerate_region_variants("7", ref, v, s, e, 19, with_counts=True)
        return out

    rows = rows_of(sub, mine) if sub is not None else []
    km = np.frombuffer("".join(r[1] for r in rows).encode(), dtype=np.uint8).reshape(-1, 19)
    sc = orc.score_kmers_table(km, sm, ptab, g["min_val"])[0] if len(km) else np.empty(0, np.int64)
    hist = torch.from_numpy(np.bincount(sc, minlength=19001).astype(np.int64))
    n_sites = torch.tensor([len(sub.pos) if sub is not None else 0], dtype=torch.int64)
    dist.all_reduce(hist)                                   # the one data-path exchange of the sharded graph path
    per_rank = [torch.zeros_like(n_sites) for _ in range(world)]
    dist.all_gather(per_rank, n_sites)
    counts = [None] * world
    dist.all_gather_object(counts, [(r[0], r[1], r[2], r[3], r[4], r[5]) for r in rows])
    if rank == 0:
        full = rows_of(idx, regions)
        km_f = np.frombuffer("".join(r[1] for r in full).encode(), dtype=np.uint8).reshape(-1, 19)
        exp = np.bincount(orc.score_kmers_table(km_f, sm, ptab, g["min_val"])[0], minlength=19001)
        ok = bool(np.array_equal(hist.numpy(), exp)) and [tuple(r) for part in counts for r in part] == [tuple(r[:6]) for r in full]
        sites = [int(x.item()) for x in per_rank]
        with open(os.path.join(workdir, f"result_{world}.txt"), "w") as fh:
            fh.write(f"{ok} {len(full)} {len(idx.pos)} {' '.join(map(str, sites))}\n")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_hold_shards_of_the_graph_and_agree_with_one_process(tmp_path, world):
    """the graph shards with the regions.  gloo world 2 / 4: every rank cuts the part of the graph its regions
    can meet out of the host index (shard_index), enumerates and scores its rows there, the histograms are all-reduced: the
    global histogram and the concatenated rows equal one process over the whole graph, and no rank holds more than its share
    (plus margins) of the site records."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from extract_helpers import make_graph_files
    from grafimo_amd.extract_regions import GraphIndex
    fasta, vcf = make_graph_files(str(tmp_path), chrom="7", length=9000, n_sites=700, n_samples=16, seed=55, rich=True)
    idx = GraphIndex.from_fasta_vcf(fasta, vcf, "7")
    idx.save(str(tmp_path / "g"))
    regions = np.array([(200 + 1000 * i, 420 + 1000 * i) for i in range(8)] + [(8800, 9000)], dtype=np.int64)
    np.save(str(tmp_path / "regions.npy"), regions)
    mp.spawn(_graph_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ok, n_rows, n_sites, *per_rank = open(tmp_path / f"result_{world}.txt").read().split()
    assert ok == "True" and int(n_rows) > 3000
    per_rank = [int(x) for x in per_rank]
    assert len(per_rank) == world and all(0 < x < int(n_sites) * 0.75 for x in per_rank) and sum(per_rank) < 1.3 * int(n_sites)
