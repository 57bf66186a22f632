"""The host half of the fused graph path, no GPU: gfm_graph_hit_columns / gfm_region_labels (csrc/hit_table.cpp) against the
numpy steps they replace (round 5's _fused_tables: keep filter, lexsort into the TSV rows' order, --recomb filter, stable
sort by p-value -- resultsTmp.py:303-314), the label table, the DataFrame made from final columns, the mapped index file,
and the ROUTING of scan_graph's manifest through every consumer (VERDICT r5 Missing #2)."""
import contextlib
import io
import os
import sys

import numpy as np
import pandas as pd
import pytest

from grafimo_amd import _native as nv
from grafimo_amd import extract_regions as xr


def _records(rng, n, W, n_regions, n_win=4000, dup_scores=True, L=None):
    r = np.zeros(n, dtype=xr.HIT_DTYPE)
    r["region"] = rng.integers(0, n_regions, n)
    wq = rng.choice(n_win * 40, size=n, replace=False)                         # (window, walk * 2 + strand) names a row: unique
    r["w"] = wq // 40
    r["q2"] = wq % 40 + (rng.integers(0, 1 << 35, n) << 6 if not dup_scores else 0)
    L = 1000 * W + 1 if L is None else L
    r["score"] = rng.integers(L // 2, L // 2 + 40 if dup_scores else L, n)     # few distinct scores: ties in p-value everywhere
    r["start"] = rng.integers(0, 1 << 40, n)
    span = np.where(rng.random(n) < 0.8, W, W + rng.integers(1, 9, n))
    r["stop"] = np.where(rng.random(n) < 0.5, r["start"] + span, r["start"] - span)
    r["freq"] = np.where(rng.random(n) < 0.3, 0, rng.integers(1, 5000, n))
    r["qvalue"] = rng.random(n)
    r["strand"] = np.where(rng.random(n) < 0.5, ord("+"), ord("-"))
    r["is_ref"] = rng.random(n) < 0.5
    r["keep"] = rng.random(n) < 0.9
    r["kmer"][:, :W] = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, (n, W))]
    return r


def _reference_columns(ptable, scale, offset, W, parts, entry_of, region_base, recomb, first_per_region):
    """round 5's numpy path, step for step"""
    kept, ekey, gid = [], [], []
    for p, recs in enumerate(parts):
        recs = recs[recs["keep"] != 0]
        kept.append(recs)
        ekey.append(entry_of[p][recs["region"]] if len(recs) else np.empty(0, np.int64))
        gid.append(recs["region"].astype(np.int64) + region_base[p])
    recs, ekey, gid = np.concatenate(kept), np.concatenate(ekey), np.concatenate(gid)
    order = np.lexsort((recs["q2"], recs["w"], ekey))                        # the TSV rows' order
    recs, gid = recs[order], gid[order]
    pv = ptable[recs["score"]]
    idx = np.arange(len(recs)) if recomb else np.nonzero(recs["freq"] > 0)[0]
    idx = idx[np.argsort(pv[idx], kind="stable")]                            # resultsTmp.py:312
    if first_per_region:
        _, first = np.unique(gid[idx], return_index=True)
        idx = idx[np.sort(first)]
    recs, gid = recs[idx], gid[idx]
    return dict(start=recs["start"], stop=recs["stop"], freq=recs["freq"], region=gid,
                logodds=recs["score"].astype(np.float64) / float(scale) + float(W) * offset, pvalue=pv[idx],
                qvalue=recs["qvalue"], strand=(recs["strand"] == ord("-")).astype(np.uint8),
                ref=((recs["is_ref"] != 0) & (np.abs(recs["stop"] - recs["start"]) == W)).astype(np.uint8),
                kmers=np.concatenate([recs["kmer"][:, :W], np.full((len(recs), 1), 10, np.uint8)], axis=1))


@pytest.mark.parametrize("W,n_parts,dup", [(19, 1, True), (8, 3, True), (64, 2, False), (1, 1, True)])
@pytest.mark.parametrize("recomb,first", [(True, False), (False, False), (False, True), (True, True)])
def test_hit_columns_equal_the_numpy_steps(W, n_parts, dup, recomb, first):
    rng = np.random.default_rng(1000 * W + n_parts + 2 * recomb + first)
    L = 1000 * W + 1
    pmf = rng.random(L)
    ptable = np.minimum.accumulate((np.cumsum(pmf[::-1])[::-1] / pmf.sum()))
    ptable[L // 2 + 10:L // 2 + 20] = ptable[L // 2 + 10]                                         # different scores, one p-value: row order decides
    n_regions = [int(rng.integers(1, 400)) for _ in range(n_parts)]
    parts = [_records(rng, int(rng.integers(0, 3000)), W, nr, dup_scores=dup) for nr in n_regions]
    if n_parts == 3:
        parts[1] = parts[1][:0]                                              # a handle without a hit
    entry_of = [np.sort(rng.integers(10 * p, 10 * p + 3, nr)).astype(np.int64) for p, nr in enumerate(n_regions)]
    region_base = np.cumsum([0] + n_regions).astype(np.int64)
    got = xr._hit_columns(ptable, 37, -12.0, W, entry_of, region_base, parts, recomb, first)
    want = _reference_columns(ptable, 37, -12.0, W, parts, entry_of, region_base, recomb, first)
    assert set(got) == set(want)
    for k in want:
        assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k]), k
    if first:
        assert len(np.unique(got["region"])) == len(got["region"])


@pytest.mark.parametrize("recomb,first", [(True, False), (False, False), (False, True)])
def test_a_large_table_is_built_by_several_threads_and_is_the_same_table(recomb, first):
    """A single table of >= 4 096 rows: gfm_graph_hit_columns deals its bucket sorts and its output rows to a few of the
    library's host threads (not with GFM_HITS_FIRST_PER_REGION, whose rows depend on the rows before them) -- the numpy steps'
    table, row for row; 30 000 and 4 096 + a few records, scores with ties and without."""
    rng = np.random.default_rng(17 + 2 * recomb + first)
    for n, dup in ((30_000, True), (4_300, False), (30_000, False)):
        W = 19
        L = 1000 * W + 1
        pmf = rng.random(L)
        ptable = np.minimum.accumulate(np.cumsum(pmf[::-1])[::-1] / pmf.sum())
        ptable[L // 2 + 10:L // 2 + 20] = ptable[L // 2 + 10]
        n_regions = [700, 900]
        parts = [_records(rng, n, W, n_regions[0], n_win=40_000, dup_scores=dup), _records(rng, n // 3, W, n_regions[1], n_win=40_000, dup_scores=dup)]
        parts[0]["keep"] = 1                                                  # (>= 4 096 rows stay whatever the filters drop)
        parts[0]["freq"] = np.maximum(parts[0]["freq"], 1)
        entry_of = [np.sort(rng.integers(10 * p, 10 * p + 3, nr)).astype(np.int64) for p, nr in enumerate(n_regions)]
        region_base = np.cumsum([0] + n_regions).astype(np.int64)
        got = xr._hit_columns(ptable, 37, -12.0, W, entry_of, region_base, parts, recomb, first)
        want = _reference_columns(ptable, 37, -12.0, W, parts, entry_of, region_base, recomb, first)
        assert first or len(got["start"]) >= 4096
        for k in want:
            assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k]), (k, n, dup)


def test_hit_columns_arguments_and_empty_input():
    pt = np.ones(19001)
    c = xr._hit_columns(pt, 62, -14.0, 19, [np.zeros(0, np.int64)], np.zeros(2, np.int64), [np.empty(0, xr.HIT_DTYPE)], True, False)
    assert all(len(v) == 0 for v in c.values()) and c["kmers"].shape == (0, 20)
    bad = np.zeros(1, dtype=xr.HIT_DTYPE)
    bad["keep"], bad["score"] = 1, 19001
    with pytest.raises(nv.NativeError) as e:
        xr._hit_columns(pt, 62, -14.0, 19, [np.zeros(1, np.int64)], np.zeros(2, np.int64), [bad], True, False)
    assert "outside the table" in e.value.msg
    n_out = __import__("ctypes").c_int64()
    assert nv.lib().gfm_graph_hit_columns(None, 1, 1, 0.0, 19, 0, None, None, None, None, 0, n_out, *([None] * 10)) == nv.GFM_ERR_INVALID
    assert nv.lib().gfm_region_labels(None, None, None, 0, None, 0) == nv.GFM_ERR_INVALID


def test_columns_of_a_motif_set_on_the_library_threads_equal_the_single_calls():
    """gfm_graph_hit_columns_start / _wait (one job per motif of a set, the caller free meanwhile) == gfm_graph_hit_columns
    per motif: 23 jobs of different widths, sizes (empty ones among them), flags and part counts -- more jobs than threads --
    several runs one after the other (the threads' scratch is reused), two runs in flight at once; a job that fails says
    which and why while the others finish."""
    rng = np.random.default_rng(77)

    def spec(i):
        W = int(rng.integers(1, 65))
        L = 1000 * W + 1
        pmf = rng.random(L)
        ptable = np.minimum.accumulate(np.cumsum(pmf[::-1])[::-1] / pmf.sum())
        n_parts = int(rng.integers(1, 4))
        n_regions = [int(rng.integers(1, 300)) for _ in range(n_parts)]
        parts = [_records(rng, 0 if i % 7 == 3 else int(rng.integers(0, 4000)), W, nr, dup_scores=bool(i % 2)) for nr in n_regions]
        entry_of = [np.sort(rng.integers(10 * p, 10 * p + 3, nr)).astype(np.int64) for p, nr in enumerate(n_regions)]
        return (ptable, 37 + i, -12.0 - i, W, entry_of, np.cumsum([0] + n_regions).astype(np.int64), parts, bool(i % 3), i % 5 == 4)

    specs = [spec(i) for i in range(23)]
    want = [xr._hit_columns(*sp) for sp in specs]
    for rep in range(3):
        got = xr._ColumnsRun(specs).wait()
        assert len(got) == len(want)
        for g_, w_ in zip(got, want):
            assert set(g_) == set(w_)
            for k in w_:
                assert g_[k].dtype == w_[k].dtype and np.array_equal(g_[k], w_[k]), k
    a, b = xr._ColumnsRun(specs[:9]), xr._ColumnsRun(specs[9:])                  # (the second finds the crew busy: threads of its own)
    for g_, w_ in zip(a.wait() + b.wait(), want):
        assert all(np.array_equal(g_[k], w_[k]) for k in w_)
    assert xr._ColumnsRun([]).wait() == []
    with pytest.raises(RuntimeError):
        a.wait()
    # one bad record in job 4: that job's status and message; the others' columns are complete
    bad = list(specs[4])
    recs = specs[4][6][0].copy() if len(specs[4][6][0]) else _records(rng, 5, specs[4][3], len(specs[4][4][0]))
    recs["keep"][0], recs["score"][0] = 1, len(specs[4][0])
    bad[6] = [recs] + list(specs[4][6][1:])
    run = xr._ColumnsRun(specs[:4] + [tuple(bad)] + specs[5:8])
    with pytest.raises(nv.NativeError) as e:
        run.wait()
    assert "outside the table" in e.value.msg
    assert [int(j.status) for j in run.jobs][:8] == [0, 0, 0, 0, nv.GFM_ERR_INVALID, 0, 0, 0]
    # a run nobody waited for is waited for by close(); the C entry points check their arguments
    r_ = xr._ColumnsRun(specs[:3])
    r_.close()
    assert r_.run is None
    import ctypes
    out = ctypes.c_void_p()
    assert nv.lib().gfm_graph_hit_columns_start(None, 2, ctypes.byref(out)) == nv.GFM_ERR_INVALID
    assert nv.lib().gfm_graph_hit_columns_start(None, 0, None) == nv.GFM_ERR_INVALID
    assert nv.lib().gfm_graph_hit_columns_wait(None) == nv.GFM_ERR_INVALID


def test_region_labels():
    rng = np.random.default_rng(7)
    runs = [("22", rng.integers(0, 1 << 40, 300), rng.integers(0, 1 << 40, 300)), ("chrUn_KI270742v1", np.array([0, 7]), np.array([5, 9])),
            ("x", np.zeros(0, np.int64), np.zeros(0, np.int64)), ("é", np.array([-3]), np.array([2 ** 62]))]
    want = np.array([f"{c}:{int(s)}-{int(e)}" for c, ss, ee in runs for s, e in zip(ss, ee)], dtype=object)
    lab = xr.RegionLabels(runs)
    assert lab.n == len(want) == 303
    few = np.array([302, 0, 300, 300, 17, 301], dtype=np.int64)
    assert lab._all is None and list(lab.take(few)) == list(want[few]) and lab._all is None      # the distinct regions only
    many = rng.integers(0, 303, 200)
    assert list(lab.take(many)) == list(want[many]) and lab._all is not None                     # a fair share: the whole table, once
    assert list(lab.all()) == list(want) and list(lab.take(few)) == list(want[few])
    assert list(xr.RegionLabels([]).take(np.zeros(0, np.int64))) == []


def test_label_tables_are_shared_by_calls_over_the_same_regions(monkeypatch):
    """RegionLabels.shared: GRAFIMO's loop makes one call per motif over one set of regions; the calls share ONE label table
    (compared by content, not by identity), a different region set gets its own, four are kept, a caller that changes its
    arrays afterwards does not change a kept table."""
    monkeypatch.setattr(xr.RegionLabels, "_kept", [])
    s_, e_ = np.arange(1000, dtype=np.int64) * 300, np.arange(1000, dtype=np.int64) * 300 + 200
    a = xr.RegionLabels.shared([("chr1", s_, e_)])
    b = xr.RegionLabels.shared([("chr1", s_.copy(), e_.copy())])
    assert a is b
    ids = np.array([5, 999, 5, 0])
    assert list(a.take(ids)) == ["chr1:1500-1700", "chr1:299700-299900", "chr1:1500-1700", "chr1:0-200"]
    assert xr.RegionLabels.shared([("chr2", s_, e_)]) is not a                      # another name
    assert xr.RegionLabels.shared([("chr1", s_, e_ + 1)]) is not a                  # other coordinates
    assert xr.RegionLabels.shared([("chr1", s_[:-1], e_[:-1])]) is not a            # fewer regions
    assert xr.RegionLabels.shared([("chr1", s_, e_)]) is a and xr.RegionLabels._kept[0] is a
    s_[5] = 7                                                                       # the caller's array, not the table's
    assert a.take(np.array([5]))[0] == "chr1:1500-1700"
    assert xr.RegionLabels.shared([("chr1", s_, e_)]) is not a
    assert len(xr.RegionLabels._kept) == 4 and a in xr.RegionLabels._kept
    for k in range(4):
        xr.RegionLabels.shared([(f"x{k}", s_, e_)])
    assert a not in xr.RegionLabels._kept and len(xr.RegionLabels._kept) == 4


def test_frame_of_columns_is_the_table_build_frame_makes():
    from grafimo_amd.resultsTmp import build_frame
    from grafimo_amd.motif import Motif
    rng = np.random.default_rng(3)
    W = 12
    recs = _records(rng, 500, W, 40)
    pt = np.linspace(1, 0, 12001)
    c = xr._hit_columns(pt, 50, -3.0, W, [np.zeros(40, np.int64)], np.zeros(2, np.int64), [recs], False, False)
    lab = xr.RegionLabels([("7", np.arange(40) * 100, np.arange(40) * 100 + 50)])

    class M:
        motif_id, motif_name = "MA0001.1", "AGL3"
    for no_q in (False, True):
        df = xr._frame_of_columns(M, c, lab.take(c["region"]), no_q)
        want = build_frame(M, seqnames=[f"7:{100 * r}-{100 * r + 50}" for r in c["region"]], starts=c["start"], stops=c["stop"],
                           strands=["-" if s else "+" for s in c["strand"]], scores=c["logodds"], pvalues=c["pvalue"],
                           qvalues=None if no_q else c["qvalue"], seqs=[bytes(k[:W]).decode() for k in c["kmers"]],
                           frequencies=c["freq"], references=["ref" if r else "non.ref" for r in c["ref"]], recomb=False)
        pd.testing.assert_frame_equal(df, want)
        assert df["matched_sequence"].map(type).eq(str).all() and df["sequence_name"].map(type).eq(str).all()


def test_frame_from_final_columns_is_the_public_constructors_frame(monkeypatch):
    """_frame_from_final_columns goes through pandas' column-arrays -> block-manager step (an internal): the frame must be the
    public constructor's -- values, dtypes, labels -- and behave like one (a new column, a cell written, sorted, written out,
    empty); anything else than final columns, or a pandas whose internal differs, falls back to the public constructor."""
    rng = np.random.default_rng(5)
    for n in (0, 1, 253):
        obj = np.empty(n, dtype=object)
        obj[:] = [f"s{i}" for i in range(n)]
        data = {"motif_id": obj, "start": rng.integers(0, 1 << 40, n), "score": rng.random(n), "strand": xr._STRAND_OBJ[rng.integers(0, 2, n)],
                "p-value": rng.random(n), "haplotype_frequency": rng.integers(0, 5000, n)}
        monkeypatch.setattr(xr, "_FAST_FRAME", None)
        first = xr._frame_from_final_columns(data)              # (the first table of a process: built both ways and compared)
        assert xr._FAST_FRAME is (True if n else None)           # (... the first with rows in it: an empty one proves nothing)
        monkeypatch.setattr(xr, "_FAST_FRAME", True)
        df = xr._frame_from_final_columns(data)
        ref = pd.DataFrame(data, copy=False)
        for got in (first, df):
            pd.testing.assert_frame_equal(got, ref)
            assert type(got) is pd.DataFrame and list(got.dtypes) == list(ref.dtypes)
        if n:
            df["extra"] = 1
            df.loc[0, "start"] = 7
            assert df.loc[0, "start"] == 7 and df.shape == (n, 7)
            assert df.sort_values("p-value").iloc[0]["p-value"] == data["p-value"].min()
            assert df.drop(columns="extra").to_csv(sep="\t") == ref.assign(start=df["start"]).to_csv(sep="\t")
            assert df[df["strand"] == "+"]["motif_id"].tolist() == [x for x, s_ in zip(obj, data["strand"]) if s_ == "+"]
    # not final columns (a list, a 2-D array, ragged lengths): the public constructor, as it would behave
    for odd in ({"a": [1, 2, 3], "b": np.arange(3)}, {"a": np.arange(3), "b": np.arange(3).astype(object).reshape(3, 1)[:, 0][::-1]}):
        pd.testing.assert_frame_equal(xr._frame_from_final_columns(odd), pd.DataFrame(odd))
    with pytest.raises(ValueError):
        xr._frame_from_final_columns({"a": np.arange(3), "b": np.arange(4)})
    assert xr._FAST_FRAME is True
    # a pandas without that internal: the same
    import pandas.core.internals.managers as mgrs
    monkeypatch.setattr(xr, "_FAST_FRAME", None)
    monkeypatch.delattr(mgrs, "create_block_manager_from_column_arrays")
    pd.testing.assert_frame_equal(xr._frame_from_final_columns(data), pd.DataFrame(data))
    assert xr._FAST_FRAME is False


def test_index_file_is_mapped_not_read(tmp_path):
    """GraphIndex.save writes a .npz numpy can read whose members are stored and 64-byte aligned; load() maps it."""
    from grafimo_amd import synth
    idx, _ = synth.make_graph_index(60, 19, n_haplotypes=200)
    p = idx.save(str(tmp_path / "c"))
    j = xr.GraphIndex.load(p)
    for k in ("ref", "pos", "del_len", "n_alts", "alt_bases", "ins_len", "ins_off", "ins_bases", "alt_bits"):
        a, b = getattr(idx, k), getattr(j, k)
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), k
        assert b.flags.aligned and b.flags.c_contiguous and (b.size == 0 or b.ctypes.data % 64 == 0), k
    assert not j.alt_bits.flags.writeable and j._mapping is not None            # views of the file's pages
    assert (j.chrom, j.n_haplotypes, j.skipped) == (idx.chrom, idx.n_haplotypes, idx.skipped)
    with np.load(p) as z:                                                        # any numpy reads it
        assert np.array_equal(z["alt_bits"], idx.alt_bits) and str(z["chrom"]) == idx.chrom
    pc = idx.save(str(tmp_path / "small"), compressed=True)                      # the deflated form (rounds 1-5) still loads
    assert os.path.getsize(pc) < os.path.getsize(p)
    k = xr.GraphIndex.load(pc)
    assert np.array_equal(k.alt_bits, idx.alt_bits) and np.array_equal(k.ref, idx.ref) and k._mapping is None
    # saving over a file that is mapped (the index loaded from this very path) must not pull the pages from under the reader
    again = xr.GraphIndex.load(p)
    assert again.save(p) == p and np.array_equal(again.alt_bits, idx.alt_bits) and np.array_equal(j.ref, idx.ref)
    assert np.array_equal(xr.GraphIndex.load(p).alt_bits, idx.alt_bits) and not [f for f in os.listdir(tmp_path) if f.endswith(".tmp")]
    # a file that is no index (truncated, another .npz, not a zip at all): the package's own error, named
    from grafimo_amd.grafimo_errors import VGError
    raw = open(p, "rb").read()
    for name, blob in (("cut", raw[:len(raw) // 2]), ("junk", b"not a zip" * 100), ("empty", b"")):
        bad = tmp_path / f"{name}.gfmidx.npz"
        bad.write_bytes(blob)
        with pytest.raises(VGError, match="is not a graph index"):
            xr.GraphIndex.load(str(bad))
    np.savez(str(tmp_path / "other.gfmidx.npz"), a=np.arange(3))
    with pytest.raises(VGError, match="is not a graph index"):
        xr.GraphIndex.load(str(tmp_path / "other.gfmidx.npz"))
    no_bits = xr.GraphIndex("1", idx.ref, idx.pos, idx.n_alts, idx.alt_bases, None, 0)
    assert xr.GraphIndex.load(no_bits.save(str(tmp_path / "nb"))).alt_bits is None


# ---------------------------------------------------------------------------------------------- routing of the manifest
class _Motif:
    def __init__(self, mid, w):
        from grafimo_amd.motif import MOTIF_FIELDS
        for f in MOTIF_FIELDS:                       # (is_motif_like: the members a Motif has; the device half is mocked here)
            setattr(self, f, None)
        self.motif_id, self.motif_name, self.width = mid, mid.lower(), w


@pytest.fixture
def manifest_dir(tmp_path, monkeypatch):
    from grafimo_amd.workflow import Findmotif
    ref = np.frombuffer(b"ACGT" * 100, dtype=np.uint8)
    xr.GraphIndex("1", ref, [10, 50], [1, 1], np.array([[67, 0, 0], [71, 0, 0]], np.uint8), None, 0).save(str(tmp_path / "chr1"))
    bed = tmp_path / "r.bed"
    bed.write_text("chr1\t0\t100\nchr1\t200\t390\n")
    wf = Findmotif(graph_genome_dir=str(tmp_path), bedfile=str(bed), chroms=["1"], chroms_prefix="chr", threshold=0.5)
    monkeypatch.setenv("GRAFIMO_SCAN_OUTPUT", "manifest")
    with contextlib.redirect_stdout(io.StringIO()):
        loc = xr.scan_graph({15, 8}, wf, True)
    # the device half replaced by a recorder: what reaches the fused path, and a table per motif
    calls = []

    def fake_prep(manifest, group=None):
        return ("prep", tuple(e["chrom"] for e in manifest["entries"]), group)

    class FakePass:                       # xr._FusedPass: enqueue -> fetch -> tables -> close; a motif set: enqueue -> fetch ->
        running = []                      # columns_start -> columns_wait -> frames -> close (native columns in flight: ONE pass's)

        def __init__(self, motifs, prep, debug, args_obj, top_graphs):
            self.motifs, self.state = motifs, []
            calls.append(([m.motif_id for m in motifs], prep, top_graphs))

        def enqueue(self):
            self.state.append("enqueue")

        def fetch(self):
            assert self.state == ["enqueue"]
            # the records are read in place from ONE page-locked buffer: nobody's columns may be running when a fetch writes there
            assert not FakePass.running
            self.state.append("fetch")

        def _frames(self):
            return [pd.DataFrame({"motif_id": [m.motif_id], "width": [m.width]}) for m in self.motifs]

        def tables(self):
            assert self.state == ["enqueue", "fetch"]
            return self._frames()

        def columns_start(self):
            assert self.state == ["enqueue", "fetch"]
            FakePass.running.append(self)
            self.state.append("start")

        def columns_wait(self):
            assert self.state == ["enqueue", "fetch", "start"]
            FakePass.running.remove(self)
            self.state.append("wait")

        def frames(self):
            assert self.state == ["enqueue", "fetch", "start", "wait"]
            return self._frames()

        def close(self):
            self.state.append("close")

    monkeypatch.setattr(xr, "_manifest_prep", fake_prep)
    monkeypatch.setattr(xr, "_FusedPass", FakePass)
    yield loc, wf, calls
    import shutil
    shutil.rmtree(loc, ignore_errors=True)


def test_every_consumer_understands_the_manifest(manifest_dir):
    """VERDICT r5 Missing #2 / Weak #2: compute_results_many and the sharded entry points globbed width_W/*.tsv in a manifest
    directory, found nothing and said "No result retrieved".  grafimo.findmotif scores EVERY motif of the set over one
    scan_graph result (grafimo.py:176-183): the set goes through the graph, the motifs of a width in one pass."""
    from grafimo_amd import distributed as dd
    from grafimo_amd import score_sequences as ss
    loc, wf, calls = manifest_dir
    assert sorted(os.listdir(loc)) == [xr.MANIFEST_NAME, "width_15", "width_8"]
    motifs = [_Motif("A", 15), _Motif("B", 8), _Motif("C", 15), _Motif("D", 15), _Motif("E", 15)]
    with contextlib.redirect_stdout(io.StringIO()):
        single = [ss.compute_results(m, loc, True, wf) for m in motifs]
    assert [c[0] for c in calls] == [["A"], ["B"], ["C"], ["D"], ["E"]]
    del calls[:]
    for fn in (ss.compute_results_many, dd.compute_results_many_sharded):
        with contextlib.redirect_stdout(io.StringIO()):
            tabs = fn(motifs, loc, True, wf)
        assert [c[0] for c in calls] == [["A", "C", "D", "E"], ["B"]]        # per width ONE fused call (groups of three inside it)
        assert all(c[1] == ("prep", ("1",), None) for c in calls)
        for a, b in zip(tabs, single):                                         # == per-motif calls, in the order of `motifs`
            pd.testing.assert_frame_equal(a, b)
        del calls[:]
    with contextlib.redirect_stdout(io.StringIO()):
        one = dd.compute_results_sharded(motifs[1], loc, True, wf)
    pd.testing.assert_frame_equal(one, single[1])
    # a width scan_graph was not asked for: the reference's own message (score_sequences.py:189-192), from every consumer
    for call in (lambda: ss.compute_results(_Motif("Z", 9), loc, True, wf), lambda: ss.compute_results_many([motifs[0], _Motif("Z", 9)], loc, True, wf),
                 lambda: dd.compute_results_many_sharded([_Motif("Z", 9)], loc, True, wf)):
        with pytest.raises(ValueError) as e, contextlib.redirect_stdout(io.StringIO()):
            call()
        assert "No result retrieved" in str(e.value)


def test_the_command_line_scores_a_motif_set_in_one_call(manifest_dir, tmp_path, monkeypatch):
    """`python -m grafimo_amd -d DIR -b BED -m FILE(s)` with more than one motif: ONE compute_results_many call over the
    manifest its own scan_graph left (the CLI held the whole list and still called compute_results per motif)."""
    from grafimo_amd import __main__ as cli
    loc, wf, calls = manifest_dir
    got = {}
    monkeypatch.setattr(cli, "get_motif_pwm", lambda f, *a, **k: [_Motif(os.path.basename(f) + "1", 15), _Motif(os.path.basename(f) + "2", 8)])
    monkeypatch.setattr(cli, "print_results", lambda res, debug: got.setdefault("tables", []).append(res))
    monkeypatch.delenv("GRAFIMO_SCAN_OUTPUT", raising=False)                     # auto: the CLI's module holds our consumers
    with contextlib.redirect_stdout(io.StringIO()):
        cli.main(["-d", str(wf.graph_genome_dir), "-b", wf.bedfile, "-m", "x.meme", "y.meme", "--chroms-prefix-find", "chr",
                  "--chroms-find", "1", "--text-only", "-j", "1"])
    assert [c[0] for c in calls] == [["x.meme1", "y.meme1"], ["x.meme2", "y.meme2"]]
    assert [t["motif_id"][0] for t in got["tables"]] == ["x.meme1", "x.meme2", "y.meme1", "y.meme2"]
