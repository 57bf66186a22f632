"""Host-side pieces that need no GPU: C-ABI surface, TSV ingest, result table rules."""
import ctypes
import os
import re

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN, REF_DATA, ROOT
from grafimo_amd import _native as nv
from grafimo_amd.motif import Motif
from grafimo_amd.resultsTmp import ResultTmp
from grafimo_amd.score_sequences import KmerTable, compute_qvalues, compute_results
from grafimo_amd.workflow import Findmotif
from oracle import oracle as orc

TSV = os.path.join(REF_DATA, "width_19", "scoring_test_input.tsv")


def test_library_loads_and_exports_every_declared_symbol():
    """Every function declared in include/grafimo_hip.h is exported by the .so and bound."""
    header = open(os.path.join(ROOT, "include", "grafimo_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(gfm_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    lib = nv.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared but not exported"
        assert name in nv.PROTOTYPES, f"{name} has no ctypes prototype"
    assert set(nv.PROTOTYPES) == declared
    assert lib.gfm_abi_version() == nv.ABI_VERSION == 12
    # importing / loading must not have initialised a device; counting devices is allowed
    assert nv.device_count() >= 0


# timing-only switches: compiled with -DGFM_LAB alone
LAB_ONLY_KNOBS = (b"GRAFIMO_FUSED_LAB", b"GRAFIMO_FUSED_TIMERS", b"GRAFIMO_FUSED_SPLIT", b"GRAFIMO_FUSED_BESIDE",
                  b"GRAFIMO_FUSED_SMALL_MASS", b"GRAFIMO_FUSED_WAVES", b"GRAFIMO_SCORE_GRID", b"GRAFIMO_SCORE_WAVES",
                  b"GRAFIMO_STORE_POLICY", b"GRAFIMO_EXTRACT_SERIAL", b"GRAFIMO_PARSE_THREADS_EXACT", b"GRAFIMO_SCAN_TRACE")
# what libgrafimo_hip.so reads: product settings, then test aids (they pick a code path, never a result)
PRODUCT_KNOBS = ("GRAFIMO_RESERVE_CUS", "GRAFIMO_SCAN_KEEP_BYTES",
                 "GRAFIMO_PLAN_MAX_WALKS", "GRAFIMO_EXTRACT_DEL_POOL", "GRAFIMO_SCAN_TEXT_BYTES", "GRAFIMO_SCAN_NO_AVX512",
                 "GRAFIMO_FUSED_WALK_CACHE_BYTES")


def test_product_library_holds_no_lab_switch():
    """the switches that turn parts of graph_score_kernel off (results wrong) and the kernels' phase timers
    exist in lab builds only (scripts/lab_build.sh -DGFM_LAB): the product library does not even contain their names."""
    blob = open(nv.LIB_PATH, "rb").read()
    for name in LAB_ONLY_KNOBS:
        assert name not in blob, name
    # ... and every variable the product library does read is one INTEGRATION.md lists (settings and test aids)
    import re
    read = set(re.findall(rb"GRAFIMO_[A-Z0-9_]+", blob))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert read == {k.encode() for k in PRODUCT_KNOBS}, sorted(read ^ {k.encode() for k in PRODUCT_KNOBS})
    for k in PRODUCT_KNOBS:
        assert k in doc, k + " is read by the library and not documented in INTEGRATION.md"
    src = open(os.path.join(ROOT, "grafimo_amd", "csrc", "graph_extract.hip")).read()
    at = src.index('getenv("GRAFIMO_FUSED_LAB")')
    assert "#ifdef GFM_LAB" in src[max(0, at - 1200):at] and "#endif" in src[at:at + 400]


def test_no_cpu_fallback_without_gpu(golden_motifs):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from grafimo_amd.device import DeviceMotif
    _, flat = golden_motifs
    m = flat["ctcf_meme_unif#0"]
    with pytest.raises(nv.NativeError) as e:
        DeviceMotif(m["score_matrix"], m["bg"], m["min_val"], m["scale"], m["offset"])
    assert e.value.code == nv.GFM_ERR_NODEVICE


def test_abi_argument_validation():
    lib = nv.lib()
    out = np.zeros(8)
    assert lib.gfm_compute_log_odds(None, 2, None, None) == nv.GFM_ERR_INVALID
    probs = np.array([[0.5, 0.5], [0.25, 0.0], [0.125, 0.25], [0.125, 0.25]])
    bg = np.full(4, 0.25)
    rc = lib.gfm_compute_log_odds(nv.ptr(probs), 2, nv.ptr(bg), nv.ptr(out))
    assert rc == nv.GFM_ERR_ASSERT and b"prob > 0" in lib.gfm_last_error()
    bad_bg = np.array([0.25, 0.25, 0.25, 0.0])
    probs[1, 1] = 0.25
    assert lib.gfm_compute_log_odds(nv.ptr(probs), 2, nv.ptr(bad_bg), nv.ptr(out)) == nv.GFM_ERR_ASSERT
    h = ctypes.c_void_p()
    sm = np.zeros((4, 70), dtype=np.int64)
    assert lib.gfm_motif_create(nv.ptr(sm), 70, nv.ptr(bg), 0, 1, 0.0, None, ctypes.byref(h)) == nv.GFM_ERR_INVALID
    assert b"width" in lib.gfm_last_error()


def test_tsv_ingest_matches_reference_row_handling():
    t = KmerTable([TSV], 19, False, 2)
    cols = orc.parse_tsv_rows([TSV])
    assert t.n == 704 and len(t.names) == 1 and t.names[0] == cols["seqname"][0]
    assert [bytes(k).decode() for k in t.kmers] == cols["seq"]
    assert list(t.start) == cols["start"] and list(t.stop) == cols["stop"]
    assert [chr(c) for c in t.strand] == cols["strand"]
    assert list(t.freq) == cols["freq"]
    ref = np.array(cols["ref"], dtype=object)
    indel = np.abs(np.array(cols["stop"]) - np.array(cols["start"])) != 19
    ref[(ref == "ref") & indel] = "non.ref"           # score_sequences.py:305-307
    assert list(t.is_ref) == [int(r == "ref") for r in ref]
    # --no-reverse drops '-' rows before they are counted (score_sequences.py:281-282)
    t2 = KmerTable([TSV], 19, True, 1)
    assert t2.n == sum(1 for s in cols["strand"] if s == "+")
    assert set(chr(c) for c in t2.strand) == {"+"}


def test_tsv_ingest_edge_cases(tmp_path):
    d = tmp_path / "width_5"
    d.mkdir()
    good = d / "a.tsv"
    good.write_text(
        "chr1:10-30\tACGTN\tchr1:10+\tchr1:15+\t7\tref\t1+,\n"
        "\n"                                               # blank line tolerated
        "chr1:10-30\tacgtn\tchr1:20-\tchr1:15-\t0\tnon.ref\t2-,3-,\n"
        "chr1:10-30  TTTTT  chr1:11+  chr1:17+  3  ref  9+,")   # spaces, indel ref, no final newline
    (d / "empty.tsv").write_text("")
    other = d / "b.tsv"
    other.write_text("chr2:5-9\tGGGGG\tchr2:5+\tchr2:10+\t1\tref\t4+,\n")
    t = KmerTable([str(good), str(d / "empty.tsv"), str(other)], 5, False, 3)
    assert t.n == 4
    assert [bytes(k).decode() for k in t.kmers] == ["ACGTN", "acgtn", "TTTTT", "GGGGG"]
    assert list(t.is_ref) == [1, 0, 0, 1]                  # 3rd row: ref but |stop-start| != W
    assert t.names == ["chr1:10-30", "chr2:5-9"] and list(t.name_id) == [0, 0, 0, 1]
    assert list(t.start) == [10, 20, 11, 5] and list(t.stop) == [15, 15, 17, 10]
    bad = d / "bad.tsv"
    bad.write_text("chr1:1-2\tACG\tchr1:1+\tchr1:4+\t1\tref\t1+,\n")
    with pytest.raises(nv.NativeError) as e:
        KmerTable([str(bad)], 5, False, 1)
    assert e.value.code == nv.GFM_ERR_IO and "k-mer length" in e.value.msg
    with pytest.raises(nv.NativeError):
        KmerTable([str(d / "missing.tsv")], 5, False, 1)
    assert KmerTable([], 5, False, 1).n == 0


def test_compute_qvalues_matches_reference(golden_json, capsys):
    for case in golden_json("bh.json"):
        q = compute_qvalues(list(case["p"]), True)
        assert np.array_equal(np.array(q), np.array(case["q"]))
    assert "Computing q-values" in capsys.readouterr().out
    with pytest.raises(TypeError):
        compute_qvalues(np.array([0.1]), True)


def _fake_motif():
    m = Motif(np.ones((4, 19)), 19, ["A", "C", "G", "T"], "MA0139.1", "CTCF",
              {n: i for i, n in enumerate("ACGT")})
    return m


@pytest.mark.parametrize("name", ["default_t1e-2", "qvalt_t0.6", "noqvalue_t5e-3", "recomb_t1",
                                  "norecomb_t1"])
def test_resulttmp_to_df_rules(golden_motifs, golden_json, name):
    """ResultTmp.to_df on the oracle's full (unthresholded) columns reproduces the reference
    table: strict <, q- or p-threshold, --recomb filter (resultsTmp.py:303-310)."""
    _, flat = golden_motifs
    g = flat["ctcf_meme_unif#0"]
    case = golden_json("compute_results.json")[name]
    kw = case["kwargs"]
    cols = orc.parse_tsv_rows([TSV])
    km = np.frombuffer("".join(cols["seq"]).encode(), dtype=np.uint8).reshape(-1, 19)
    _, lo, pv = orc.score_kmers(km, g["score_matrix"], g["pmf"], g["min_val"], g["scale"], g["offset"])
    ref = np.array(cols["ref"], dtype=object)
    ref[(ref == "ref") & (np.abs(np.array(cols["stop"]) - np.array(cols["start"])) != 19)] = "non.ref"
    r = ResultTmp()
    assert r.isempty()
    r.append_list(cols["seqname"], cols["seq"], cols["chrom"], cols["start"], cols["stop"],
                  cols["strand"], list(lo), list(pv), cols["freq"], list(ref))
    assert not r.isempty() and r.size() == 704
    if not kw.get("no_qvalue", False):
        r.add_qvalues(list(orc.fdr_bh(pv)))
    df = r.to_df(_fake_motif(), float(kw.get("threshold", 1e-4)), kw.get("qval_t", False),
                 kw.get("recomb", False), ignore_qvals=kw.get("no_qvalue", False))
    exp = pd.DataFrame(case["df"]["rows"], columns=case["df"]["columns"])
    assert list(df.columns) == list(exp.columns)
    key = ["p-value", "start", "stop", "strand"]
    a = df.sort_values(key).reset_index(drop=True)
    b = exp.sort_values(key).reset_index(drop=True)
    assert len(a) == len(b)
    for c in exp.columns:
        if b[c].dtype.kind == "f":
            assert np.array_equal(a[c].to_numpy(float), b[c].to_numpy(float)), c
        else:
            assert (a[c].astype(str) == b[c].astype(str)).all(), c
    assert (np.diff(df["p-value"].to_numpy()) >= 0).all()
    with pytest.raises(TypeError):
        r.to_df(_fake_motif(), 1, False, True)          # threshold must be float
    with pytest.raises(TypeError):
        r.append_list(*([np.zeros(1)] * 10))


def test_compute_results_argument_errors(tmp_path):
    m = _fake_motif()
    with pytest.raises(TypeError):
        compute_results("not a motif", str(tmp_path), True, Findmotif())
    with pytest.raises(FileNotFoundError):
        compute_results(m, str(tmp_path / "nope"), True, Findmotif())
    with pytest.raises(TypeError):
        compute_results(m, str(tmp_path), True, object())
    with pytest.raises(SystemExit) as e:                  # debug=False: message + exit(1)
        compute_results(m, str(tmp_path / "nope"), False, Findmotif())
    assert e.value.code == 1
    # no TSV for this width: the reference's "No result retrieved" ValueError
    (tmp_path / "width_19").mkdir()
    with pytest.raises(ValueError) as e:
        compute_results(m, str(tmp_path), True, Findmotif())
    assert "No result retrieved" in str(e.value)


def test_tsv_ingest_property_random_rows(tmp_path):
    """Randomised rows (hypothesis): the C++ ingest and the Python restatement of the reference's
    row handling agree on every column, for any mix of strands, ref flags, widths and spacing."""
    from hypothesis import given, settings, strategies as st

    row = st.tuples(
        st.text(alphabet="ACGTNacgt", min_size=7, max_size=7),
        st.integers(0, 10**9), st.integers(-30, 30), st.sampled_from("+-"), st.integers(0, 6000),
        st.sampled_from(["ref", "non.ref"]), st.sampled_from(["\t", "  ", " \t "]),
        st.sampled_from(["chr1:5-90", "22:100-300", "x:0-20"]))

    @settings(max_examples=40, deadline=None)
    @given(st.lists(row, min_size=0, max_size=30), st.booleans())
    def check(rows, skip_rev):
        d = tmp_path / "width_7"
        d.mkdir(exist_ok=True)
        f = d / "r.tsv"
        with open(f, "w") as fh:
            for k, (seq, start, delta, strand, freq, ref, sep, region) in enumerate(rows):
                chrom = region.split(":")[0]
                stop = start + delta
                lead = ["", " ", "\t"][k % 3] if sep != "\t" else ""          # leading white space (line.strip().split())
                fh.write(lead + sep.join([region, seq, f"{chrom}:{start}{strand}", f"{chrom}:{stop}{strand}",
                                          str(freq), ref, f"1{strand},2{strand},"]) + "\n")
        t = KmerTable([str(f)], 7, skip_rev, 1)
        cols = orc.parse_tsv_rows([str(f)], no_reverse=skip_rev)
        assert t.n == len(cols["seq"])
        counted = ctypes.c_int64(-1)                  # the streamed scan's counting pass says the same
        assert nv.lib().gfm_tsv_count_rows(str(f).encode(), int(skip_rev), ctypes.byref(counted)) == 0
        assert counted.value == t.n
        assert [bytes(k).decode() for k in t.kmers] == cols["seq"]
        assert list(t.start) == cols["start"] and list(t.stop) == cols["stop"]
        assert [chr(c) for c in t.strand] == cols["strand"] and list(t.freq) == cols["freq"]
        exp_ref = [int(r == "ref" and abs(b - a) == 7)
                   for r, a, b in zip(cols["ref"], cols["start"], cols["stop"])]
        assert list(t.is_ref) == exp_ref
        assert [t.names[i] for i in t.name_id] == cols["seqname"]

    check()


def test_reference_shaped_motif_is_accepted_by_the_boundary_helpers(golden_motifs):
    """Duck-typed boundary: an object with the reference Motif's members --
    rows in its own nucsmap order, bg as a dict -- maps onto the dense C-ABI layout."""
    from grafimo_amd.motif import Motif, dense_bg, dense_score_matrix, is_motif_like
    from ref_shapes import RefShapedMotif
    _, flat = golden_motifs
    rec = flat["ctcf_meme_unif#0"]
    m = RefShapedMotif(rec, with_pmf=False)
    assert not isinstance(m, Motif) and is_motif_like(m)
    assert not hasattr(m, "dense_score_matrix")
    assert np.array_equal(dense_score_matrix(m), rec["score_matrix"])
    assert np.array_equal(dense_bg(m), rec["bg"])
    assert not is_motif_like(object()) and not is_motif_like("MA0139.1")


def _read_table(files):
    t = KmerTable(files, 19, False, 8)
    return t.n, bytes(t.kmers.tobytes()), list(t.start), list(t.name_id)


def _read_in_child(files, q):
    q.put(_read_table(files)[:2])


def test_reader_threads_are_a_kept_crew(tmp_path, golden_motifs):
    """The TSV reader's host threads come from one process-wide crew (csrc/gfm_workers.cpp).  Calls from several
    host threads at once (one gets the crew, the others start threads of their own), repeated calls (the crew
    is reused) and a call in a fork()ed child (which holds none of the parent's threads) must all return what
    a single-threaded read returns."""
    import multiprocessing as mp
    import threading
    from grafimo_amd import synth
    _, flat = golden_motifs
    batch = synth.make_batch(40, 2000, 19, np.asarray(flat["ctcf_meme_unif#0"]["probs"]), 5)
    synth.write_tsv_dir(batch, str(tmp_path))
    files = sorted(str(p) for p in (tmp_path / "width_19").glob("*.tsv"))
    assert sum(os.path.getsize(f) for f in files) > (4 << 20)       # several threads are worth starting
    t1 = KmerTable(files, 19, False, 1)
    want = (t1.n, bytes(t1.kmers.tobytes()), list(t1.start), list(t1.name_id))
    assert want[0] == len(batch)
    for _ in range(3):
        assert _read_table(files) == want
    got = [None] * 4
    th = [threading.Thread(target=lambda i=i: got.__setitem__(i, _read_table(files))) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert all(g == want for g in got)
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=_read_in_child, args=(files, q))
    p.start()
    child = q.get(timeout=120)
    p.join(timeout=120)
    assert p.exitcode == 0 and child == want[:2]
    assert _read_table(files) == want


def test_paths_are_handed_over_as_one_blob():
    """nv.c_paths: the `const char *const *` of a path list as one encoded blob + an array of addresses into it (what
    StreamScan passes to gfm_scan_tsv_begin): every entry reads back as its path, non-ASCII included, none for no paths."""
    from grafimo_amd import _native as nv
    paths = [f"/tmp/dir/width_19/region_{i:05d}.tsv" for i in range(1000)] + ["/tmp/ü/β.tsv", "x"]
    arr, keep = nv.c_paths(paths)
    assert keep is not None
    for i in (0, 1, 499, 999, 1000, 1001):
        assert arr[i] == paths[i].encode()
    arr0, keep0 = nv.c_paths([])
    assert keep0 is None and not arr0


def test_lease_key_fingerprint_of_a_score_distribution():
    """DeviceMotif.lease keys a kept handle by the motif's numbers; the score distribution (152 KB) goes in as a fingerprint
    that sees any changed entry and entries that trade places, and costs a fraction of hashing the bytes."""
    from grafimo_amd.device import _fingerprint
    rng = np.random.default_rng(8)
    a = rng.random(19001)
    f = _fingerprint(a)
    assert f == _fingerprint(a.copy()) == _fingerprint(list(a)) and f[0] == 19001
    seen = {f}
    for i in rng.choice(len(a), size=200, replace=False):
        b = a.copy()
        b[i] = np.nextafter(b[i], 2.0)                      # one ulp in one entry
        seen.add(_fingerprint(b))
        j = int(rng.integers(0, len(a)))
        if a[i] != a[j]:
            c = a.copy()
            c[[i, j]] = c[[j, i]]                           # two entries trade places
            assert _fingerprint(c) != f
    assert len(seen) == 201
    assert _fingerprint(np.zeros(0)) == (0, 0, 0) and _fingerprint(a[:1000]) != _fingerprint(a[:1001])
