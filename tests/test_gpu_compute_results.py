"""compute_results (the S3 drop-in) on a real MI355X vs the reference's golden tables.
Written like the reference's own test_scoring (tests/grafimo_run_test.py:119-140)."""
import contextlib
import io
import datetime
import os

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN, REF_DATA
from ref_shapes import RefShapedMotif

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

FLOAT_RTOL = 1e-9   # north_star: within 1e-6 of the reference; the tables agree to ~1e-15


def _ctcf(pvalue_matrix):
    from grafimo_amd.motif_ops import build_motif_meme
    with contextlib.redirect_stdout(io.StringIO()):
        return build_motif_meme(os.path.join(REF_DATA, "MA0139.1.meme"), "unfrm_dst", 0.1, False,
                                os.cpu_count(), False, True, pvalue_matrix=pvalue_matrix)[0]


def _compare(df, exp):
    assert list(df.columns) == list(exp.columns)
    key = ["p-value", "start", "stop", "strand"]
    a = df.sort_values(key, kind="stable").reset_index(drop=True)
    b = exp.sort_values(key, kind="stable").reset_index(drop=True)
    assert len(a) == len(b)
    for c in exp.columns:
        if c in ("p-value", "q-value"):
            np.testing.assert_allclose(a[c].to_numpy(float), b[c].to_numpy(float), rtol=FLOAT_RTOL, atol=0)
        elif c == "score":
            assert np.array_equal(a[c].to_numpy(float), b[c].to_numpy(float))   # exact: s/scale + W*offset
        else:
            assert (a[c].astype(str) == b[c].astype(str)).all(), c


def test_motif_processing_with_device_dp_matches_reference(golden_motifs):
    """process_motif_for_logodds end to end (DP on the GPU): pval_matrix bit-identical."""
    _, flat = golden_motifs
    m = _ctcf(True)
    assert np.array_equal(m.pval_matrix, flat["ctcf_meme_unif#0"]["pmf"])
    assert m.pval_matrix.dtype == np.float64 and len(m.pval_matrix) == 19001


def test_scoring_like_the_reference_test(capsys):
    """test_scoring: compute_results(motif, dir, True, None, testmode=True) vs scoring_results.tsv"""
    from grafimo_amd.score_sequences import compute_results
    motif = _ctcf(True)
    exp = pd.read_csv(os.path.join(REF_DATA, "scoring_results.tsv"), sep="\t", index_col=0)
    res = compute_results(motif, REF_DATA, True, None, testmode=True)
    out = capsys.readouterr().out
    assert "Scoring hits for motif +MA0139.1." in out and "Scoring hits for motif -MA0139.1." in out
    assert "Scanned sequences:\t704" in out and "Scanned nucleotides:\t13376" in out
    assert "Computing q-values" in out
    tmp = os.path.join("/tmp", f"gpu_scoring_{os.getpid()}.tsv")
    res.to_csv(tmp, sep="\t")
    got = pd.read_csv(tmp, sep="\t", index_col=0)
    os.remove(tmp)
    _compare(got, exp)
    assert (np.diff(res["p-value"].to_numpy()) >= 0).all()     # sorted by p-value


@pytest.mark.parametrize("with_pmf", [True, False])
def test_reference_shaped_motif_drops_in(golden_motifs, with_pmf):
    """the S1/S3 seams take the reference's own Motif object (duck-typed on
    score_matrix + nucsmap, bg dict, pval_matrix, min_val/scale/offset/width/ids)."""
    from grafimo_amd import motif_processing
    from grafimo_amd.score_sequences import compute_results
    _, flat = golden_motifs
    rec = flat["ctcf_meme_unif#0"]
    motif = RefShapedMotif(rec, with_pmf)
    pmf = motif_processing.comp_pval_mat(motif, True)                    # S1, DP on the GPU
    assert np.array_equal(pmf, rec["pmf"])
    exp = pd.read_csv(os.path.join(REF_DATA, "scoring_results.tsv"), sep="\t", index_col=0)
    with contextlib.redirect_stdout(io.StringIO()):
        res = compute_results(motif, REF_DATA, True, None, testmode=True)   # S3
    tmp = os.path.join("/tmp", f"gpu_refshape_{os.getpid()}.tsv")
    res.to_csv(tmp, sep="\t")
    got = pd.read_csv(tmp, sep="\t", index_col=0)
    os.remove(tmp)
    _compare(got, exp)


@pytest.mark.parametrize("name", ["default_t1e-2", "qvalt_t0.6", "noqvalue_t5e-3", "norev_t1e-1",
                                  "recomb_t1", "norecomb_t1", "cores4_t5e-2"])
def test_flag_settings(golden_json, name):
    from grafimo_amd.score_sequences import compute_results
    from grafimo_amd.workflow import Findmotif
    case = golden_json("compute_results.json")[name]
    kw = case["kwargs"]
    wf = Findmotif(cores=kw.get("cores", 1), threshold=kw.get("threshold", 1e-4),
                   no_qvalue=kw.get("no_qvalue", False), qval_t=kw.get("qval_t", False),
                   no_reverse=kw.get("no_reverse", False), recomb=kw.get("recomb", False))
    motif = _ctcf(False)          # no pval_matrix on the Motif: the device computes it
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        df = compute_results(motif, REF_DATA, True, wf)
    exp = pd.DataFrame(case["df"]["rows"], columns=case["df"]["columns"])
    _compare(df, exp)
    # same user-visible counters as the reference printed
    for line in case["stdout"].splitlines():
        if line.startswith("Scanned"):
            assert line in buf.getvalue()


def test_synthetic_tsv_directory_end_to_end(tmp_path, golden_motifs):
    """vg-style TSV files written from a synthetic batch -> ingest -> GPU -> table, against the
    CPU oracle's compute_results on the same directory (multi-file, N rows, both strands)."""
    from grafimo_amd import synth
    from grafimo_amd.score_sequences import compute_results
    from grafimo_amd.workflow import Findmotif
    from oracle import oracle as orc
    _, flat = golden_motifs
    g = flat["ctcf_meme_unif#0"]
    motif = _ctcf(True)
    batch = synth.make_batch(6, 500, 19, g["probs"], synth.seed_for(1))
    synth.write_tsv_dir(batch, str(tmp_path))
    for kw in [dict(threshold=1e-3), dict(threshold=0.2, qval_t=True), dict(threshold=1e-2, no_reverse=True, recomb=True)]:
        wf = Findmotif(cores=3, **kw)
        with contextlib.redirect_stdout(io.StringIO()):
            df = compute_results(motif, str(tmp_path), True, wf)
        md = dict(score_matrix=g["score_matrix"], pmf=g["pmf"], min_val=g["min_val"], scale=g["scale"],
                  offset=g["offset"], width=19, motif_id=g["motif_id"], motif_name=g["motif_name"])
        ref = orc.compute_results(md, str(tmp_path), threshold=kw["threshold"], qval_t=kw.get("qval_t", False),
                                  no_reverse=kw.get("no_reverse", False), recomb=kw.get("recomb", False))
        cols = [c for c in ref if not c.startswith("_")]
        exp = pd.DataFrame({c: ref[c] for c in cols})
        assert len(df) > 0
        _compare(df, exp)


def test_streamed_scan_chunks_do_not_change_the_result(tmp_path, golden_motifs):
    """gfm_scan_tsv (the pipeline under compute_results): rows cut into many ragged chunks (hit list
    appended chunk after chunk, one histogram, row ids global) == one chunk == the monolithic
    KmerTable + gfm_scan_host path, for p- and q-value thresholds, --no-qvalue and --no-reverse."""
    from grafimo_amd import synth
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.score_sequences import KmerTable, StreamScan
    import glob
    _, flat = golden_motifs
    g = flat["ctcf_meme_unif#0"]
    batch = synth.make_batch(23, 700, 19, g["probs"], synth.seed_for(4))
    synth.write_tsv_dir(batch, str(tmp_path), regions_per_file=3)
    files = sorted(glob.glob(os.path.join(str(tmp_path), "width_19", "*.tsv")))
    assert len(files) == 8
    dm = DeviceMotif(g["score_matrix"], g["bg"], g["min_val"], g["scale"], g["offset"], g["pmf"])
    for thr, on_q, want_q, norev in [(1e-2, False, True, False), (0.3, True, True, False), (1e-3, False, False, True),
                                     (1.0, False, True, False)]:
        table = KmerTable(files, 19, norev, 2)
        ref = dm.scan_host(table.kmers, thr, on_qvalue=on_q, want_qvalues=want_q)
        assert len(ref["rows"]) > 0
        for chunk_rows in (0, 256 * 5, 256):
            sc = StreamScan(dm, files, norev, 3, thr, on_q, want_q, chunk_rows=chunk_rows)
            assert sc.n == table.n and sc.stats.n_rows == table.n
            assert sc.stats.n_chunks == (1 if chunk_rows == 0 else -(-table.n // chunk_rows))
            assert np.array_equal(sc.rows, ref["rows"]) and np.array_equal(sc.scaled, ref["scaled"])
            assert np.array_equal(sc.logodds, ref["logodds"]) and np.array_equal(sc.pvalue, ref["pvalue"])
            if want_q:
                assert np.array_equal(sc.qvalue, ref["qvalue"])
            r = sc.rows
            assert np.array_equal(sc.kmers, table.kmers[r]) and np.array_equal(sc.start, table.start[r])
            assert np.array_equal(sc.stop, table.stop[r]) and np.array_equal(sc.strand, table.strand[r])
            assert np.array_equal(sc.freq, table.freq[r]) and np.array_equal(sc.is_ref, table.is_ref[r])
            assert [sc.names[i] for i in sc.name_id] == [table.names[i] for i in table.name_id[r]]
    # a malformed row anywhere fails the whole scan like the ingest does
    with open(files[3], "a") as fh:
        fh.write("chr22:1-2\tACGT\tchr22:1+\tchr22:5+\t1\tref\t1+,\n")
    with pytest.raises(Exception) as e:
        StreamScan(dm, files, False, 3, 1e-2, False, True, chunk_rows=256)
    assert "k-mer length" in str(e.value)
    dm.close()


def test_streamed_scan_grows_its_hit_list(tmp_path, golden_motifs):
    """More hits than the pooled hit list holds (threshold 1: every one of 1.2e6 rows).  The scan stores no scores (round 5):
    the score kernel counts what its list could not hold, the library grows the list from that count and asks for the scan
    again (GFM_ERR_OVERFLOW), StreamScan runs both phases once more -- the same rows as the oracle's.  Also with a q-value
    threshold, where the list that overflows is the one of the p < t candidates."""
    from grafimo_amd import synth
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.score_sequences import StreamScan
    import glob
    _, flat = golden_motifs
    g = flat["ctcf_meme_unif#0"]
    batch = synth.make_batch(600, 2000, 19, g["probs"], synth.seed_for(5))
    synth.write_tsv_dir(batch, str(tmp_path), regions_per_file=100)
    files = sorted(glob.glob(os.path.join(str(tmp_path), "width_19", "*.tsv")))
    dm = DeviceMotif(g["score_matrix"], g["bg"], g["min_val"], g["scale"], g["offset"], g["pmf"])
    nv_lib = __import__("grafimo_amd._native", fromlist=["lib"]).lib()
    nv_lib.gfm_scan_release_buffers()                   # start from the default hit capacity (2^20)
    sc = StreamScan(dm, files, False, 4, 1.0, False, True, chunk_rows=256 * 1024)
    order = np.argsort(batch.region, kind="stable")     # files hold the regions in ascending order
    from oracle import oracle as orc
    ptab = orc.p_table(g["pmf"])
    exp_sc, exp_p = orc.score_kmers_table(batch.kmers[order], g["score_matrix"], ptab, g["min_val"])
    exp = np.nonzero(exp_p < 1.0)[0]                    # strict: rows holding N (p = 1) are no hits
    assert sc.n == 1_200_000 and len(exp) > (1 << 20)
    assert np.array_equal(sc.rows, exp) and np.array_equal(sc.scaled, exp_sc[exp])
    assert np.array_equal(sc.kmers, batch.kmers[order][exp])
    nv_lib.gfm_scan_release_buffers()                   # the default capacity again
    sq = StreamScan(dm, files, False, 4, 1.0, True, True, chunk_rows=256 * 1024)      # q < 1: the candidates are all p < 1 rows
    q = orc.fdr_bh(exp_p)
    exp_q = np.nonzero(q < 1.0)[0]
    assert len(exp_q) > (1 << 20) and np.array_equal(sq.rows, exp_q) and np.allclose(sq.qvalue, q[exp_q], rtol=1e-12, atol=0)
    dm.close()


def test_a_closed_scan_leaves_o_chunk_memory_behind(tmp_path):
    """host text arena, 8 bytes of host memory per row and device score blocks stayed O(dataset) for the
    life of the process.  Now no score is stored at all, and what a closed scan keeps is bounded (GRAFIMO_SCAN_KEEP_BYTES;
    here 8 MiB): after compute_results over 3e6 rows (270 MB of text) the process holds neither the text nor per-row host
    memory nor per-row device memory -- resident set and free HBM are back to within a chunk's worth of where they were."""
    import subprocess
    import sys
    from conftest import ROOT
    code = """
import contextlib, io, os, sys, resource
sys.path.insert(0, %r)
import numpy as np, torch
from grafimo_amd import synth
from grafimo_amd.motif_ops import build_motif_meme_host
from grafimo_amd.score_sequences import compute_results
from grafimo_amd.workflow import Findmotif
def rss():
    return int(open('/proc/self/statm').read().split()[1]) * os.sysconf('SC_PAGE_SIZE')
motif = build_motif_meme_host(os.path.join(%r, 'MA0139.1.meme'), 'unfrm_dst', 0.1, False)[0]
d = sys.argv[1]
batch = synth.make_batch(1500, 2000, 19, np.asarray(motif.count_matrix, dtype=np.float64), synth.seed_for(2))
synth.write_tsv_dir(batch, d, regions_per_file=10)
del batch
text_bytes = sum(os.path.getsize(os.path.join(d, 'width_19', f)) for f in os.listdir(os.path.join(d, 'width_19')))
wf = Findmotif(cores=8, threshold=1e-3)
with contextlib.redirect_stdout(io.StringIO()):
    small = compute_results(motif, d, False, Findmotif(cores=8, threshold=1e-9))      # warm: runtime, streams, slots
torch.cuda.synchronize()
free0, rss0 = torch.cuda.mem_get_info()[0], rss()
with contextlib.redirect_stdout(io.StringIO()):
    df = compute_results(motif, d, False, wf)
torch.cuda.synchronize()
free1, rss1 = torch.cuda.mem_get_info()[0], rss()
print('ROWS', 3000000, 'HITS', len(df), 'TEXT', text_bytes, 'RSS_DELTA', rss1 - rss0, 'HBM_DELTA', free0 - free1)
"""
    from conftest import REF_DATA
    r = subprocess.run([sys.executable, "-c", code % (ROOT, REF_DATA), str(tmp_path)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, GRAFIMO_SCAN_KEEP_BYTES=str(8 << 20)))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    f = r.stdout.split()
    val = {f[i]: int(f[i + 1]) for i in range(0, len(f), 2)}
    assert val["TEXT"] > 250_000_000 and val["HITS"] > 1000
    assert val["RSS_DELTA"] < 96 << 20, val            # (the text alone is 270 MB, the line offsets 24 MB)
    assert val["HBM_DELTA"] < 64 << 20, val            # (3e6 int32 scores would be 12 MB per motif -- and are not there)


def test_a_grown_hit_list_survives_the_retry(tmp_path):
    """a hit list that turns out too short is grown at finish and the scan run again -- the retry CLOSES the scan
    first, and the closed scan's trim() dropped every list above GRAFIMO_SCAN_KEEP_BYTES: the second begin reserved the default
    again and overflowed again ('run the scan again' on every attempt) once more than keep / 16 rows passed.  Here keep =
    1 MiB (65 536 entries) and 1.6e6 rows of 3e6... pass p < 0.6: compute_results returns them all."""
    import subprocess
    import sys
    from conftest import REF_DATA, ROOT
    code = """
import contextlib, io, os, sys
sys.path.insert(0, %r)
import numpy as np
from grafimo_amd import synth
from grafimo_amd.motif_ops import build_motif_meme_host
from grafimo_amd.score_sequences import compute_results, compute_results_many
from grafimo_amd.workflow import Findmotif
from oracle import oracle as orc
motif = build_motif_meme_host(os.path.join(%r, 'MA0139.1.meme'), 'unfrm_dst', 0.1, False)[0]
d = sys.argv[1]
batch = synth.make_batch(1100, 2000, 19, np.asarray(motif.count_matrix, dtype=np.float64), synth.seed_for(2))
synth.write_tsv_dir(batch, d, regions_per_file=10)
sm, bg = motif.dense_score_matrix(), motif.dense_bg()
pmf = orc.comp_pval_mat(sm, bg)
_, _, pv = orc.score_kmers(batch.kmers, sm, pmf, motif.min_val, motif.scale, float(motif.offset), sum_mode=1)
for t in (0.6, 0.9):
    wf = Findmotif(cores=8, threshold=t, recomb=True)
    with contextlib.redirect_stdout(io.StringIO()):
        df = compute_results(motif, d, False, wf)
        many = compute_results_many([motif, motif], d, False, wf)
    want = int((pv < t).sum())
    print('T', t, 'HITS', len(df), 'WANT', want, 'MANY', len(many[0]), len(many[1]))
    assert len(df) == want and len(many[0]) == want and len(many[1]) == want and want > 1200000
"""
    r = subprocess.run([sys.executable, "-c", code % (ROOT, REF_DATA), str(tmp_path)], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, GRAFIMO_SCAN_KEEP_BYTES=str(1 << 20)))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert r.stdout.count("HITS") == 2


def test_sharded_entry_point_on_one_gpu(golden_json):
    """compute_results_sharded with the HIP backend and no process group == compute_results."""
    from grafimo_amd.distributed import compute_results_sharded
    from grafimo_amd.score_sequences import compute_results
    from grafimo_amd.workflow import Findmotif
    motif = _ctcf(True)
    for kw in [dict(threshold=1e-2), dict(threshold=0.6, qval_t=True), dict(threshold=1.0, recomb=True)]:
        with contextlib.redirect_stdout(io.StringIO()):
            a = compute_results_sharded(motif, REF_DATA, True, Findmotif(**kw))
            b = compute_results(motif, REF_DATA, True, Findmotif(**kw))
        assert len(a) == len(b) and len(a) > 0
        _compare(a, b)


def test_sharded_same_width_set_uses_the_batched_launch(golden_motifs):
    """sharded_scan_same_width with HipBackends (gfm_score_kmers_multi under it) == one sharded_scan
    per motif: three W=19 motifs (CTCF with three pseudocounts)."""
    from grafimo_amd import synth
    from grafimo_amd.distributed import HipBackend, sharded_scan, sharded_scan_same_width
    from grafimo_amd.motif_ops import build_motif_meme_host
    path = os.path.join(REF_DATA, "MA0139.1.meme")
    motifs = [build_motif_meme_host(path, "unfrm_dst", pc, False)[0] for pc in (0.1, 0.5, 2.0)]
    assert len({m.scale for m in motifs} | {m.min_val for m in motifs}) > 2     # really different matrices
    batch = synth.make_batch(40, 500, 19, np.asarray(motifs[0].count_matrix), synth.seed_for(9))
    backends = [HipBackend(m) for m in motifs]
    try:
        for on_q, thr in [(False, 1e-3), (True, 0.3)]:
            many = sharded_scan_same_width(backends, batch.kmers, thr, on_q, True)
            for j, b in enumerate(backends):
                one = sharded_scan(b, batch.kmers, thr, on_q, True)
                assert len(one["rows"]) > 0 and many[j]["n_scored"] == len(batch)
                for key in ("rows", "scaled", "logodds", "pvalue", "qvalue"):
                    assert np.array_equal(many[j][key], one[key]), (j, key)
    finally:
        for b in backends:
            b.close()


def test_scanner_collectives_on_one_gpu_rccl(golden_motifs):
    """KmerScanner with an RCCL process group of one rank: the all-reduce of the histogram and
    the gather of hits are really issued (same calls as at N = 8), results unchanged."""
    import socket
    import torch.distributed as dist
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.scan import KmerScanner
    from grafimo_amd import synth
    _, flat = golden_motifs
    g = flat["ctcf_meme_unif#0"]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    # a short collective timeout: a hung collective fails this test after a minute instead of holding the run
    # for RCCL's default ten
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=dev, timeout=datetime.timedelta(seconds=60))
    try:
        dm = DeviceMotif(g["score_matrix"], g["bg"], g["min_val"], g["scale"], g["offset"], g["pmf"])
        batch = synth.make_batch(50, 500, 19, g["probs"], synth.seed_for(3))
        d_k = torch.from_numpy(batch.kmers).to(dev)
        n = len(batch)
        plain = KmerScanner(dm, n, device=dev, side_stream=False)
        # (one communicator, as the product uses it: the gather of batch k and the all-reduce of batch k + 1
        # then execute in issue order.  A second communicator for the gather lets their kernels overlap on the
        # device, which RCCL does not promise to survive: one run in about forty hung here until the watchdog.)
        coll = KmerScanner(dm, n, device=dev, side_stream=True, always_collective=True)
        for on_q, thr in [(False, 1e-3), (True, 0.2)]:
            r0 = plain.collect(plain.enqueue(d_k, thr, on_qvalue=on_q))
            for _ in range(3):   # slots rotate; the side stream hands buffers back cleared
                slot = coll.enqueue(d_k, thr, on_qvalue=on_q, gather_hits=True)
            r1 = coll.collect(slot)
            assert np.array_equal(r0["rows"], r1["rows"]) and np.array_equal(r0["scaled"], r1["scaled"])
            assert np.array_equal(r0["qtable"], r1["qtable"]) and r0["n_scored"] == r1["n_scored"] == n
            assert len(r0["rows"]) > 0
        # the per-step gather is cut down to what is hit: p < 1e-3 leaves a few hundred hits of
        # 25 000 rows; the gathered slice shrinks from the whole hit buffer to one 512-entry granule or two ...
        slot = coll.enqueue(d_k, 1e-3, gather_hits=True)
        k = len(coll.collect(slot)["rows"])
        full = coll.gather_len
        small = coll.size_gather()
        assert full == n + 1 and k < small - 1 <= 2 * 512 * (1 + k // 512) and small < full
        for _ in range(3):
            slot = coll.enqueue(d_k, 1e-3, gather_hits=True)
        assert slot.gathered[0].numel() == small
        r2 = coll.collect(slot)
        assert np.array_equal(r2["rows"], r0["rows"] if False else plain.collect(plain.enqueue(d_k, 1e-3))["rows"])
        # ... and a later batch with more hits than that is reported, not truncated
        slot = coll.enqueue(d_k, 0.5, gather_hits=True)
        with pytest.raises(OverflowError):
            coll.collect(slot)
        dm.close()
    finally:
        dist.destroy_process_group()


def test_cli_scoring_stage_writes_the_reference_reports(tmp_path, golden_json, capsys):
    """python -m grafimo_amd -m MA0139.1.meme -s <dir> -t 1e-2 -o out  vs the reference's report"""
    from grafimo_amd.__main__ import main
    out = tmp_path / "out"
    main(["-m", os.path.join(REF_DATA, "MA0139.1.meme"), "-s", REF_DATA, "-t", "1e-2", "-o", str(out), "-j", "2"])
    text = capsys.readouterr().out
    assert "Scanned sequences:\t704" in text and "Elapsed time" in text
    got = pd.read_csv(out / "grafimo_out.tsv", sep="\t", index_col=0)
    exp = pd.read_csv(os.path.join(GOLDEN, "report", "default_t1e-2.tsv"), sep="\t", index_col=0)
    _compare(got, exp)
    gff = open(out / "grafimo_out.gff").read().splitlines()
    ref = open(os.path.join(GOLDEN, "report", "default_t1e-2.gff")).read().splitlines()
    assert len(gff) == len(ref) and gff[0] == "##gff-version 3"
    # columns 1-8 (coordinates, rounded score, strand) are identical text; p/q differ at 1e-16 at most
    assert sorted(l.split("\t")[:8] for l in gff[1:]) == sorted(l.split("\t")[:8] for l in ref[1:])
    assert os.path.isfile(out / "grafimo_out.html")
    main(["-m", os.path.join(REF_DATA, "MA0139.1.jaspar"), "-s", REF_DATA, "-t", "5e-3", "-q", "-f"])
    assert "matched_sequence" in capsys.readouterr().out


def test_compute_results_many_equals_per_motif_calls(tmp_path):
    """The motif-set form (one ingest and one upload per width, batched launches) returns, motif by
    motif, the table compute_results returns: the six motifs of a MEME file over TSV directories of
    their widths, p- and q-value thresholds, and the CLI path that uses it."""
    from grafimo_amd import synth
    from grafimo_amd.__main__ import main
    from grafimo_amd.motif_ops import build_motif_meme_host
    from grafimo_amd.score_sequences import compute_results, compute_results_many
    from grafimo_amd.workflow import Findmotif
    meme = os.path.join(GOLDEN, "synth", "multi.meme")
    motifs = build_motif_meme_host(meme, os.path.join(GOLDEN, "synth", "bg_1.txt"), 0.1, False)
    motifs += build_motif_meme_host(os.path.join(REF_DATA, "MA0139.1.meme"), "unfrm_dst", 0.1, False)
    widths = sorted({m.width for m in motifs})
    assert len(motifs) == 7 and len(widths) < len(motifs)          # some widths are shared
    for w in widths:
        probs = np.asarray([m for m in motifs if m.width == w][0].count_matrix)
        synth.write_tsv_dir(synth.make_batch(3, 400, w, probs, synth.seed_for(w)), str(tmp_path))
    for kw in [dict(threshold=1e-2), dict(threshold=0.5, qval_t=True, recomb=True), dict(threshold=1e-2, no_qvalue=True)]:
        wf = Findmotif(cores=2, **kw)
        with contextlib.redirect_stdout(io.StringIO()) as out:
            many = compute_results_many(motifs, str(tmp_path), True, wf)
        assert out.getvalue().count("Scanned sequences:") == len(motifs)
        for m, df in zip(motifs, many):
            with contextlib.redirect_stdout(io.StringIO()):
                one = compute_results(m, str(tmp_path), True, wf)
            assert len(df) == len(one) > 0, (m.motif_id, kw)
            _compare(df, one)
    main(["-m", meme, "-k", os.path.join(GOLDEN, "synth", "bg_1.txt"), "-s", str(tmp_path), "-t", "1e-2",
          "-o", str(tmp_path / "out")])
    assert len([f for f in os.listdir(tmp_path / "out") if f.endswith(".tsv")]) == 6


def test_two_ranks_on_one_gpu_through_the_scanner():
    """The N = 2 path of KmerScanner on the real kernels: two processes share this GPU over gloo (RCCL
    refuses a duplicate GPU): histogram all-reduce, hit gather, global row ids and the merge on rank 0
    give what one process gives over all rows (scripts/two_rank_probe.py exits 1 otherwise)."""
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "two_rank_probe.py")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("two ranks == one process: True") == 2


@pytest.mark.parametrize("n_slots,ring", [(2, None), (3, None), (4, 2), (6, 2), (8, 3), (9, None)])
def test_scanner_slot_rings_under_pipelining(golden_motifs, n_slots, ring):
    """Batches enqueued back to back without a host synchronisation in between (two slots: the device orders
    slot reuse; three and more: the host paces it and the main stream carries score kernels only; more than four: more than the
    library's workspace ring, which then orders the reuse itself; `ring`: the score arrays as a shorter ring than the
    slots).  Forty-one batches walk the library's workspace ring (4) and hit-counter ring (8) many times; thresholds
    alternate between one that makes every wave flush its hit queue mid-run and selective ones, with and
    without a q-value threshold.  Every batch must equal the oracle's scores / histogram-derived q-table."""
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.scan import KmerScanner
    from oracle import oracle as orc
    _, flat = golden_motifs
    g = flat["ctcf_meme_unif#0"]
    dev = torch.device("cuda:0")
    dm = DeviceMotif(g["score_matrix"], g["bg"], g["min_val"], g["scale"], g["offset"], g["pmf"])
    ptab = orc.p_table(g["pmf"])
    rng = np.random.default_rng(5)
    n = 60_000
    batches = []
    for b in range(41):
        km = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(n - 7 * b, 19))
        km[rng.integers(0, len(km), 40), rng.integers(0, 19, 40)] = ord("N")
        batches.append(km)
    d_batches = [torch.from_numpy(k).to(dev) for k in batches]
    plan = [(1.0, False), (1e-2, False), (0.3, True), (1e-3, False), (1.0, False), (0.5, True), (1e-2, False)]
    sc = KmerScanner(dm, n, device=dev, n_slots=n_slots, score_buffers=ring)
    assert sc.host_paced == (n_slots >= 3)
    assert len({s_.scores.data_ptr() for s_ in sc.slots}) == (ring or n_slots)
    b = 0
    while b < len(batches):
        group = list(range(b, min(b + n_slots, len(batches))))
        slots = [sc.enqueue(d_batches[i], plan[i % len(plan)][0], on_qvalue=plan[i % len(plan)][1]) for i in group]
        for i, slot in zip(group, slots):
            thr, on_q = plan[i % len(plan)]
            res = sc.collect(slot)
            exp_scores, p = orc.score_kmers_table(batches[i], g["score_matrix"], ptab, g["min_val"])
            q = orc.fdr_bh(p)
            keep = np.nonzero((q if on_q else p) < thr)[0]
            assert res["n_scored"] == len(batches[i])
            assert np.array_equal(res["rows"], keep), (i, thr, on_q)
            assert np.array_equal(res["scaled"], exp_scores[keep])
            assert np.allclose(res["qtable"][exp_scores[keep]], q[keep], rtol=1e-12, atol=0)
        b += n_slots
    dm.close()


def test_timed_launches_with_a_tail_stream_keep_their_order(golden_motifs):
    """The timing events of gfm_profile_enable ride on the score kernel's dispatch packet, and on a timed launch
    the tail stream waits on the timer's stop event instead of the library's own (csrc dispatch_quad).  Every
    launch timed, tail on a side stream, batches back to back: results unchanged, one plausible duration per
    launch."""
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.scan import KmerScanner
    from oracle import oracle as orc
    _, flat = golden_motifs
    g = flat["ctcf_meme_unif#0"]
    dev = torch.device("cuda:0")
    dm = DeviceMotif(g["score_matrix"], g["bg"], g["min_val"], g["scale"], g["offset"], g["pmf"])
    ptab = orc.p_table(g["pmf"])
    rng = np.random.default_rng(8)
    n = 300_000
    kms = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(n, 19)) for _ in range(3)]
    d_kms = [torch.from_numpy(k).to(dev) for k in kms]
    sc = KmerScanner(dm, n, device=dev, n_slots=3)
    dm.profile_enable(16, every=1)
    for rep in range(3):
        slots = [sc.enqueue(d, 1e-3) for d in d_kms]
        for km, slot in zip(kms, slots):
            res = sc.collect(slot)
            exp_sc, p = orc.score_kmers_table(km, g["score_matrix"], ptab, g["min_val"])
            keep = np.nonzero(p < 1e-3)[0]
            assert np.array_equal(res["rows"], keep) and np.array_equal(res["scaled"], exp_sc[keep])
            assert res["n_scored"] == n
    ms = dm.profile_read()
    dm.profile_enable(0)
    assert len(ms) == 9 and np.all(ms > 0.0005) and np.all(ms < 50.0)
    dm.close()


def test_two_scanners_share_one_handle_interleaved(golden_motifs):
    """the workspace ring and the hit-counter ring belong to the DeviceMotif, not to a scanner.  Two
    pipelined scanners (three slots each: both promise 'my batch four back is done' through
    GFM_FLAG_CALLER_ORDERS_REUSE) that take turns on ONE handle shift each other's ring positions; the library must
    notice the second user and order the workspace reuse itself.  Large batches (the post kernels are still pending
    when the next score kernels are enqueued), no host synchronisation inside a round; every batch equals the oracle."""
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.scan import KmerScanner
    from oracle import oracle as orc
    _, flat = golden_motifs
    g = flat["ctcf_meme_unif#0"]
    dev = torch.device("cuda:0")
    dm = DeviceMotif(g["score_matrix"], g["bg"], g["min_val"], g["scale"], g["offset"], g["pmf"])
    ptab = orc.p_table(g["pmf"])
    rng = np.random.default_rng(17)
    n = 1_500_000
    batches = [rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(n - 1000 * b, 19)) for b in range(4)]
    d_batches = [torch.from_numpy(k).to(dev) for k in batches]
    expect = []
    for km in batches:
        sc_exp, p = orc.score_kmers_table(km, g["score_matrix"], ptab, g["min_val"])
        expect.append((sc_exp, p, np.bincount(sc_exp, minlength=dm.L)))
    a, b = KmerScanner(dm, n, device=dev, n_slots=3), KmerScanner(dm, n, device=dev, n_slots=3)
    for rnd in range(6):
        order = [(a, 0), (b, 1), (a, 2), (b, 3), (a, 1), (b, 0)] if rnd % 2 == 0 else [(a, 3), (a, 2), (b, 1), (a, 0), (b, 2), (b, 3)]
        slots = [(sc, i, sc.enqueue(d_batches[i], 1e-2 if (i + rnd) % 2 else 1.0)) for sc, i in order[:3]]
        slots += [(sc, i, sc.enqueue(d_batches[i], 1e-3)) for sc, i in order[3:]]
        for k, (sc, i, slot) in enumerate(slots):
            thr = (1e-2 if (i + rnd) % 2 else 1.0) if k < 3 else 1e-3
            res = sc.collect(slot)
            sc_exp, p, _ = expect[i]
            keep = np.nonzero(p < thr)[0]
            assert res["n_scored"] == len(batches[i]), (rnd, k)
            assert np.array_equal(res["rows"], keep), (rnd, k, thr)
            assert np.array_equal(res["scaled"], sc_exp[keep])
    dm.close()


def test_streamed_scan_of_a_motif_set_in_chunks(tmp_path, golden_motifs):
    """gfm_scan_tsv_begin / _finish with several motifs of one width (every chunk scored by gfm_score_kmers_multi),
    cut into ragged chunks and fed by many more files than parse threads: every motif's hits equal the single-motif
    streamed scan, for p- and q-value thresholds, --no-qvalue and --no-reverse; and the two-phase form with caller-owned
    histogram tensors equals the one-call form (the histograms are the bincount of the scores, doubled in between to
    show that finish() reads them as they are then)."""
    import glob
    from grafimo_amd import synth
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.score_sequences import StreamScan
    from oracle import oracle as orc
    _, flat = golden_motifs
    keys = ["ctcf_meme_unif#0", "ctcf_meme_bgnt#0", "multi_meme_bg1#3", "ctcf_jaspar_bgnt_p1#0"]
    gs = [flat[k] for k in keys]
    batch = synth.make_batch(37, 500, 19, gs[0]["probs"], synth.seed_for(6))
    synth.write_tsv_dir(batch, str(tmp_path), regions_per_file=2)
    files = sorted(glob.glob(os.path.join(str(tmp_path), "width_19", "*.tsv")))
    assert len(files) == 19
    dms = [DeviceMotif(g["score_matrix"], g["bg"], g["min_val"], g["scale"], g["offset"], g["pmf"]) for g in gs]
    cols = ("rows", "scaled", "logodds", "pvalue", "kmers", "start", "stop", "strand", "freq", "is_ref")
    for thr, on_q, want_q, norev in [(1e-2, False, True, False), (0.3, True, True, False), (1e-3, False, False, True)]:
        singles = [StreamScan(dm, files, norev, 3, thr, on_q, want_q) for dm in dms]
        for chunk in (0, 1024, 4096 + 256):
            multi = StreamScan(dms, files, norev, 5, thr, on_q, want_q, chunk_rows=chunk)
            assert multi.n == singles[0].n and len(multi.hits) == len(dms)
            assert multi.hits[0].n_hits > 0                   # the batch is planted with the first motif
            for one, h in zip(singles, multi.hits):
                assert h.n_hits == one.n_hits
                for c in cols:
                    assert np.array_equal(getattr(h, c), getattr(one, c)), (thr, chunk, c)
                # REGION names: ids index the scan's own table, which holds the names of ITS hit rows only
                assert [multi.names[i] for i in h.name_id] == [one.names[i] for i in one.name_id]
                if want_q:
                    assert np.array_equal(h.qvalue, one.qvalue)
            assert set(multi.names) == set().union(*[set(x.names) for x in singles])
    # two phases, caller-owned histograms
    dev = torch.device("cuda:0")
    hist = torch.ones((len(dms), dms[0].L), dtype=torch.int64, device=dev)          # begin() zeroes them
    two = StreamScan(dms, files, False, 4, 0.3, True, True, chunk_rows=2048, hists=[hist[j] for j in range(len(dms))],
                     defer=True)
    ptabs = [orc.p_table(g["pmf"]) for g in gs]
    for j, g in enumerate(gs):
        exp, _ = orc.score_kmers_table(batch.kmers, g["score_matrix"], ptabs[j], g["min_val"])
        assert np.array_equal(hist[j].cpu().numpy(), np.bincount(exp, minlength=dms[j].L))
    one_call = StreamScan(dms, files, False, 4, 0.3, True, True)
    hist *= 2                                   # what a two-rank all-reduce of identical shards would leave
    torch.cuda.synchronize()
    two.finish()
    for j, g in enumerate(gs):
        # BH q-values are invariant under doubling every count (n and every rank double): same rows, same q
        assert np.array_equal(two.hits[j].rows, one_call.hits[j].rows)
        np.testing.assert_allclose(two.hits[j].qvalue, one_call.hits[j].qvalue, rtol=1e-12, atol=0)
    for dm in dms:
        dm.close()


def test_many_motifs_sharded_entry_point_on_one_gpu(tmp_path):
    """compute_results_many_sharded (one streamed pass per width for the whole motif set) with no process group ==
    compute_results per motif."""
    from grafimo_amd import synth
    from grafimo_amd.distributed import compute_results_many_sharded
    from grafimo_amd.motif_ops import build_motif_meme_host
    from grafimo_amd.score_sequences import compute_results
    from grafimo_amd.workflow import Findmotif
    motifs = build_motif_meme_host(os.path.join(GOLDEN, "synth", "multi.meme"), os.path.join(GOLDEN, "synth", "bg_1.txt"),
                                   0.1, False)
    widths = sorted({m.width for m in motifs})
    for w in widths:
        probs = np.asarray([m for m in motifs if m.width == w][0].count_matrix, dtype=np.float64)
        synth.write_tsv_dir(synth.make_batch(5, 2 * (200 - w + 1) + 120, w, probs, synth.seed_for(8) + w), str(tmp_path))
    for kw in (dict(threshold=1e-2), dict(threshold=0.5, qval_t=True, recomb=True)):
        wf = Findmotif(cores=2, **kw)
        with contextlib.redirect_stdout(io.StringIO()):
            many = compute_results_many_sharded(motifs, str(tmp_path), True, wf)
            for m, df in zip(motifs, many):
                _compare(df, compute_results(m, str(tmp_path), True, wf))
    assert sum(len(df) for df in many) > 0


def test_two_ranks_on_one_gpu_through_the_sharded_entry_points():
    """The N = 2 path of compute_results_sharded / compute_results_many_sharded on the real kernels: two processes share
    the GPU over gloo, each runs the streamed scan over its shard, the histograms are all-reduced between the two phases
    of the scan; rank 0's tables equal compute_results over all files (scripts/two_rank_sharded_probe.py exits 1 otherwise)."""
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "two_rank_sharded_probe.py")], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert r.stdout.count("two ranks == one process: True") == 9
    assert r.stdout.count("ranks holding shards of the graph == one process with all of it: True") == 8
    # ... and scan_graph's manifest through compute_results[_many]_sharded: disjoint graph shards per rank, the motif set in
    # one enumeration per rank
    assert r.stdout.count("the manifest under two ranks == one process with the whole graph: True") == 4


def test_hit_columns_from_the_kept_text_and_from_the_files_agree(tmp_path, monkeypatch):
    """The streamed scan keeps k-mer + line offset per row and reads the OTHER columns for the hit rows only: from the
    files' text kept in the pool's arena, or -- for what does not fit it -- back from the files (pread per hit, or the
    whole file when it holds many).  All three ways (arena for everything, for nothing, for a part) give the table the
    oracle gives, at a hit density of a few rows and of every row."""
    from grafimo_amd import synth
    from grafimo_amd.score_sequences import compute_results
    from grafimo_amd.workflow import Findmotif
    from oracle import oracle as orc
    motif = _ctcf(True)
    batch = synth.make_batch(23, 700, 19, np.asarray(motif.count_matrix, dtype=np.float64), synth.seed_for(12))
    synth.write_tsv_dir(batch, str(tmp_path))
    md = dict(score_matrix=motif.dense_score_matrix(), pmf=np.asarray(motif.pval_matrix), min_val=motif.min_val, scale=motif.scale,
              offset=float(motif.offset), width=19, motif_id=motif.motif_id, motif_name=motif.motif_name)
    for kw in (dict(threshold=1e-4), dict(threshold=1.0, recomb=True), dict(threshold=0.2, qval_t=True, no_reverse=True)):
        ref = orc.compute_results(md, str(tmp_path), threshold=kw["threshold"], qval_t=kw.get("qval_t", False), no_qvalue=False,
                                  no_reverse=kw.get("no_reverse", False), recomb=kw.get("recomb", False))
        exp = pd.DataFrame({c: ref[c] for c in ref if not c.startswith("_")})
        for text_bytes in (None, "0", "200000"):
            if text_bytes is None:
                monkeypatch.delenv("GRAFIMO_SCAN_TEXT_BYTES", raising=False)
            else:
                monkeypatch.setenv("GRAFIMO_SCAN_TEXT_BYTES", text_bytes)
            with contextlib.redirect_stdout(io.StringIO()):
                df = compute_results(motif, str(tmp_path), True, Findmotif(cores=3, **kw))
            key = ["p-value", "sequence_name", "start", "stop", "strand", "matched_sequence", "haplotype_frequency"]
            _compare(df.sort_values(key).reset_index(drop=True), exp.sort_values(key).reset_index(drop=True))
    assert len(exp) > 10


@pytest.mark.timeout(120)
def test_streamed_scan_without_any_file_returns_an_empty_scan(golden_motifs):
    """gfm_scan_tsv_begin with n_paths == 0 -- the shard of a rank when there are more ranks than files -- returns an
    empty scan instead of waiting for a first chunk that never comes."""
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.score_sequences import StreamScan
    dm = DeviceMotif.from_motif(_ctcf(True))
    for on_q, want_q in ((False, True), (True, True), (False, False)):
        hist = torch.full((dm.L,), 7, dtype=torch.int64, device="cuda:0") if want_q else None
        scan = StreamScan(dm, [], False, 4, 1e-3, on_q, want_q, hists=[hist] if want_q else None)
        assert scan.n == 0 and scan.hits[0].n_hits == 0 and scan.names == []
        if want_q:
            assert int(hist.abs().sum().item()) == 0        # the caller's histogram was zeroed: it joins an all-reduce
    dm.close()


@pytest.mark.timeout(600)
def test_two_ranks_one_file_leaves_a_rank_without_files():
    """two processes, ONE TSV file: rank 1 runs the streamed scan over nothing and still takes part in the histogram
    all-reduce and the gathers (scripts/two_rank_sharded_probe.py --files 1)."""
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "two_rank_sharded_probe.py"), "--files", "1"],
                       capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert r.stdout.count("two ranks == one process: True") == 9
    assert r.stdout.count("ranks holding shards of the graph == one process with all of it: True") == 8


@pytest.mark.timeout(900)
def test_four_ranks_on_one_gpu_through_the_sharded_entry_points():
    """scripts/two_rank_sharded_probe.py with FOUR processes sharing the test GPU (nine files over four ranks: ragged
    shards): rank 0's tables equal compute_results over all files."""
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "two_rank_sharded_probe.py"), "--ranks", "4"],
                       capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert r.stdout.count("ranks == one process: True") == 9
    assert r.stdout.count("ranks holding shards of the graph == one process with all of it: True") == 8


@pytest.mark.timeout(900)
def test_bench_distributed_path_on_one_gpu_with_reserved_cus():
    """The N > 1 bench path on the one GPU the test box has: --force-dist initialises RCCL with a one-rank group, every
    step issues the histogram all-reduce and the sized hit gather on the tail / gather streams beside a persistent score
    grid that leaves 16 CUs free (GRAFIMO_RESERVE_CUS=16, what bench.py sets for N > 1); the line must carry the group's
    size, the tail timing with the collectives in it, and have passed its own oracle slice."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, GRAFIMO_RESERVE_CUS="16", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--config", "3", "--rows", "8000000",
                        "--steps", "12", "--warmup", "3", "--bursts", "2", "--no-cpu-baseline", "--no-e2e", "--no-extras"],
                       capture_output=True, text=True, env=env, timeout=800)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][-1])
    assert rec["rccl_world"] == 1 and rec["n_gpus"] == 1 and rec["scaling"] == "weak"
    assert rec["tail_ms"] and rec["tail_ms"]["timed"] > 0 and "all-reduce" in rec["tail_ms"]["what"] and "gather" in rec["tail_ms"]["what"]
    assert rec["config"]["oracle_checked_rows"] == 1_000_000 and rec["config"]["hits_last_step"] > 0
    assert rec["scaling_curve_point"]["source"] == "value" and rec["scaling_curve_point"]["kmers_per_s"] == rec["value"]


@pytest.mark.timeout(900)
def test_bench_product_paths_block_under_a_process_group():
    """What `bench.py --gpus N` adds at N > 1 -- the fused graph path (a rank uploads its shard of the graph) and the streamed
    scan under the run's process group -- driven with a one-rank RCCL group on the test box's one GPU: both blocks report rows
    and rates, no error string."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, GRAFIMO_RESERVE_CUS="16", MASTER_ADDR="127.0.0.1", MASTER_PORT="29579")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--config", "3", "--rows", "4000000",
                        "--steps", "6", "--warmup", "2", "--bursts", "2", "--no-cpu-baseline", "--no-e2e"],
                       capture_output=True, text=True, env=env, timeout=800)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][-1])
    pp = rec["product_paths"]
    assert "error" not in pp, pp
    gp, ss = pp["graph_path"], pp["streamed_scan"]
    assert gp["regions"] == 4000 and gp["rows"] > 2_000_000 and gp["rows_per_s"] > 1e8 and gp["sites_per_rank"] == [gp["sites_total"]]
    assert ss["rows"] == 800_000 and ss["files"] == 400 and ss["rows_per_s"] > 1e6 and ss["hits"] > 0


def test_kept_motif_handles(golden_motifs):
    """DeviceMotif.lease / release: the entry points that run once per motif and once more per chromosome keep their
    handle -- same numbers, same handle; other numbers, another; a plain handle is untouched by it; drop_kept() destroys
    what nobody holds."""
    from grafimo_amd.device import DeviceMotif
    DeviceMotif.drop_kept()
    motif = _ctcf(True)
    a = DeviceMotif.lease(motif)
    b = DeviceMotif.lease(motif)
    assert a is b and a.handle
    plain = DeviceMotif.from_motif(motif)
    assert plain is not a
    other = DeviceMotif.lease(_ctcf(False))              # no score distribution on the host: the DP runs on the device
    assert other is not a and other.handle
    pa, pb = a.tables()[1], other.tables()[1]
    np.testing.assert_allclose(pa, pb, rtol=1e-12)
    a.release(); b.release()
    DeviceMotif.drop_kept()                              # `other` is still held: only a / b go
    assert a.handle is None and other.handle
    again = DeviceMotif.lease(motif)
    assert again is not a and again.handle
    again.release(); other.release(); plain.release()    # release() of a plain handle closes it
    assert plain.handle is None
    DeviceMotif.drop_kept()
    assert again.handle is None and other.handle is None and not DeviceMotif._kept
