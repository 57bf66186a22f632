"""HIP path vs the CPU oracle / golden vectors -- through the C ABI, on a real MI355X."""
import os

import numpy as np
import pytest

from conftest import REF_DATA, kmers_from_strings

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from grafimo_amd import _native as nv
    assert os.path.exists(nv.LIB_PATH), "libgrafimo_hip.so not built"
    assert nv.device_count() >= 1
    return torch.device("cuda:0")


def random_kmers(rng, n, w, n_frac=0.001, lower_frac=0.05):
    km = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(n, w))
    if n:
        nn = rng.random(n) < n_frac
        pos = rng.integers(0, w, size=n)
        km[nn, pos[nn]] = ord("N")
        low = rng.random(n) < lower_frac
        km[low] |= 0x20  # lowercase (turns N into n on a few rows as well)
        # the reference leaves lowercase 'n' undefined; keep inputs inside its domain
        km[km == ord("n")] = ord("N")
    return km


def test_dp_bit_exact_all_golden_motifs(dev, golden_motifs):
    from grafimo_amd.device import comp_pval_mat_dense
    _, flat = golden_motifs
    for key, m in flat.items():
        pmf = comp_pval_mat_dense(m["score_matrix"], m["bg"])
        assert np.array_equal(pmf, m["pmf"]), key


def test_host_log_odds_and_scaling(dev, golden_motifs):
    from grafimo_amd.device import compute_log_odds_dense, scale_pwm_dense
    _, flat = golden_motifs
    for key, m in flat.items():
        lo = compute_log_odds_dense(m["probs"], m["bg"])
        np.testing.assert_allclose(lo, m["logodds"], rtol=1e-14, atol=0)
        sm, mn, mx, sc, off = scale_pwm_dense(lo)
        assert (sm == m["score_matrix"]).all(), key
        assert (mn, mx, sc, float(off)) == (m["min_val"], m["max_val"], m["scale"], m["offset"])


def test_ptable_matches_per_row_tail_sums(dev, golden_motifs):
    from grafimo_amd.device import DeviceMotif
    from oracle import oracle as orc
    _, flat = golden_motifs
    lib = orc.lib()
    for key in ["ctcf_meme_unif#0", "syn30_jaspar_bg0#0", "multi_meme_bg1#0", "gata1_meme_bgnt#0"]:
        m = flat[key]
        dm = DeviceMotif(m["score_matrix"], m["bg"], m["min_val"], m["scale"], m["offset"])
        pmf, pt = dm.tables()
        assert np.array_equal(pmf, m["pmf"])
        assert (pt[: dm.score_lo + 1] == 1.0).all()
        assert (np.diff(pt) <= 0).all()
        tot = lib.orc_np_sum(pmf.ctypes.data_as(orc._c_dp), len(pmf))
        rng = np.random.default_rng(7)
        for s in rng.integers(dm.score_lo, dm.score_hi + 1, size=200):
            tail = lib.orc_np_sum(pmf[s:].ctypes.data_as(orc._c_dp), len(pmf) - int(s))
            # tolerance: north_star asks 1e-6 on floats; the tables agree to ~1e-15
            assert abs(pt[s] - tail / tot) <= 1e-12 * max(pt[s], 1e-300), (key, s)
        dm.close()


@pytest.mark.parametrize("key", ["ctcf_meme_unif#0", "syn30_jaspar_bg0#0", "multi_meme_bg1#0",
                                 "multi_meme_bg1#2", "multi_meme_bg1#5", "gata1_meme_bgnt#0"])
def test_scores_and_histogram_vs_oracle(dev, golden_motifs, key):
    from grafimo_amd.device import DeviceMotif
    from oracle import oracle as orc
    _, flat = golden_motifs
    m = flat[key]
    W = m["width"]
    dm = DeviceMotif(m["score_matrix"], m["bg"], m["min_val"], m["scale"], m["offset"], m["pmf"])
    _, pt = dm.tables()
    rng = np.random.default_rng(W)
    for n in [0, 1, 63, 64, 65, 127, 128, 129, 1000, 4097, 100003]:
        km = random_kmers(rng, n, W, n_frac=0.01)
        d_km = torch.from_numpy(km).to(dev) if n else torch.empty((0, W), dtype=torch.uint8, device=dev)
        d_sc = torch.full((n,), -7, dtype=torch.int32, device=dev)
        d_hist = torch.zeros(dm.L, dtype=torch.int64, device=dev)
        dm.score(d_km, d_sc, hist=d_hist)
        torch.cuda.synchronize()
        got = d_sc.cpu().numpy()
        exp, _ = orc.score_kmers_table(km, m["score_matrix"], pt, m["min_val"]) if n else (np.empty(0, np.int32), None)
        assert np.array_equal(got, exp), (key, n)
        hist = d_hist.cpu().numpy()
        assert np.array_equal(hist, np.bincount(exp, minlength=dm.L)), (key, n)
        # accumulate semantics: a second pass doubles the histogram
        if n:
            dm.score(d_km, d_sc, hist=d_hist)
            torch.cuda.synchronize()
            assert np.array_equal(d_hist.cpu().numpy(), 2 * hist)
    dm.close()


def test_fused_and_separate_selection(dev, golden_motifs):
    from grafimo_amd.device import DeviceMotif
    _, flat = golden_motifs
    m = flat["ctcf_meme_unif#0"]
    dm = DeviceMotif(m["score_matrix"], m["bg"], m["min_val"], m["scale"], m["offset"], m["pmf"])
    rng = np.random.default_rng(3)
    n = 300001
    km = random_kmers(rng, n, 19)
    d_km = torch.from_numpy(km).to(dev)
    d_sc = torch.empty(n, dtype=torch.int32, device=dev)
    for thr in [1e-1, 1e-2, 1e-3]:
        cut = dm.pvalue_cutoff(thr)
        rows = torch.empty(n, dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        dm.score(d_km, d_sc, select_cutoff=cut, row_base=1000, hit_rows=rows, hit_count=cnt)
        torch.cuda.synchronize()
        sc = d_sc.cpu().numpy()
        exp = np.nonzero(sc >= cut)[0] + 1000
        k = int(cnt.item())
        assert k == len(exp)
        got = np.sort(rows[:k].cpu().numpy())
        assert np.array_equal(got >> 20, exp)
        assert np.array_equal(got & 0xFFFFF, sc[exp - 1000])
        # separate pass with a device-side cutoff
        rows2 = torch.empty(n, dtype=torch.int64, device=dev)
        cnt2 = torch.zeros(1, dtype=torch.int64, device=dev)
        d_cut = torch.tensor([cut], dtype=torch.int32, device=dev)
        dm.select_hits(d_sc, d_cut, rows2, cnt2, row_base=1000)
        torch.cuda.synchronize()
        assert int(cnt2.item()) == k
        assert np.array_equal(np.sort(rows2[:k].cpu().numpy()), got)
        # capacity overflow: counted, not stored past the end
        small = torch.full((8,), -1, dtype=torch.int64, device=dev)
        cnt3 = torch.zeros(1, dtype=torch.int64, device=dev)
        dm.select_hits(d_sc, d_cut, small, cnt3)
        torch.cuda.synchronize()
        assert int(cnt3.item()) == k
    dm.close()


def test_selecting_and_plain_calls_alternate_on_one_handle(dev, golden_motifs):
    """a selecting call that flushes its wave queues mid-run (threshold 1.0: every
    row is a hit), then a call that selects nothing (the score pass of --qvalueT), then selecting calls
    again on the same handle -- every hit list must be complete and hold no stale count."""
    from grafimo_amd.device import DeviceMotif
    _, flat = golden_motifs
    m = flat["ctcf_meme_unif#0"]
    dm = DeviceMotif(m["score_matrix"], m["bg"], m["min_val"], m["scale"], m["offset"], m["pmf"])
    rng = np.random.default_rng(11)
    n = 200_000
    d_km = torch.from_numpy(random_kmers(rng, n, 19)).to(dev)
    d_sc = torch.empty(n, dtype=torch.int32, device=dev)
    d_hist = torch.zeros(dm.L, dtype=torch.int64, device=dev)
    for stream in (None, torch.cuda.Stream(device=dev)):          # one stream, then a separate tail stream
        for round_ in range(3):
            for thr, select in [(1.0, True), (None, False), (None, False), (1e-2, True), (1.0, True), (None, False),
                                (1e-3, True)]:
                if not select:
                    dm.score(d_km, d_sc, hist=d_hist, tail_stream=stream)
                    continue
                cut = dm.pvalue_cutoff(thr)
                rows = torch.full((n,), -1, dtype=torch.int64, device=dev)
                cnt = torch.zeros(1, dtype=torch.int64, device=dev)
                dm.score(d_km, d_sc, select_cutoff=cut, hit_rows=rows, hit_count=cnt, reset_hits=True,
                         tail_stream=stream)
                torch.cuda.synchronize()
                sc = d_sc.cpu().numpy()
                exp = np.nonzero(sc >= cut)[0]
                assert int(cnt.item()) == len(exp), (thr, round_)
                got = np.sort(rows[:len(exp)].cpu().numpy())
                assert np.array_equal(got >> 20, exp) and np.array_equal(got & 0xFFFFF, sc[exp])
    dm.close()


def test_qvalue_table_vs_sorted_bh(dev, golden_motifs):
    from grafimo_amd.device import DeviceMotif
    from oracle import oracle as orc
    _, flat = golden_motifs
    for key, n in [("ctcf_meme_unif#0", 50000), ("multi_meme_bg1#0", 20000), ("syn30_jaspar_bg0#0", 3000)]:
        m = flat[key]
        dm = DeviceMotif(m["score_matrix"], m["bg"], m["min_val"], m["scale"], m["offset"], m["pmf"])
        _, pt = dm.tables()
        rng = np.random.default_rng(n)
        km = random_kmers(rng, n, m["width"], n_frac=0.01)
        d_km = torch.from_numpy(km).to(dev)
        d_sc = torch.empty(n, dtype=torch.int32, device=dev)
        d_hist = torch.zeros(dm.L, dtype=torch.int64, device=dev)
        dm.score(d_km, d_sc, hist=d_hist)
        q = torch.empty(dm.L, dtype=torch.float64, device=dev)
        cut = torch.zeros(1, dtype=torch.int32, device=dev)
        nrows = torch.zeros(1, dtype=torch.int64, device=dev)
        for on_q, thr in [(False, 1e-2), (True, 0.5), (True, 1.0), (False, 1.0)]:
            dm.qvalue_table(d_hist, thr, on_q, q, cut, nrows)
            torch.cuda.synchronize()
            sc = d_sc.cpu().numpy()
            pv = pt[sc]
            q_exp = orc.fdr_bh(pv)
            q_got = q.cpu().numpy()[sc]
            np.testing.assert_allclose(q_got, q_exp, rtol=1e-12, atol=0)
            assert int(nrows.item()) == n
            val = q_got if on_q else pv
            exp_hits = val < thr
            assert np.array_equal(sc >= int(cut.item()), exp_hits), (key, on_q, thr)
        dm.close()


def test_selection_from_candidates_equals_selection_from_scores(dev, golden_motifs):
    """gfm_select_hits_from: with a q-value threshold the score kernel collects the rows with p < t (q >= p) and
    the selection filters that list instead of reading every score.  Same hit list as gfm_select_hits for a
    complete candidate list, for one that overflowed (the device falls back to the scores), for a cutoff nothing
    reaches and for one everything reaches; the scanner's --qvalueT path equals the oracle either way."""
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.scan import KmerScanner
    from oracle import oracle as orc
    _, flat = golden_motifs
    m = flat["ctcf_meme_unif#0"]
    dm = DeviceMotif(m["score_matrix"], m["bg"], m["min_val"], m["scale"], m["offset"], m["pmf"])
    rng = np.random.default_rng(3)
    n = 300_001
    km = random_kmers(rng, n, 19, n_frac=0.01)
    d_km = torch.from_numpy(km).to(dev)
    d_sc = torch.empty(n + 3, dtype=torch.int32, device=dev)[:n]
    low = dm.pvalue_cutoff(0.05)                      # candidates: ~5 % of the rows
    for cap in (n, 4096):                             # complete / overflowing candidate list
        cand = torch.zeros(cap + 1, dtype=torch.int64, device=dev)
        dm.score(d_km, d_sc, select_cutoff=low, hit_rows=cand[1:], hit_count=cand[:1], reset_hits=True)
        torch.cuda.synchronize()
        sc = d_sc.cpu().numpy()
        assert (int(cand[0].item()) > cap) == (cap == 4096)
        for cut in (low, low + 40, low + 400, int(sc.max()) + 1, low - 1 if cap == 4096 else low):
            d_cut = torch.tensor([cut], dtype=torch.int32, device=dev)
            want = torch.full((n + 1,), -1, dtype=torch.int64, device=dev)
            dm.select_hits(d_sc, d_cut, want[1:], want[:1], row_base=7, reset_hits=True)
            got = torch.full((n + 1,), 5, dtype=torch.int64, device=dev)     # stale count and entries
            dm.select_hits_from(d_sc, d_cut, cand[1:], cand[:1], got[1:], got[:1], row_base=7)
            torch.cuda.synchronize()
            k = int(want[0].item())
            assert int(got[0].item()) == k == int((sc >= cut).sum())
            if cap == n:      # candidates carry row_base 0 (they were collected with it); rows differ by the base
                a = np.sort(got[1:1 + k].cpu().numpy()); b = np.sort(want[1:1 + k].cpu().numpy())
                assert np.array_equal(a + (7 << 20), b)
            else:
                assert np.array_equal(np.sort(got[1:1 + k].cpu().numpy()), np.sort(want[1:1 + k].cpu().numpy()))
    # the scanner on a q-value threshold, hit capacity too small for the candidates but enough for the hits
    ptab = orc.p_table(m["pmf"])
    exp_sc, p = orc.score_kmers_table(km, m["score_matrix"], ptab, m["min_val"])
    q = orc.fdr_bh(p)
    for cap in (n, 2000):
        scn = KmerScanner(dm, n, hit_capacity=cap, device=dev)
        for thr in (0.9, 0.05):
            keep = np.nonzero(q < thr)[0]
            if len(keep) > cap:
                continue
            res = scn.collect(scn.enqueue(d_km, thr, on_qvalue=True))
            assert np.array_equal(res["rows"], keep) and np.array_equal(res["scaled"], exp_sc[keep]), (cap, thr)
    dm.close()


def test_qvalue_tables_of_a_motif_set_in_one_call(dev, golden_motifs):
    """gfm_qvalue_table_multi (three launches per eight motifs) against one gfm_qvalue_table call per motif:
    eleven motifs of four widths -- more than one group of eight -- with optional outputs left out for some;
    every table, cutoff and row count must be bit-identical, and the histograms come back cleared."""
    from grafimo_amd.device import DeviceMotif, qvalue_table_multi
    _, flat = golden_motifs
    keys = ["ctcf_meme_unif#0", "atf3_meme_unif#0", "syn30_jaspar_bg0#0", "gata1_meme_bgnt#0", "multi_meme_bg1#0"]
    keys = [k for k in keys if k in flat] or list(flat)[:4]
    dms, hists = [], []
    rng = np.random.default_rng(77)
    for i in range(11):
        m = flat[keys[i % len(keys)]]
        dm = DeviceMotif(m["score_matrix"], m["bg"], m["min_val"], m["scale"], m["offset"], m["pmf"])
        n = 3000 + 1500 * i
        d_km = torch.from_numpy(random_kmers(rng, n, m["width"], n_frac=0.01 if i % 3 else 0.0)).to(dev)
        d_sc = torch.empty(n, dtype=torch.int32, device=dev)
        h = torch.zeros(dm.L, dtype=torch.int64, device=dev)
        dm.score(d_km, d_sc, hist=h)
        dms.append(dm); hists.append(h)
    for on_q, thr in [(False, 1e-2), (True, 0.5)]:
        single = []
        for dm, h in zip(dms, hists):
            q = torch.empty(dm.L, dtype=torch.float64, device=dev)
            cut = torch.zeros(1, dtype=torch.int32, device=dev); nr = torch.zeros(1, dtype=torch.int64, device=dev)
            dm.qvalue_table(h, thr, on_q, q, cut, nr)
            single.append((q, cut, nr))
        work = [h.clone() for h in hists]
        qs = [torch.empty(dm.L, dtype=torch.float64, device=dev) if i != 4 else None for i, dm in enumerate(dms)]
        cuts = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in dms]
        nrs = [torch.zeros(1, dtype=torch.int64, device=dev) if i != 9 else None for i in range(len(dms))]
        qvalue_table_multi(dms, work, thr, on_q, qs, cuts, nrs, clear_hist=True)
        torch.cuda.synchronize()
        for i, (q, cut, nr) in enumerate(single):
            if qs[i] is not None:
                assert torch.equal(qs[i], q), (i, on_q)
            assert int(cuts[i].item()) == int(cut.item())
            if nrs[i] is not None:
                assert int(nrs[i].item()) == int(nr.item())
            assert int(work[i].abs().sum().item()) == 0
    with pytest.raises(Exception):
        qvalue_table_multi([dms[0], dms[0]], [hists[0], hists[1]], 0.1, False)
    for dm in dms:
        dm.close()


def test_reference_704_row_fixture_through_scan_host(dev, golden_motifs, golden_json):
    """BASELINE config 1: the reference's own scoring fixture, GPU path end to end
    (numeric core; the DataFrame layer is covered by test_compute_results_gpu)."""
    from grafimo_amd.device import DeviceMotif
    from oracle import oracle as orc
    _, flat = golden_motifs
    m = flat["ctcf_meme_unif#0"]
    cols = orc.parse_tsv_rows([os.path.join(REF_DATA, "width_19", "scoring_test_input.tsv")])
    km = kmers_from_strings(cols["seq"])
    dm = DeviceMotif(m["score_matrix"], m["bg"], m["min_val"], m["scale"], m["offset"])  # device DP
    res = dm.scan_host(km, 1.0, on_qvalue=False, want_qvalues=True)
    assert len(res["rows"]) == 704 and np.array_equal(res["rows"], np.arange(704))
    exp = golden_json("compute_results.json")["recomb_t1"]["df"]
    ecols = exp["columns"]
    erows = {(r[ecols.index("start")], r[ecols.index("stop")], r[ecols.index("strand")],
              r[ecols.index("matched_sequence")]): r for r in exp["rows"]}
    for i in range(704):
        r = erows[(cols["start"][i], cols["stop"][i], cols["strand"][i], cols["seq"][i])]
        assert res["logodds"][i] == r[ecols.index("score")]
        assert abs(res["pvalue"][i] - r[ecols.index("p-value")]) <= 1e-12 * r[ecols.index("p-value")]
        assert abs(res["qvalue"][i] - r[ecols.index("q-value")]) <= 1e-12 * r[ecols.index("q-value")]
    # q-value threshold mode selects the same rows as the reference
    expq = golden_json("compute_results.json")["qvalt_t0.6"]["df"]
    resq = dm.scan_host(km, 0.6, on_qvalue=True, want_qvalues=True)
    # reference applied --recomb=False there: drop freq == 0 rows on the host
    keep = [i for i in resq["rows"] if cols["freq"][i] > 0]
    assert len(keep) == len(expq["rows"])
    dm.close()


def test_full_size_config2_scores_exact(dev, golden_motifs):
    """BASELINE config 2 size (2e7 k-mers, W=19): every integer score equals the CPU
    restatement's, the histogram is the bincount, totals are conserved."""
    from grafimo_amd.device import DeviceMotif
    from oracle import oracle as orc
    _, flat = golden_motifs
    m = flat["ctcf_meme_unif#0"]
    dm = DeviceMotif(m["score_matrix"], m["bg"], m["min_val"], m["scale"], m["offset"], m["pmf"])
    _, pt = dm.tables()
    n = 20_000_000
    rng = np.random.default_rng(20240139 + 2)
    km = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(n, 19))
    km[rng.integers(0, n, size=n // 1000), rng.integers(0, 19, size=n // 1000)] = ord("N")
    d_km = torch.from_numpy(km).to(dev)
    d_sc = torch.empty(n, dtype=torch.int32, device=dev)
    d_hist = torch.zeros(dm.L, dtype=torch.int64, device=dev)
    dm.score(d_km, d_sc, hist=d_hist)
    torch.cuda.synchronize()
    got = d_sc.cpu().numpy()
    exp, _ = orc.score_kmers_table(km, m["score_matrix"], pt, m["min_val"])
    assert np.array_equal(got, exp)
    hist = d_hist.cpu().numpy()
    assert hist.sum() == n
    assert np.array_equal(hist, np.bincount(exp, minlength=dm.L))
    dm.close()


def test_config4_shape_w30_qvalue_threshold(dev, golden_motifs):
    """BASELINE config 4 shape (W=30 JASPAR-style motif, both strands, --qvalueT -t 1e-4), scaled to
    2e6 rows so that the CPU side (C restatement + sorted BH) finishes in seconds."""
    from grafimo_amd import synth
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.scan import KmerScanner
    from oracle import oracle as orc
    _, flat = golden_motifs
    g = flat["syn30_jaspar_unif#0"]
    batch = synth.make_batch(1000, 2000, 30, g["probs"], synth.seed_for(4))
    n = len(batch)
    dm = DeviceMotif(g["score_matrix"], g["bg"], g["min_val"], g["scale"], g["offset"])   # device DP
    pmf, pt = dm.tables()
    assert np.array_equal(pmf, g["pmf"])
    sc_exp, pv = orc.score_kmers_table(batch.kmers, g["score_matrix"], pt, g["min_val"])
    q_exp = orc.fdr_bh(pv)
    scanner = KmerScanner(dm, n, device=dev)
    d_k = torch.from_numpy(batch.kmers).to(dev)
    for thr in (1e-4, 1e-2):
        slot = scanner.enqueue(d_k, thr, on_qvalue=True, want_qvalues=True)
        res = scanner.collect(slot)
        assert np.array_equal(slot.scores.cpu().numpy(), sc_exp)
        hits = np.nonzero(q_exp < thr)[0]
        assert len(hits) > 100
        assert np.array_equal(res["rows"], hits)
        assert np.array_equal(res["scaled"], sc_exp[hits])
        np.testing.assert_allclose(res["qtable"][res["scaled"]], q_exp[hits], rtol=1e-12, atol=0)
        assert res["n_scored"] == n
    dm.close()


def test_config5_shape_multi_motif_per_motif_background(dev, golden_motifs):
    """BASELINE config 5 shape: several PWMs of different widths (8..25), each with its own
    background, scanned one after the other the way grafimo.findmotif loops over a MotifSet
    (grafimo.py:177-183)."""
    from grafimo_amd import synth
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.scan import KmerScanner
    from oracle import oracle as orc
    cases, flat = golden_motifs
    keys = [f"multi_meme_bg1#{i}" for i in range(6)] + ["multi_meme_bg2_norev#3", "gata1_meme_bgnt#0",
                                                         "atf3_meme_unif#0", "example_meme_unif#0"]
    for key in keys:
        g = flat[key]
        W = g["width"]
        batch = synth.make_batch(60, 1000, W, g["probs"], synth.seed_for(5) + W)
        dm = DeviceMotif(g["score_matrix"], g["bg"], g["min_val"], g["scale"], g["offset"])
        pmf, pt = dm.tables()
        assert np.array_equal(pmf, g["pmf"]), key                     # per-motif bg enters the DP
        sc_exp, pv = orc.score_kmers_table(batch.kmers, g["score_matrix"], pt, g["min_val"])
        scanner = KmerScanner(dm, len(batch), device=dev, side_stream=False)
        res = scanner.collect(scanner.enqueue(torch.from_numpy(batch.kmers).to(dev), 1e-3))
        hits = np.nonzero(pv < 1e-3)[0]
        assert np.array_equal(res["rows"], hits), key
        lo, p = dm.annotate(res["scaled"])
        assert np.array_equal(lo, sc_exp[hits] / g["scale"] + W * g["offset"]), key
        np.testing.assert_allclose(res["qtable"][res["scaled"]], orc.fdr_bh(pv)[hits], rtol=1e-12, atol=0)
        dm.close()


def test_config3_shard_size_invariants(dev, golden_motifs):
    """BASELINE config 3's per-GPU shard (1.25e8 k-mers, W=19; 2.4 GB of k-mers generated on the
    device).  Too big for a row-by-row CPU check, so: (1) a 2e6-row slice is compared with the CPU
    restatement exactly, (2) size-independent properties on the full batch -- the histogram sums to
    N and equals torch.bincount of the scores, hits == rows with score >= cutoff, the batch scored
    as two halves (row_base, appended hit list, accumulated histogram) equals the single launch."""
    from grafimo_amd.device import DeviceMotif
    from oracle import oracle as orc
    _, flat = golden_motifs
    g = flat["ctcf_meme_unif#0"]
    dm = DeviceMotif(g["score_matrix"], g["bg"], g["min_val"], g["scale"], g["offset"], g["pmf"])
    _, pt = dm.tables()
    n = 125_000_000
    gen = torch.Generator(device=dev)
    gen.manual_seed(20240139 + 3)
    alphabet = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    d_k = alphabet[torch.randint(0, 4, (n, 19), generator=gen, device=dev, dtype=torch.int64)]
    d_k[torch.randint(0, n, (n // 1000,), generator=gen, device=dev), 7] = ord("N")
    d_sc = torch.empty(n, dtype=torch.int32, device=dev)
    d_hist = torch.zeros(dm.L, dtype=torch.int64, device=dev)
    cut = dm.pvalue_cutoff(1e-4)
    cap = n // 1000
    hits = torch.zeros(cap + 1, dtype=torch.int64, device=dev)
    dm.score(d_k, d_sc, hist=d_hist, select_cutoff=cut, hit_rows=hits[1:], hit_count=hits[:1], reset_hits=True)
    torch.cuda.synchronize()
    # (1) slice vs CPU
    lo_, hi_ = 61_000_000, 63_000_000
    exp, _ = orc.score_kmers_table(d_k[lo_:hi_].cpu().numpy(), g["score_matrix"], pt, g["min_val"])
    assert np.array_equal(d_sc[lo_:hi_].cpu().numpy(), exp)
    # (2) invariants
    assert int(d_hist.sum().item()) == n
    assert torch.equal(d_hist, torch.bincount(d_sc, minlength=dm.L))
    k = int(hits[0].item())
    assert 0 < k <= cap
    got = torch.sort(hits[1:1 + k]).values
    exp_rows = torch.nonzero(d_sc >= cut).flatten()
    assert torch.equal(got >> 20, exp_rows)
    assert torch.equal((got & 0xFFFFF).to(torch.int32), d_sc[exp_rows])
    # two halves == one launch
    h2 = torch.zeros(dm.L, dtype=torch.int64, device=dev)
    hits2 = torch.zeros(cap + 1, dtype=torch.int64, device=dev)
    half = 62_500_037   # ragged split
    sc2 = torch.empty(n, dtype=torch.int32, device=dev)
    dm.score(d_k[:half], sc2[:half], hist=h2, select_cutoff=cut, hit_rows=hits2[1:], hit_count=hits2[:1],
             reset_hits=True)
    # the second half's k-mer matrix starts at a 16-byte aligned row: 62_500_037 * 19 is not, so copy
    second = d_k[half:].clone()
    dm.score(second, sc2[half:], hist=h2, select_cutoff=cut, row_base=half, hit_rows=hits2[1:],
             hit_count=hits2[:1])
    torch.cuda.synchronize()
    assert torch.equal(h2, d_hist) and torch.equal(sc2, d_sc)
    assert int(hits2[0].item()) == k
    assert torch.equal(torch.sort(hits2[1:1 + k]).values, got)
    dm.close()


@pytest.mark.parametrize("W", [1, 2, 3, 4, 5, 7, 8, 16, 32, 33, 40, 48, 63, 64])
def test_width_sweep_random_matrices(dev, W):
    """Every kernel instantiation (NDW = 1..16), incl. score ranges too wide for the LDS histogram
    (W >= 40 with full-range columns -> partial LDS window + global spill counters): DP bit-exact,
    scores/histogram exact, q-table and selection vs sorted BH, on random integer matrices."""
    from grafimo_amd.device import DeviceMotif
    from oracle import oracle as orc
    rng = np.random.default_rng(1000 + W)
    sm = rng.integers(0, 1001, size=(4, W)).astype(np.int64)
    sm[rng.integers(0, 4), 0] = 0
    sm[rng.integers(0, 4), W - 1] = 1000          # keep the [0, 1000] span of scale_pwm
    if W >= 40:                                    # force a window wider than any LDS budget
        sm[0, :] = 0
        sm[3, :] = 1000
    bg = rng.dirichlet([20, 20, 20, 20])
    min_val, scale, offset = int(sm.min()), int(rng.integers(20, 200)), float(-rng.integers(3, 20))
    dm = DeviceMotif(sm, bg, min_val, scale, offset)
    pmf, pt = dm.tables()
    assert np.array_equal(pmf, orc.comp_pval_mat(sm, bg))
    assert dm.score_lo == int(sm.min(0).sum()) and dm.score_hi == int(sm.max(0).sum())
    n = 50_017
    km = random_kmers(rng, n, W, n_frac=0.02)
    exp, pv = orc.score_kmers_table(km, sm, pt, min_val)
    d_k = torch.from_numpy(km).to(dev)
    d_sc = torch.empty(n, dtype=torch.int32, device=dev)
    d_hist = torch.zeros(dm.L, dtype=torch.int64, device=dev)
    thr = 0.05
    cut = dm.pvalue_cutoff(thr)
    hits = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    dm.score(d_k, d_sc, hist=d_hist, select_cutoff=cut, hit_rows=hits[1:], hit_count=hits[:1], reset_hits=True)
    q = torch.empty(dm.L, dtype=torch.float64, device=dev)
    dcut = torch.zeros(1, dtype=torch.int32, device=dev)
    nrows = torch.zeros(1, dtype=torch.int64, device=dev)
    dm.qvalue_table(d_hist, 0.5, True, q, dcut, nrows)
    torch.cuda.synchronize()
    assert np.array_equal(d_sc.cpu().numpy(), exp)
    assert np.array_equal(d_hist.cpu().numpy(), np.bincount(exp, minlength=dm.L))
    k = int(hits[0].item())
    got = np.sort(hits[1:1 + k].cpu().numpy())
    assert np.array_equal(got >> 20, np.nonzero(pv < thr)[0])
    q_exp = orc.fdr_bh(pv)
    np.testing.assert_allclose(q.cpu().numpy()[exp], q_exp, rtol=1e-12, atol=0)
    assert np.array_equal(exp >= int(dcut.item()), q_exp < 0.5)
    lo, p = dm.annotate(exp[:100])
    assert np.array_equal(lo, exp[:100] / scale + W * offset) and np.array_equal(p, pt[exp[:100]])
    # the workspaces (slabs, spill counters) come back clean: three more calls accumulate exactly
    h3 = torch.zeros(dm.L, dtype=torch.int64, device=dev)
    for _ in range(3):
        dm.score(d_k, d_sc, hist=h3)
    torch.cuda.synchronize()
    assert np.array_equal(h3.cpu().numpy(), 3 * np.bincount(exp, minlength=dm.L))
    dm.close()


def test_batched_same_width_motifs_share_one_read(dev, golden_motifs):
    """gfm_score_kmers_multi (BASELINE config 5: several PWMs of one width, each with its own
    background): scores, histograms and hit lists of every motif equal the single-motif launches
    and the CPU restatement, for 1..5 motifs per call (groups of <= 3 share one pass)."""
    from grafimo_amd.device import DeviceMotif, score_multi
    from oracle import oracle as orc
    rng = np.random.default_rng(505)
    for W, n in [(19, 70_001), (8, 30_000), (25, 20_003)]:
        motifs, mats = [], []
        for k in range(5):
            sm = rng.integers(0, 1001, size=(4, W)).astype(np.int64)
            sm[rng.integers(0, 4), 0] = 0
            sm[rng.integers(0, 4), W - 1] = 1000
            bg = rng.dirichlet([30, 20, 20, 30])
            mats.append((sm, bg))
            motifs.append(DeviceMotif(sm, bg, int(sm.min()), 50 + k, -10.0 - k))
        km = random_kmers(rng, n, W, n_frac=0.01)
        d_k = torch.from_numpy(km).to(dev)
        for M in (1, 2, 3, 5):
            ms = motifs[:M]
            scores = [torch.empty(n, dtype=torch.int32, device=dev) for _ in ms]
            hists = [torch.zeros(m.L, dtype=torch.int64, device=dev) for m in ms]
            hits = [torch.zeros(n + 1, dtype=torch.int64, device=dev) for _ in ms]
            cuts = [m.pvalue_cutoff(0.02) for m in ms]
            if M == 3:
                cuts[1] = None                      # one motif without selection
            score_multi(ms, d_k, scores, hists=hists, cutoffs=cuts, row_base=7,
                        hit_rows=[h[1:] for h in hits], hit_counts=[h[:1] for h in hits], reset_hits=True)
            torch.cuda.synchronize()
            for j, m in enumerate(ms):
                _, pt = m.tables()
                exp, pv = orc.score_kmers_table(km, mats[j][0], pt, int(mats[j][0].min()))
                assert np.array_equal(scores[j].cpu().numpy(), exp), (W, M, j)
                assert np.array_equal(hists[j].cpu().numpy(), np.bincount(exp, minlength=m.L)), (W, M, j)
                k = int(hits[j][0].item())
                if cuts[j] is None:
                    assert k == 0
                    continue
                got = np.sort(hits[j][1:1 + k].cpu().numpy())
                rows = np.nonzero(exp >= cuts[j])[0]
                assert np.array_equal(got >> 20, rows + 7) and np.array_equal(got & 0xFFFFF, exp[rows])
        # a second batched call appends to the lists and accumulates the histograms
        score_multi(motifs[:2], d_k, scores[:2], hists=hists[:2], cutoffs=[cuts[0], cuts[0]], row_base=n + 7,
                    hit_rows=[h[1:] for h in hits[:2]], hit_counts=[h[:1] for h in hits[:2]])
        torch.cuda.synchronize()
        _, pt = motifs[0].tables()
        exp, _ = orc.score_kmers_table(km, mats[0][0], pt, int(mats[0][0].min()))
        assert np.array_equal(hists[0].cpu().numpy(), 2 * np.bincount(exp, minlength=motifs[0].L))
        assert int(hits[0][0].item()) == 2 * int((exp >= cuts[0]).sum())
        with pytest.raises(Exception):
            score_multi([motifs[0], motifs[0]], d_k, scores[:2])           # same motif twice
        for m in motifs:
            m.close()
    a = DeviceMotif(np.zeros((4, 5), np.int64) + 3, [0.25] * 4, 3, 10, -1.0)
    b = DeviceMotif(np.zeros((4, 6), np.int64) + 3, [0.25] * 4, 3, 10, -1.0)
    with pytest.raises(Exception):                                          # widths differ
        score_multi([a, b], torch.zeros((16, 5), dtype=torch.uint8, device=dev),
                    [torch.empty(16, dtype=torch.int32, device=dev)] * 2)


def test_scan_same_width_equals_single_motif_scans(dev, golden_motifs):
    from grafimo_amd import synth
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.scan import KmerScanner, scan_same_width
    _, flat = golden_motifs
    # three W=19 motifs with different backgrounds / matrices
    keys = ["ctcf_meme_unif#0", "ctcf_meme_bgnt#0", "multi_meme_bg1#3", "ctcf_jaspar_bgnt_p1#0"]
    gs = [flat[k] for k in keys]
    assert all(g["width"] == 19 for g in gs)
    dms = [DeviceMotif(g["score_matrix"], g["bg"], g["min_val"], g["scale"], g["offset"], g["pmf"]) for g in gs]
    batch = synth.make_batch(40, 1000, 19, gs[0]["probs"], synth.seed_for(5))
    d_k = torch.from_numpy(batch.kmers).to(dev)
    for on_q, thr in [(False, 1e-3), (True, 0.3)]:
        multi = scan_same_width(dms, d_k, thr, on_qvalue=on_q)
        for dm, res in zip(dms, multi):
            sc = KmerScanner(dm, len(batch), device=dev, side_stream=False)
            one = sc.collect(sc.enqueue(d_k, thr, on_qvalue=on_q))
            assert np.array_equal(res["rows"], one["rows"]) and np.array_equal(res["scaled"], one["scaled"])
            assert np.array_equal(res["qtable"], one["qtable"]) and res["n_scored"] == one["n_scored"]
        assert sum(len(r["rows"]) for r in multi) > 0
    # the resident form (one set of buffers, enqueued repeatedly), with and without its side stream; a hit
    # capacity the p < t candidates of the q-value threshold do not fit makes the device fall back to the scores
    from grafimo_amd.scan import SameWidthScanner
    n = len(batch)
    for side, cap in [(False, n), (True, n), (False, 600)]:
        sw = SameWidthScanner(dms, n, cap, dev, side_stream=side)
        for on_q, thr in [(False, 1e-3), (True, 0.3), (True, 0.3), (False, 1e-3)]:
            sw.enqueue(d_k, thr, on_qvalue=on_q)
            sw.finish()
            torch.cuda.synchronize()
            for j, dm in enumerate(dms):
                sc = KmerScanner(dm, n, device=dev, side_stream=False)
                one = sc.collect(sc.enqueue(d_k, thr, on_qvalue=on_q))
                k = int(sw.hits[j, 0].item())
                assert k == len(one["rows"])
                if k <= cap:
                    packed = np.sort(sw.hits[j, 1:1 + k].cpu().numpy())
                    assert np.array_equal(packed >> 20, one["rows"]) and np.array_equal(packed & 0xFFFFF, one["scaled"])
                assert np.array_equal(sw.qtable[j].cpu().numpy(), one["qtable"])
    for dm in dms:
        dm.close()


def test_single_stream_path_is_graph_capturable(dev, golden_motifs):
    """include/grafimo_hip.h promises that the stream-taking entry points only enqueue (no
    allocation, no synchronisation): capture two steps (both workspace parities) into a HIP graph,
    replay, compare with the eager results."""
    from grafimo_amd import synth
    from grafimo_amd.device import DeviceMotif
    _, flat = golden_motifs
    g = flat["ctcf_meme_unif#0"]
    dm = DeviceMotif(g["score_matrix"], g["bg"], g["min_val"], g["scale"], g["offset"], g["pmf"])
    batch = synth.make_batch(100, 1000, 19, g["probs"], synth.seed_for(2))
    n = len(batch)
    d = torch.from_numpy(batch.kmers).to(dev)
    sc = torch.empty(n, dtype=torch.int32, device=dev)
    hist = torch.zeros(dm.L, dtype=torch.int64, device=dev)
    hits = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    q = torch.empty(dm.L, dtype=torch.float64, device=dev)
    cut = torch.zeros(1, dtype=torch.int32, device=dev)
    nr = torch.zeros(1, dtype=torch.int64, device=dev)
    c = dm.pvalue_cutoff(1e-3)

    def step():
        dm.score(d, sc, hist=hist, select_cutoff=c, hit_rows=hits[1:], hit_count=hits[:1], reset_hits=True)
        dm.qvalue_table(hist, 1e-3, False, q, cut, nr, clear_hist=True)

    step(); step()
    torch.cuda.synchronize()
    ref = (sc.clone(), hits.clone().sort().values, q.clone(), int(nr))
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step(); step()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=s):
            step(); step()
    sc.zero_(); hits.zero_(); q.zero_()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(sc, ref[0]) and torch.equal(hits.sort().values, ref[1])
    assert torch.equal(q, ref[2]) and int(nr) == ref[3] == n
    dm.close()


def test_no_score_store_leaves_histogram_and_hits_unchanged(golden_motifs):
    """gfm_score_kmers / gfm_score_kmers_multi with d_scores == NULL (what the product's scans pass: with a threshold the
    cutoff is known before scoring, nothing ever reads the int32 [N] array): histograms and hit lists equal the ones of the
    call that stores the scores -- and the oracle's -- for 1, 2 and 3 motifs per launch, ragged row counts included."""
    import ctypes
    from grafimo_amd import _native as nv
    from grafimo_amd import synth
    from grafimo_amd.device import DeviceMotif
    from oracle import oracle as orc
    _, flat = golden_motifs
    g = flat["ctcf_meme_unif#0"]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    recs = [synth.synthetic_motif(19, rng, np.full(4, 0.25)) for _ in range(3)]
    dms = [DeviceMotif(r["sm"], r["bg"], r["min_val"], r["scale"], r["offset"]) for r in recs]
    for n in (1_000_003, 777, 256 * 50):
        batch = synth.make_batch(max(1, n // 2000 + 1), 2000, 19, g["probs"], synth.seed_for(3))
        km = np.ascontiguousarray(batch.kmers[:n])
        d_k = torch.from_numpy(km).to(dev)
        for M in (1, 2, 3):
            outs = []
            for store in (True, False):
                scores = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(M)] if store else None
                hists = [torch.zeros(dms[j].L, dtype=torch.int64, device=dev) for j in range(M)]
                hits = [torch.zeros(n, dtype=torch.int64, device=dev) for _ in range(M)]
                counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(M)]
                cuts = [dms[j].pvalue_cutoff(1e-2) for j in range(M)]
                vp = ctypes.c_void_p
                arr = lambda ts: (vp * M)(*[t.data_ptr() for t in ts])      # noqa: E731
                if M == 1:
                    nv.check(nv.lib().gfm_score_kmers(dms[0].handle, d_k.data_ptr(), n, scores[0].data_ptr() if store else None,
                                                      hists[0].data_ptr(), cuts[0], 0, hits[0].data_ptr(), n, counts[0].data_ptr(),
                                                      nv.GFM_FLAG_RESET_HITS, None, None))
                else:
                    nv.check(nv.lib().gfm_score_kmers_multi((vp * M)(*[d.handle for d in dms[:M]]), M, d_k.data_ptr(), n,
                                                            arr(scores) if store else None, arr(hists), (ctypes.c_int32 * M)(*cuts), 0,
                                                            arr(hits), (ctypes.c_int64 * M)(*([n] * M)), arr(counts),
                                                            nv.GFM_FLAG_RESET_HITS, None))
                torch.cuda.synchronize()
                outs.append(([h.cpu().numpy() for h in hists],
                             [np.sort(hits[j][:int(counts[j].item())].cpu().numpy()) for j in range(M)]))
            for j in range(M):
                assert np.array_equal(outs[0][0][j], outs[1][0][j]), (n, M, j)
                assert np.array_equal(outs[0][1][j], outs[1][1][j]) and len(outs[1][1][j]) > 0, (n, M, j)
                ptab = orc.p_table(orc.comp_pval_mat(recs[j]["sm"], recs[j]["bg"]))
                exp_sc, _ = orc.score_kmers_table(km, recs[j]["sm"], ptab, recs[j]["min_val"])
                assert np.array_equal(outs[1][0][j], np.bincount(exp_sc, minlength=dms[j].L)), (n, M, j)
    for d in dms:
        d.close()
    # nothing to do at all is refused
    dm = DeviceMotif(recs[0]["sm"], recs[0]["bg"], recs[0]["min_val"], recs[0]["scale"], recs[0]["offset"])
    d_k = torch.zeros((256, 19), dtype=torch.uint8, device=dev)
    assert nv.lib().gfm_score_kmers(dm.handle, d_k.data_ptr(), 256, None, None, nv.GFM_NO_SELECT, 0, None, 0, None, 0, None, None) == nv.GFM_ERR_INVALID
    dm.close()
