"""Every score_quad_kernel<W, MM> instantiation against the CPU oracle, and BASELINE configs 4 and 5 at full size.

The library holds 64 single-motif instantiations (W = 1..64) and 2 x 32 batched ones (MM = 2, 3 motifs of one width
sharing each read of the k-mers, W = 1..32).  The four strip layouts (W odd, W = 2 mod 4, W = 4 mod 8, W = 0 mod 8)
are different code, so every width is compared with oracle/grafimo_oracle.c (compute_score_seq,
score_sequences.py:331-396) -- scores, histograms, hit lists -- through the C ABI.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from grafimo_amd import _native as nv
    assert os.path.exists(nv.LIB_PATH), "libgrafimo_hip.so not built"
    return torch.device("cuda:0")


def random_kmers(rng, n, w, n_frac=0.01, lower_frac=0.05):
    km = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(n, w))
    if n:
        nn = rng.random(n) < n_frac
        pos = rng.integers(0, w, size=n)
        km[nn, pos[nn]] = ord("N")
        low = rng.random(n) < lower_frac
        km[low] |= 0x20
        km[km == ord("n")] = ord("N")     # lowercase n is outside the reference's domain
    return km


def band_matrix(rng, W, span):
    """int64 [4, W] with every column inside a band of `span` scores (so that the reachable range
    sum_j (max_j - min_j) + 1 stays small enough for whole LDS windows), 0 and a column maximum present."""
    span = int(max(1, min(1000, span)))
    base = rng.integers(0, 1001 - span, size=W)
    sm = base[None, :] + rng.integers(0, span + 1, size=(4, W))
    sm[rng.integers(0, 4), rng.integers(0, W)] = 0
    return sm.astype(np.int64)


def check_against_oracle(dev, motifs, mats, km, scores, hists, hits, cuts, row_base, tag):
    from oracle import oracle as orc
    for j, m in enumerate(motifs):
        _, pt = m.tables()
        exp, _ = orc.score_kmers_table(km, mats[j][0], pt, int(mats[j][0].min()))
        assert np.array_equal(scores[j].cpu().numpy(), exp), (tag, j, "scores")
        if hists[j] is not None:
            assert np.array_equal(hists[j].cpu().numpy(), np.bincount(exp, minlength=m.L)), (tag, j, "hist")
        k = int(hits[j][0].item())
        if cuts[j] is None:
            assert k == 0, (tag, j)
            continue
        got = np.sort(hits[j][1:1 + k].cpu().numpy())
        rows = np.nonzero(exp >= cuts[j])[0]
        assert k == len(rows), (tag, j, "hit count")
        assert np.array_equal(got >> 20, rows + row_base) and np.array_equal(got & 0xFFFFF, exp[rows]), (tag, j, "hits")


@pytest.mark.parametrize("MM", [2, 3])
@pytest.mark.parametrize("W", list(range(1, 33)))
def test_batched_kernel_every_width(dev, W, MM):
    """score_quad_kernel<W, MM>, MM = 2, 3, every W = 1..32: per-motif background, N and lowercase rows, a ragged
    row count, one motif without selection / without histogram.  The launch plan is asserted, so the MM-motif
    instantiation of this width is what ran."""
    from grafimo_amd.device import DeviceMotif, multi_plan, score_multi
    rng = np.random.default_rng(7000 + 100 * MM + W)
    motifs, mats = [], []
    for k in range(MM):
        sm = band_matrix(rng, W, 4200 // (W * MM) + 40 // W)
        bg = rng.dirichlet([30, 20, 20, 30])
        mats.append((sm, bg))
        motifs.append(DeviceMotif(sm, bg, int(sm.min()), 40 + k, -9.0 - k))        # DP on the device
    sizes, waves = multi_plan(motifs)
    assert list(sizes) == [MM] * MM, (W, MM, sizes, waves)
    n = 41_003 + 17 * W
    km = random_kmers(rng, n, W)
    d_k = torch.from_numpy(km).to(dev)
    for variant in range(2):
        scores = [torch.full((n,), -7, dtype=torch.int32, device=dev) for _ in motifs]
        hists = [torch.zeros(m.L, dtype=torch.int64, device=dev) for m in motifs]
        hits = [torch.zeros(n + 1, dtype=torch.int64, device=dev) for _ in motifs]
        cuts = [m.pvalue_cutoff(0.03) for m in motifs]
        if variant == 1:
            cuts[MM - 1] = None                      # one motif without selection
            hists[0] = None                          # one without histogram
            sizes, _ = multi_plan(motifs, [h is not None for h in hists])
            assert list(sizes) == [MM] * MM
        score_multi(motifs, d_k, scores, hists=hists, cutoffs=cuts, row_base=11,
                    hit_rows=[h[1:] for h in hits], hit_counts=[h[:1] for h in hits], reset_hits=True)
        torch.cuda.synchronize()
        check_against_oracle(dev, motifs, mats, km, scores, hists, hits, cuts, 11, (W, MM, variant))
    for m in motifs:
        m.close()


@pytest.mark.parametrize("W", [6, 10, 12, 14, 18, 20, 22, 24, 32])
def test_batched_kernel_wide_ranges_partial_windows(dev, W):
    """Full-range random matrices: the histogram windows of a group no longer fit whole, so the plan falls back to
    partial windows + the global spill counters, to 8 waves, or to smaller groups -- whatever it picks must equal
    the oracle (and the motifs of one call may ride in different group sizes)."""
    from grafimo_amd.device import DeviceMotif, multi_plan, score_multi
    rng = np.random.default_rng(9000 + W)
    motifs, mats = [], []
    for k in range(5):
        sm = rng.integers(0, 1001, size=(4, W)).astype(np.int64)
        sm[rng.integers(0, 4), 0] = 0
        sm[rng.integers(0, 4), W - 1] = 1000
        bg = rng.dirichlet([20, 20, 20, 20])
        mats.append((sm, bg))
        motifs.append(DeviceMotif(sm, bg, int(sm.min()), 50 + k, -10.0 - k))
    sizes, waves = multi_plan(motifs)
    assert sizes.sum() >= 5 and set(waves) <= {8, 16}
    n = 60_007
    km = random_kmers(rng, n, W)
    d_k = torch.from_numpy(km).to(dev)
    scores = [torch.empty(n, dtype=torch.int32, device=dev) for _ in motifs]
    hists = [torch.zeros(m.L, dtype=torch.int64, device=dev) for m in motifs]
    hits = [torch.zeros(n + 1, dtype=torch.int64, device=dev) for _ in motifs]
    cuts = [m.pvalue_cutoff(0.02) for m in motifs]
    score_multi(motifs, d_k, scores, hists=hists, cutoffs=cuts, row_base=0,
                hit_rows=[h[1:] for h in hits], hit_counts=[h[:1] for h in hits], reset_hits=True)
    torch.cuda.synchronize()
    check_against_oracle(dev, motifs, mats, km, scores, hists, hits, cuts, 0, (W, "wide"))
    for m in motifs:
        m.close()


@pytest.mark.parametrize("W", list(range(1, 65)))
def test_single_motif_kernel_every_width(dev, W):
    """score_quad_kernel<W, 1>, every W = 1..64, on random integer matrices (band-limited for even W so that whole
    windows are used there, full range for odd W: partial windows + spill for the wide ones): device DP bit-exact,
    scores / histogram / fused selection exact."""
    from grafimo_amd.device import DeviceMotif
    from oracle import oracle as orc
    rng = np.random.default_rng(3000 + W)
    if W % 2 == 0:
        sm = band_matrix(rng, W, 9000 // W)
    else:
        sm = rng.integers(0, 1001, size=(4, W)).astype(np.int64)
        sm[rng.integers(0, 4), 0] = 0
        sm[rng.integers(0, 4), W - 1] = 1000
    bg = rng.dirichlet([20, 20, 20, 20])
    min_val = int(sm.min())
    dm = DeviceMotif(sm, bg, min_val, int(rng.integers(20, 200)), float(-rng.integers(3, 20)))
    pmf, pt = dm.tables()
    assert np.array_equal(pmf, orc.comp_pval_mat(sm, bg))
    for n in (30_011 + W, 255, 256, 257):
        km = random_kmers(rng, n, W, n_frac=0.02)
        exp, pv = orc.score_kmers_table(km, sm, pt, min_val)
        d_k = torch.from_numpy(km).to(dev)
        d_sc = torch.full((n,), -3, dtype=torch.int32, device=dev)
        d_hist = torch.zeros(dm.L, dtype=torch.int64, device=dev)
        cut = dm.pvalue_cutoff(0.05)
        hits = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        dm.score(d_k, d_sc, hist=d_hist, select_cutoff=cut, row_base=5, hit_rows=hits[1:], hit_count=hits[:1],
                 reset_hits=True)
        torch.cuda.synchronize()
        assert np.array_equal(d_sc.cpu().numpy(), exp), (W, n)
        assert np.array_equal(d_hist.cpu().numpy(), np.bincount(exp, minlength=dm.L)), (W, n)
        k = int(hits[0].item())
        got = np.sort(hits[1:1 + k].cpu().numpy())
        rows = np.nonzero(exp >= cut)[0]
        assert np.array_equal(got >> 20, rows + 5) and np.array_equal(got & 0xFFFFF, exp[rows]), (W, n)
    dm.close()


# ------------------------------------------------------------------------------------------------ full size
def test_config4_full_size_invariants(dev):
    """BASELINE config 4 at full size: the synthetic W=30 PWM, 1.0e8 k-mers generated on the device (3 GB),
    --qvalueT 1e-4.  (1) a 2e6-row slice of the scores equals the CPU restatement; (2) the histogram equals
    torch.bincount of the scores and sums to N; (3) the hit list equals {rows with q < t}, q looked up in the
    q-table by score (q = min(1, min_{s'<=s} p(s') n / C(s')) recomputed here from the histogram in f64 as well);
    (4) the candidate path (p < t collected while scoring, filtered on q) equals the pass over every score."""
    from grafimo_amd import synth
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.scan import KmerScanner
    from oracle import oracle as orc
    m = synth.config_motifs(4)[0]
    dm = DeviceMotif(m["sm"], m["bg"], m["min_val"], m["scale"], m["offset"])
    pmf, pt = dm.tables()
    assert np.array_equal(pmf, orc.comp_pval_mat(m["sm"], m["bg"]))
    n = 100_000_000
    thr = 1e-4
    d_k = synth.make_device_kmers(n, 30, m["probs"], synth.seed_for(4), dev)
    res = {}
    for cand in (True, False):
        sc = KmerScanner(dm, n, hit_capacity=n // 64, device=dev, n_slots=2, candidates=cand)
        slot = sc.enqueue(d_k, thr, on_qvalue=True, want_qvalues=True, row_base=0)
        out = sc.collect(slot)
        res[cand] = (out, slot.scores.clone() if cand else slot.scores)
        if cand:
            # (1) slice vs the oracle
            lo_, hi_ = 48_000_000, 50_000_000
            exp, _ = orc.score_kmers_table(d_k[lo_:hi_].cpu().numpy(), m["sm"], pt, m["min_val"])
            assert np.array_equal(slot.scores[lo_:hi_].cpu().numpy(), exp)
            # (2) histogram (the scanner clears its own: rebuild it from a plain scoring call)
            d_hist = torch.zeros(dm.L, dtype=torch.int64, device=dev)
            d_sc2 = torch.empty(n, dtype=torch.int32, device=dev)
            dm.score(d_k, d_sc2, hist=d_hist)
            torch.cuda.synchronize()
            assert torch.equal(d_sc2, slot.scores)
            assert int(d_hist.sum().item()) == n
            assert torch.equal(d_hist, torch.bincount(d_sc2, minlength=dm.L))
            del d_sc2
            # (3) BH from the histogram, on the host in f64: q(s) = min(1, min_{s' <= s, hist > 0} p(s') n / C(s'))
            hist = d_hist.cpu().numpy()
            C = np.cumsum(hist[::-1])[::-1].astype(np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                raw = np.where(hist > 0, pt / (C / float(n)), np.inf)
            q_host = np.minimum(1.0, np.minimum.accumulate(raw))
            occupied = hist > 0
            np.testing.assert_allclose(out["qtable"][occupied], q_host[occupied], rtol=1e-12, atol=0)
            assert out["n_scored"] == n
            d_q = torch.from_numpy(out["qtable"]).to(dev)
            exp_rows = torch.nonzero(d_q[slot.scores.long()] < thr).flatten().cpu().numpy()
            assert len(exp_rows) > 1000
            assert np.array_equal(out["rows"], exp_rows)
            assert np.array_equal(out["scaled"], slot.scores[torch.from_numpy(exp_rows).to(dev)].cpu().numpy())
        del sc
    # (4) candidates == scores path
    a, b = res[True][0], res[False][0]
    assert np.array_equal(a["rows"], b["rows"]) and np.array_equal(a["scaled"], b["scaled"])
    assert np.array_equal(a["qtable"], b["qtable"])
    assert torch.equal(res[True][1], res[False][1])
    dm.close()


@pytest.mark.parametrize("W", [10, 20, 22])
def test_config5_full_size_batched_equals_single_and_oracle(dev, W):
    """BASELINE config 5 at full size for three of its widths -- 10 (W = 2 mod 4, three motifs per launch), 20
    (W = 4 mod 8, three motifs) and 22 (W = 2 mod 4, two motifs): 1.0e8 k-mers generated on the device, the config's
    own synthetic PWMs with their per-motif backgrounds.  The batched launch equals the single-motif launches
    (scores, histograms, hit lists), a 1e6-row slice equals the CPU restatement, histograms equal torch.bincount."""
    from grafimo_amd import synth
    from grafimo_amd.device import DeviceMotif, multi_plan, score_multi
    from oracle import oracle as orc
    mots = [m for m in synth.config_motifs(5) if m["width"] == W]
    assert len(mots) == (3 if W <= 21 else 2)
    dms = [DeviceMotif(m["sm"], m["bg"], m["min_val"], m["scale"], m["offset"]) for m in mots]
    sizes, _ = multi_plan(dms)
    assert list(sizes) == [len(mots)] * len(mots)         # the whole width rides in one launch
    n = 100_000_000
    d_k = synth.make_device_kmers(n, W, mots[0]["probs"], synth.seed_for(50 + W), dev)
    M = len(dms)
    cap = n // 64
    scores = [torch.empty(n, dtype=torch.int32, device=dev) for _ in dms]
    hists = [torch.zeros(d.L, dtype=torch.int64, device=dev) for d in dms]
    hits = [torch.zeros(cap + 1, dtype=torch.int64, device=dev) for _ in dms]
    cuts = [d.pvalue_cutoff(1e-4) for d in dms]
    score_multi(dms, d_k, scores, hists=hists, cutoffs=cuts, row_base=0,
                hit_rows=[h[1:] for h in hits], hit_counts=[h[:1] for h in hits], reset_hits=True)
    torch.cuda.synchronize()
    lo_, hi_ = 73_000_000, 74_000_000
    km = d_k[lo_:hi_].cpu().numpy()
    one_sc = torch.empty(n, dtype=torch.int32, device=dev)
    for j, (d, m) in enumerate(zip(dms, mots)):
        _, pt = d.tables()
        exp, _ = orc.score_kmers_table(km, m["sm"], pt, m["min_val"])
        assert np.array_equal(scores[j][lo_:hi_].cpu().numpy(), exp), (W, j)
        one_h = torch.zeros(d.L, dtype=torch.int64, device=dev)
        one_hits = torch.zeros(cap + 1, dtype=torch.int64, device=dev)
        d.score(d_k, one_sc, hist=one_h, select_cutoff=cuts[j], hit_rows=one_hits[1:], hit_count=one_hits[:1],
                reset_hits=True)
        torch.cuda.synchronize()
        assert torch.equal(one_sc, scores[j]), (W, j)
        assert torch.equal(one_h, hists[j]) and int(one_h.sum().item()) == n
        assert torch.equal(hists[j], torch.bincount(scores[j], minlength=d.L))
        k = int(hits[j][0].item())
        assert 0 < k <= cap and k == int(one_hits[0].item())
        got = torch.sort(hits[j][1:1 + k]).values
        assert torch.equal(got, torch.sort(one_hits[1:1 + k]).values)
        exp_rows = torch.nonzero(scores[j] >= cuts[j]).flatten()
        assert torch.equal(got >> 20, exp_rows)
    for d in dms:
        d.close()


@pytest.mark.parametrize("W", [40, 44, 57, 64])
def test_wide_motif_without_histogram_through_the_batched_entry_point(dev, W):
    """A motif too wide for 16 waves' strips, scored through gfm_score_kmers_multi WITHOUT a histogram: the launch plan
    must still step down to 8 waves (found by scripts/score_fuzz.py: with no window to place the plan kept 16 waves and
    asked for more LDS than a CU has)."""
    from grafimo_amd.device import DeviceMotif, multi_plan, score_multi
    rng = np.random.default_rng(4400 + W)
    sm = band_matrix(rng, W, 4200 // W)
    bg = rng.dirichlet([30, 20, 20, 30])
    m = DeviceMotif(sm, bg, int(sm.min()), 40, -9.0)
    sizes, waves = multi_plan([m], [False])
    assert list(sizes) == [1] and list(waves) == [8]
    n = 264_825
    km = random_kmers(rng, n, W)
    d_k = torch.from_numpy(km).to(dev)
    for with_hist in (False, True):
        scores = [torch.full((n,), -7, dtype=torch.int32, device=dev)]
        hists = [torch.zeros(m.L, dtype=torch.int64, device=dev) if with_hist else None]
        hits = [torch.zeros(n + 1, dtype=torch.int64, device=dev)]
        cuts = [m.pvalue_cutoff(0.03)]
        score_multi([m], d_k, scores, hists=hists, cutoffs=cuts, row_base=0,
                    hit_rows=[h[1:] for h in hits], hit_counts=[h[:1] for h in hits], reset_hits=True)
        torch.cuda.synchronize()
        check_against_oracle(dev, [m], [(sm, bg)], km, scores, hists, hits, cuts, 0, (W, with_hist))
    m.close()
