"""Pin the CPU oracle against the reference's golden vectors (CPU only)."""
import os

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN, REF_DATA, kmers_from_strings
from oracle import oracle as orc


def test_pairwise_sum_is_numpy_sum():
    rng = np.random.default_rng(1)
    lib = orc.lib()
    for n in [1, 5, 8, 9, 127, 128, 129, 1000, 19001, 30001, 12345]:
        a = rng.random(n) * 10.0 ** rng.integers(-12, 3, n)
        got = lib.orc_np_sum(a.ctypes.data_as(orc._c_dp), n)
        assert got == a.sum(), n


def test_log_odds_scale_dp_match_reference(golden_motifs):
    _, flat = golden_motifs
    for key, m in flat.items():
        lo = orc.compute_log_odds(m["probs"], m["bg"])
        np.testing.assert_allclose(lo, m["logodds"], rtol=1e-14, atol=0, err_msg=key)
        sm, mn, mx, sc, off = orc.scale_pwm(lo)
        assert (sm == m["score_matrix"]).all(), key
        assert (mn, mx, sc, float(off)) == (m["min_val"], m["max_val"], m["scale"], m["offset"]), key
        pmf = orc.comp_pval_mat(sm, m["bg"])
        assert pmf.shape == m["pmf"].shape
        assert np.array_equal(pmf, m["pmf"]), key  # bit-exact DP


def test_reference_golden_score_matrices(golden_motifs):
    """The reference's own known-answer files (tests/grafimo_run_test.py:68-116)."""
    cases, flat = golden_motifs
    meme = np.loadtxt(os.path.join(REF_DATA, "motif_processing_test_meme.txt")).astype(int)
    jasp = np.loadtxt(os.path.join(REF_DATA, "motif_processing_test_jaspar.txt")).astype(int)
    for key, exp in [("ctcf_meme_unif#0", meme), ("ctcf_jaspar_unif#0", jasp),
                     ("ctcf_transfac_unif#0", jasp), ("ctcf_pfm_unif#0", jasp)]:
        m = flat[key]
        sm, *_ = orc.scale_pwm(orc.compute_log_odds(m["probs"], m["bg"]))
        assert (sm == exp).all(), key


def test_pseudo_bg(golden_motifs):
    cases, flat = golden_motifs
    bg_nt = [("A", 0.2951), ("C", 0.2047), ("T", 0.2955), ("G", 0.2048)]  # file order
    unif = [(n, 0.25) for n in "ACGT"]
    assert np.array_equal(orc.pseudo_bg(unif, False), flat["ctcf_meme_unif#0"]["bg"])
    assert np.array_equal(orc.pseudo_bg(bg_nt, False), flat["ctcf_meme_bgnt#0"]["bg"])
    assert np.array_equal(orc.pseudo_bg(bg_nt, True), flat["ctcf_meme_bgnt_norev#0"]["bg"])


def test_score_seq_edge_rows(golden_motifs, golden_json):
    _, flat = golden_motifs
    for key, rows in golden_json("score_seq.json").items():
        m = flat[key]
        rows_ok = [r for r in rows if not r.get("reference_assert")]
        km = kmers_from_strings([r["seq"] for r in rows_ok])
        sc, lo, pv = orc.score_kmers(km, m["score_matrix"], m["pmf"], m["min_val"],
                                     m["scale"], m["offset"], sum_mode=0)
        assert np.array_equal(lo, np.array([r["score"] for r in rows_ok])), key
        assert np.array_equal(pv, np.array([r["pvalue"] for r in rows_ok])), key
        # rows on which the reference trips its own `pvalue <= 1` assert: the
        # oracle reproduces the >1 quotient
        bad = [r for r in rows if r.get("reference_assert")]
        if bad:
            km = kmers_from_strings([r["seq"] for r in bad])
            _, _, pvb = orc.score_kmers(km, m["score_matrix"], m["pmf"], m["min_val"],
                                        m["scale"], m["offset"], sum_mode=0)
            assert (pvb > 1.0).all()
        # numba-style sequential sums agree to 1e-12 relative
        _, _, pv1 = orc.score_kmers(kmers_from_strings([r["seq"] for r in rows]),
                                    m["score_matrix"], m["pmf"], m["min_val"], m["scale"],
                                    m["offset"], sum_mode=1)
        _, _, pv0 = orc.score_kmers(kmers_from_strings([r["seq"] for r in rows]),
                                    m["score_matrix"], m["pmf"], m["min_val"], m["scale"],
                                    m["offset"], sum_mode=0)
        np.testing.assert_allclose(pv1, pv0, rtol=1e-12)


def test_bh(golden_json):
    for case in golden_json("bh.json"):
        q = orc.fdr_bh(case["p"])
        assert np.array_equal(q, np.array(case["q"]))


def _motif_dict(m):
    return dict(score_matrix=m["score_matrix"], pmf=m["pmf"], min_val=m["min_val"],
                scale=m["scale"], offset=m["offset"], width=m["width"],
                motif_id=m["motif_id"], motif_name=m["motif_name"])


def test_reference_scoring_fixture(golden_motifs):
    """tests/grafimo_run_test.py:119-140 (test_scoring), oracle side."""
    _, flat = golden_motifs
    res = orc.compute_results(_motif_dict(flat["ctcf_meme_unif#0"]), REF_DATA,
                              threshold=1.0, recomb=True)
    cols = [c for c in res if not c.startswith("_")]
    df = pd.DataFrame({c: res[c] for c in cols})
    tmp = os.path.join("/tmp", f"orc_scoring_{os.getpid()}.tsv")
    df.to_csv(tmp, sep="\t")
    got = pd.read_csv(tmp, sep="\t", index_col=0).sort_values(
        ["p-value", "start", "stop"], ascending=True).reset_index(drop=True)
    os.remove(tmp)
    exp = pd.read_csv(os.path.join(REF_DATA, "scoring_results.tsv"), sep="\t", index_col=0
                      ).sort_values(["p-value", "start", "stop"], ascending=True
                                    ).reset_index(drop=True)
    assert got.equals(exp)


@pytest.mark.parametrize("name", ["default_t1e-2", "qvalt_t0.6", "noqvalue_t5e-3", "norev_t1e-1",
                                  "recomb_t1", "norecomb_t1", "cores4_t5e-2"])
def test_compute_results_flag_settings(golden_motifs, golden_json, name):
    _, flat = golden_motifs
    case = golden_json("compute_results.json")[name]
    kw = case["kwargs"]
    res = orc.compute_results(
        _motif_dict(flat["ctcf_meme_unif#0"]), REF_DATA,
        threshold=kw.get("threshold", 1e-4), no_qvalue=kw.get("no_qvalue", False),
        qval_t=kw.get("qval_t", False), no_reverse=kw.get("no_reverse", False),
        recomb=kw.get("recomb", False))
    cols = case["df"]["columns"]
    exp = pd.DataFrame(case["df"]["rows"], columns=cols)
    got = pd.DataFrame({c: res[c] for c in cols})
    key = ["p-value", "start", "stop", "strand"]
    exp = exp.sort_values(key).reset_index(drop=True)
    got = got.sort_values(key).reset_index(drop=True)
    assert len(got) == len(exp)
    for c in cols:
        if exp[c].dtype.kind == "f":
            assert np.array_equal(got[c].to_numpy(dtype=float), exp[c].to_numpy(dtype=float)), c
        else:
            assert (got[c].astype(str) == exp[c].astype(str)).all(), c


def test_text_loop_of_the_cpu_baseline_matches_the_row_functions(golden_motifs):
    """orc_score_tsv_text (bench.py's CPU baseline incl. the text handling of score_sequences.py:273-321)
    == parse_tsv_rows + score_kmers on the reference's 704-row fixture, both p-value variants."""
    _, flat = golden_motifs
    g = flat["ctcf_meme_unif#0"]
    path = os.path.join(REF_DATA, "width_19", "scoring_test_input.tsv")
    text = open(path, "rb").read()
    cols = orc.parse_tsv_rows([path])
    km = kmers_from_strings(cols["seq"])
    sc, lo, pv = orc.score_kmers(km, g["score_matrix"], g["pmf"], g["min_val"], g["scale"], g["offset"])
    refs = [r == "ref" and abs(b - a) == 19 for r, a, b in zip(cols["ref"], cols["start"], cols["stop"])]
    want = float(sc.sum()) + float(lo.sum()) + float(pv.sum()) + float(sum(cols["start"])) + float(sum(cols["freq"])) \
        + float(sum(refs))
    for table in (False, True):
        tab = orc.p_table(g["pmf"]) if table else g["pmf"]
        n, chk = orc.score_tsv_text(text, 19, g["score_matrix"], tab, g["min_val"], g["scale"], g["offset"], table=table)
        assert n == 704 and abs(chk - want) <= 1e-6 * abs(want)
    n, _ = orc.score_tsv_text(text, 19, g["score_matrix"], g["pmf"], g["min_val"], g["scale"], g["offset"],
                              no_reverse=True)
    assert n == sum(1 for s in cols["strand"] if s == "+")
