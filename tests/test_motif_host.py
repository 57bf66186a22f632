"""Host-side motif pipeline (parsers, background, pseudocounts, log-odds, scaling) vs vectors
captured from the reference.  CPU only: the p-value DP (GPU) is skipped via pvalue_matrix=False."""
import contextlib
import io
import os

import numpy as np
import pytest

from conftest import GOLDEN, REF_DATA
from grafimo_amd import motif_ops
from grafimo_amd.motif import Motif
from grafimo_amd.utils import UNIF


def _build(case):
    f = os.path.join(GOLDEN, case["file"])
    bg = case["bg_file"] if case["bg_file"] == UNIF else os.path.join(GOLDEN, case["bg_file"])
    ps, nr = float(case["pseudocount"]), case["no_reverse"]
    with contextlib.redirect_stdout(io.StringIO()):
        if case["format"] == "meme":
            return motif_ops.build_motif_meme(f, bg, ps, nr, 1, False, True, pvalue_matrix=False)
        fn = {"jaspar": motif_ops.build_motif_jaspar, "transfac": motif_ops.build_motif_transfac,
              "pfm": motif_ops.build_motif_pfm}[case["format"]]
        return [fn(f, bg, ps, nr, False, True, pvalue_matrix=False)]


def test_all_golden_cases_bit_exact(golden_motifs):
    cases, _ = golden_motifs
    for name, case in cases.items():
        ms = _build(case)
        assert len(ms) == len(case["motifs"]), name
        for m, g in zip(ms, case["motifs"]):
            assert isinstance(m, Motif)
            assert (m.motif_id, m.motif_name, m.width) == (g["motif_id"], g["motif_name"], g["width"])
            idx = [m.nucsmap[n] for n in "ACGT"]
            assert np.array_equal(np.asarray(m.count_matrix)[idx], np.array(g["probs"])), name
            assert [float(m.bg[n]) for n in "ACGT"] == g["bg"], name
            assert (m.dense_score_matrix() == np.array(g["score_matrix"])).all(), name
            assert (m.min_val, m.max_val, m.scale, float(m.offset)) == \
                (g["min_val"], g["max_val"], g["scale"], g["offset"]), name
            assert isinstance(m.scale, int) and isinstance(m.offset, np.double)
            assert m.is_scaled


def test_reference_known_answer_score_matrices():
    """tests/grafimo_run_test.py:68-116 restated against this package."""
    meme = np.loadtxt(os.path.join(REF_DATA, "motif_processing_test_meme.txt")).astype(int)
    jasp = np.loadtxt(os.path.join(REF_DATA, "motif_processing_test_jaspar.txt")).astype(int)
    with contextlib.redirect_stdout(io.StringIO()):
        m = motif_ops.build_motif_meme(os.path.join(REF_DATA, "MA0139.1.meme"), UNIF, 0.1, False,
                                       os.cpu_count(), False, True, pvalue_matrix=False)[0]
    assert (m.score_matrix == meme).all()
    for fn, f in [(motif_ops.build_motif_jaspar, "MA0139.1.jaspar"),
                  (motif_ops.build_motif_transfac, "MA0139.1.transfac"),
                  (motif_ops.build_motif_pfm, "MA0139.1.pfm")]:
        mm = fn(os.path.join(REF_DATA, f), UNIF, 0.1, False, False, True, pvalue_matrix=False)
        assert (mm.score_matrix == jasp).all(), f


def test_format_sniffers():
    p = lambda n: os.path.join(REF_DATA, n)  # noqa: E731
    assert motif_ops.is_meme(p("MA0139.1.meme"), True)
    assert motif_ops.is_jaspar(p("MA0139.1.jaspar"), True)
    assert not motif_ops.is_jaspar(p("MA0139.1.meme"), True)
    assert motif_ops.is_transfac(p("MA0139.1.transfac"), True)
    assert not motif_ops.is_transfac(p("MA0139.1.pfm"), True)
    assert motif_ops.is_pfm(p("MA0139.1.pfm"), True)
    assert not motif_ops.is_meme(p("MA0139.1.pfm"), True)


def test_get_motif_pwm_dispatch():
    class WF:
        bgfile, pseudo, noreverse, verbose = UNIF, 0.1, False, False
    for f in ["MA0139.1.meme", "MA0139.1.jaspar", "MA0139.1.transfac", "MA0139.1.pfm"]:
        with contextlib.redirect_stdout(io.StringIO()):
            ms = motif_ops.get_motif_pwm(os.path.join(REF_DATA, f), WF(), 1, True, pvalue_matrix=False)
        assert isinstance(ms, list) and ms[0].width == 19


def test_error_behaviour_mirrors_exception_handler(tmp_path):
    # debug=True raises the typed exception with the "\n\n" prefix (utils.py:63-78)
    with pytest.raises(FileNotFoundError) as e:
        motif_ops.build_motif_jaspar(str(tmp_path / "missing.jaspar"), UNIF, 0.1, False, False, True)
    assert str(e.value).startswith("\n\nUnable to locate")
    # debug=False prints "ERROR: ..." and exits with status 1
    with pytest.raises(SystemExit) as e:
        motif_ops.build_motif_jaspar(str(tmp_path / "missing.jaspar"), UNIF, 0.1, False, False, False)
    assert e.value.code == 1
    with pytest.raises(ValueError):
        motif_ops.build_motif_pfm(os.path.join(REF_DATA, "MA0139.1.pfm"), UNIF, -1.0, False, False, True)


def test_motif_setters_type_checks():
    m = Motif(np.ones((4, 3)), 3, ["A", "C", "G", "T"], "id", "name", {n: i for i, n in enumerate("ACGT")})
    with pytest.raises(TypeError):
        m.set_scale(2.0)
    with pytest.raises(TypeError):
        m.set_offset(1.0)          # must be numpy.double (motif.py:252-256)
    m.set_offset(np.double(-3.0))
    m.set_scale(7)
    with pytest.raises(ValueError):
        m.set_scale(0)
    with pytest.raises(AttributeError):
        _ = m.score_matrix
    m.set_is_scaled()
    with pytest.raises(AssertionError):
        m.set_is_scaled()
