"""Report writers vs files written by the reference's own res_writer on the same tables."""
import os

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN
from grafimo_amd.motif import Motif
from grafimo_amd.res_writer import DEFAULT_OUTDIR, print_results, write_results, writeGFF3

REPORT = os.path.join(GOLDEN, "report")


def _table(golden_json, name):
    case = golden_json("compute_results.json")[name]
    return pd.DataFrame(case["df"]["rows"], columns=case["df"]["columns"]), case["kwargs"]


def _motif():
    return Motif(np.ones((4, 19)), 19, ["A", "C", "G", "T"], "MA0139.1", "CTCF",
                 {n: i for i, n in enumerate("ACGT")})


class Args:
    def __init__(self, outdir, noqvalue=False, top_graphs=0, verbose=False):
        self.outdir, self.noqvalue, self.top_graphs, self.verbose = outdir, noqvalue, top_graphs, verbose


@pytest.mark.parametrize("name", ["default_t1e-2", "noqvalue_t5e-3", "qvalt_t0.6"])
def test_gff3_and_tsv_byte_identical(golden_json, tmp_path, name):
    df, kw = _table(golden_json, name)
    noq = bool(kw.get("no_qvalue", False))
    writeGFF3(str(tmp_path / "out"), df, noq, True)
    assert open(tmp_path / "out.gff").read() == open(os.path.join(REPORT, name + ".gff")).read()
    write_results(df, _motif(), 1, Args(str(tmp_path / "res"), noqvalue=noq), True)
    assert open(tmp_path / "res" / "grafimo_out.tsv").read() == open(os.path.join(REPORT, name + ".tsv")).read()
    assert open(tmp_path / "res" / "grafimo_out.gff").read() == open(os.path.join(REPORT, name + ".gff")).read()
    html = open(tmp_path / "res" / "grafimo_out.html").read()
    assert html.startswith("<table") and "matched_sequence" in html


def test_naming_rules_and_errors(golden_json, tmp_path, capsys, monkeypatch):
    df, _ = _table(golden_json, "default_t1e-2")
    # several motifs into a user-given directory: per-motif prefix (res_writer.py:128-132)
    write_results(df, _motif(), 3, Args(str(tmp_path / "multi")), True)
    assert sorted(os.listdir(tmp_path / "multi")) == ["grafimo_out_MA0139.1.gff", "grafimo_out_MA0139.1.html",
                                                       "grafimo_out_MA0139.1.tsv"]
    # default directory name carries the PID and the motif id (res_writer.py:110-117)
    monkeypatch.chdir(tmp_path)
    write_results(df, _motif(), 1, Args(DEFAULT_OUTDIR), True)
    d = f"grafimo_out_{os.getpid()}_MA0139.1"
    assert os.path.isfile(os.path.join(tmp_path, d, "grafimo_out.tsv"))
    assert f"Writing results in {d}." in capsys.readouterr().out
    with pytest.raises(ValueError):
        write_results(df.iloc[:0], _motif(), 1, Args(str(tmp_path / "e")), True)
    with pytest.raises(ValueError):
        write_results(df, _motif(), 0, Args(str(tmp_path / "e")), True)
    with pytest.raises(TypeError):
        write_results("table", _motif(), 1, Args(str(tmp_path / "e")), True)
    with pytest.raises(NotImplementedError):
        write_results(df, _motif(), 1, Args(str(tmp_path / "e"), top_graphs=2), True)
    with pytest.raises(ValueError):                       # q-values requested but absent
        writeGFF3(str(tmp_path / "x"), df.drop(columns=["q-value"]), False, True)
    print_results(df, True)
    assert "matched_sequence" in capsys.readouterr().out
