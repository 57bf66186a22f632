"""The reference's own tutorial, as written (tutorials/findmotif_tutorial/findmotif_tutorial.sh:33-81, README.md:177-215):

    grafimo findmotif -d data/mygenome/ -m data/example.meme -b data/regions.bed  [-k data/bg_nt | -t 0.05 | --qvalueT -t 1e-4 |
                                                                                    --recomb | --chroms-find x]

on its own input files -- vg's x.xg / x.gbwt / y.xg / y.gbwt, never converted by hand: scan_graph finds them where the
reference would hand them to `vg find -x XG -H GBWT` (extract_regions.py:172-180), reads them (grafimo_amd/vg_files.py) and
compute_results scores the walks on the GPU.  Expected side: the CPU oracle end to end -- the rows of `vg find -K W -E -H`
from the walk enumerator over xy.fa + xy2.vcf.gz (the FASTA and VCF the tutorial built those graphs from), scored and filtered
by oracle.compute_results -- no HIP kernel and no byte of the XG / GBWT readers on that side."""
import contextlib
import io
import os
import shutil
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

from conftest import REF_DATA
from extract_helpers import assert_table_equals_oracle, motif_as_oracle_dict, variants_from_index
from grafimo_amd.extract_regions import scan_graph                    # <- grafimo.py:25
from grafimo_amd.score_sequences import compute_results              # <- grafimo.py:26

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
MYGENOME = os.path.join(REF_DATA, "mygenome")
BED = os.path.join(REF_DATA, "regions.bed")


def _oracle_tables(tmp, motif, chroms, **kw):
    """rows of both chromosomes into one directory (q-values are computed over all of them), then the oracle's compute_results"""
    from grafimo_amd.extract_regions import GraphIndex, read_bed_regions
    from oracle import extract_oracle as xo
    from oracle import oracle as orc
    W = int(motif.width)
    d = os.path.join(str(tmp), f"width_{W}")
    if os.path.isdir(d):
        shutil.rmtree(d)
    os.makedirs(d)
    regions = read_bed_regions(BED)
    for c in chroms:
        idx = GraphIndex.from_fasta_vcf(os.path.join(REF_DATA, "xy.fa"), os.path.join(REF_DATA, "xy2.vcf.gz"), c)
        v = variants_from_index(idx)
        ref = idx.ref
        for s, e in regions["chr" + c]:
            rows = xo.enumerate_region_variants(c, ref, v, int(s), int(e), W, with_counts=True)
            with open(os.path.join(d, f"{c}_{int(s)}-{int(e)}.tsv"), "w") as fh:
                for r in rows:
                    fh.write("\t".join(str(x) for x in r) + "\t1+,\n")
    res = orc.compute_results(motif_as_oracle_dict(motif), str(tmp), threshold=kw.get("threshold", 1e-4),
                              qval_t=kw.get("qval_t", False), no_qvalue=False, no_reverse=False, recomb=kw.get("recomb", False),
                              sum_mode=1)
    return pd.DataFrame({c: res[c] for c in res if not c.startswith("_")})


@pytest.fixture()
def mygenome(tmp_path, monkeypatch):
    g = tmp_path / "data" / "mygenome"
    shutil.copytree(MYGENOME, g)                        # (scan_graph saves x.gfmidx.npz beside x.xg: not into the repository)
    monkeypatch.setenv("GRAFIMO_INDEX_CACHE", str(tmp_path / "cache"))
    monkeypatch.delenv("GRAFIMO_SCAN_OUTPUT", raising=False)
    return str(g)


@pytest.mark.parametrize("flags", [dict(threshold=0.05), dict(threshold=0.05, bgfile="bg_nt"), dict(threshold=0.3, qval_t=True),
                                   dict(threshold=0.05, recomb=True), dict(threshold=0.05, chroms=["x"]), dict()])
def test_the_tutorial_on_vgs_own_files_equals_the_oracle(tmp_path, mygenome, flags):
    from grafimo_amd import extract_regions as xr
    from grafimo_amd.motif_ops import get_motif_pwm
    from grafimo_amd.workflow import Findmotif
    kw = dict(flags)
    if "bgfile" in kw:
        kw["bgfile"] = os.path.join(REF_DATA, kw["bgfile"])
    workflow = Findmotif(graph_genome_dir=mygenome, bedfile=BED, cores=2, **kw)
    motif_set = get_motif_pwm(os.path.join(REF_DATA, "example.meme"), workflow, 2, True, pvalue_matrix=False)
    assert len(motif_set) == 1
    # ---- grafimo.py:176-183
    with contextlib.redirect_stdout(io.StringIO()):
        sequences_loc = scan_graph({m.width for m in motif_set}, workflow, True)
        assert os.listdir(sequences_loc).count(xr.MANIFEST_NAME) == 1
        exp = _oracle_tables(tmp_path / "oracle", motif_set[0], flags.get("chroms", ["x", "y"]),
                             **{k: v for k, v in flags.items() if k != "chroms"})
        if len(exp) == 0:
            # the tutorial's default threshold on a 2 kb toy genome: nothing passes, and the reference stops there
            # (score_sequences.py:193-196 "No result retrieved. Unable to proceed.")
            with pytest.raises(Exception, match="No result retrieved"):
                compute_results(motif_set[0], sequences_loc, True, workflow)
            shutil.rmtree(sequences_loc)
            return
        res = compute_results(motif_set[0], sequences_loc, True, workflow)
    shutil.rmtree(sequences_loc)
    assert "x.gfmidx.npz" in os.listdir(mygenome) and ("y.gfmidx.npz" in os.listdir(mygenome)) == ("chroms" not in flags)
    assert len(res) > (5 if flags.get("qval_t") or "threshold" not in flags else 20), len(res)
    assert_table_equals_oracle(res, exp, str(flags))
    assert set(res["sequence_name"].str.split(":").str[0]) == set(flags.get("chroms", ["x", "y"]))
    if not flags.get("recomb"):
        assert res["haplotype_frequency"].isin([1, 2]).all()            # one sample: two haplotypes
        assert (res["reference"] == "non.ref").any() and (res["reference"] == "ref").any()


def test_the_tutorial_command_line(tmp_path, mygenome):
    """`python -m grafimo_amd -d data/mygenome/ -m data/example.meme -b data/regions.bed -t 0.05 -o OUT`: the reference's
    command with its program name exchanged; the TSV report holds the oracle's rows"""
    out = tmp_path / "grafimo_out_05"
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    cmd = [sys.executable, "-m", "grafimo_amd", "-d", mygenome + "/", "-m", os.path.join(REF_DATA, "example.meme"), "-b", BED,
           "-t", "0.05", "-o", str(out), "-j", "2"]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    report = pd.read_csv(out / "grafimo_out.tsv", sep="\t", index_col=0)
    from grafimo_amd.motif_ops import get_motif_pwm
    from grafimo_amd.workflow import Findmotif
    motif = get_motif_pwm(os.path.join(REF_DATA, "example.meme"), Findmotif(threshold=0.05), 1, True, pvalue_matrix=False)[0]
    exp = _oracle_tables(tmp_path / "oracle", motif, ["x", "y"], threshold=0.05)
    assert len(report) == len(exp) > 20
    key = ["sequence_name", "start", "stop", "strand", "matched_sequence"]
    a = report.sort_values(key).reset_index(drop=True)
    b = exp.sort_values(key).reset_index(drop=True)
    for c in key + ["haplotype_frequency", "reference"]:
        assert (a[c].astype(str) == b[c].astype(str)).all(), c
    for c in ("score", "p-value", "q-value"):
        np.testing.assert_allclose(a[c].to_numpy(float), b[c].to_numpy(float), rtol=1e-9)
    assert sorted(os.listdir(out)) == ["grafimo_out.gff", "grafimo_out.html", "grafimo_out.tsv"]


def test_the_command_line_with_a_motif_set(tmp_path, mygenome):
    """`python -m grafimo_amd -d DIR -b BED -m A.meme B.meme C.meme -t 0.05 -o OUT`: the motif set is scored in ONE
    compute_results_many call over the manifest its scan_graph left (the CLI held the whole list and still called
    compute_results per motif -- the shared enumeration was unreachable from any command line).  Every motif's report == the
    report of a run with that motif alone == the oracle's rows."""
    from grafimo_amd.motif_ops import get_motif_pwm
    from grafimo_amd.workflow import Findmotif
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    files = [os.path.join(REF_DATA, f) for f in ("example.meme", "MA0605.2.meme", "MA0035.4.meme")]      # widths 15, 12, 11
    out = tmp_path / "set"
    cmd = [sys.executable, "-m", "grafimo_amd", "-d", mygenome + "/", "-b", BED, "-t", "0.05", "-o", str(out), "-j", "2", "-m"] + files
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    assert done.stdout.count("Scanned sequences:") == 3
    motifs = [get_motif_pwm(f, Findmotif(threshold=0.05), 1, True, pvalue_matrix=False)[0] for f in files]
    names = sorted(os.listdir(out))
    assert names == sorted(f"grafimo_out_{m.motif_id}.{ext}" for m in motifs for ext in ("tsv", "html", "gff")), names
    key = ["sequence_name", "start", "stop", "strand", "matched_sequence"]
    for f, m in zip(files, motifs):
        report = pd.read_csv(out / f"grafimo_out_{m.motif_id}.tsv", sep="\t", index_col=0)
        alone = tmp_path / ("alone_" + m.motif_id)
        one = subprocess.run([sys.executable, "-m", "grafimo_amd", "-d", mygenome + "/", "-b", BED, "-t", "0.05", "-o", str(alone),
                              "-j", "2", "-m", f], env=env, capture_output=True, text=True, timeout=600)
        assert one.returncode == 0, one.stderr[-2000:]
        assert (alone / "grafimo_out.tsv").read_bytes() == (out / f"grafimo_out_{m.motif_id}.tsv").read_bytes(), m.motif_id
        exp = _oracle_tables(tmp_path / "oracle", m, ["x", "y"], threshold=0.05)
        assert len(report) == len(exp) > 10, (m.motif_id, len(report), len(exp))
        a, b = report.sort_values(key).reset_index(drop=True), exp.sort_values(key).reset_index(drop=True)
        for c in key + ["haplotype_frequency", "reference"]:
            assert (a[c].astype(str) == b[c].astype(str)).all(), (m.motif_id, c)
        for c in ("score", "p-value", "q-value"):
            np.testing.assert_allclose(a[c].to_numpy(float), b[c].to_numpy(float), rtol=1e-9)
