"""The native TSV writer (csrc/graph_tsv_writer.cpp, what gfm_graph_write_tsvs runs per chunk) against the Python writer of
rounds 1-4 (tests/extract_helpers.write_region_tsvs_reference), byte for byte and WITHOUT a GPU: rows come from the oracle's
walk enumerator (oracle/extract_oracle.py), the native side is driven on host arrays by tests/native/tsv_writer_host.cpp,
built with g++ -fsanitize=address,undefined.  Rich graphs: SNPs, multi-allelic sites, insertions (several per anchor, longer
than a node), deletions (overlapping, nested), multi-base substitutions, complex alleles; regions that end inside the
chromosome (walks that must end inside the region), rows handed over in chunks that cut regions and windows apart."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from extract_helpers import make_consistent_graph_files, make_graph_files, write_region_tsvs_reference

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def writer_exe(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    csrc = os.path.join(ROOT, "grafimo_amd", "csrc")
    exe = str(tmp_path_factory.mktemp("tsvw") / "tsv_writer_host")
    base = ["g++", "-O1", "-g", "-std=c++17", f"-I{os.path.join(ROOT, 'include')}", f"-I{csrc}",
            os.path.join(ROOT, "tests", "native", "tsv_writer_host.cpp"), os.path.join(csrc, "graph_tsv_writer.cpp"),
            os.path.join(csrc, "gfm_workers.cpp"), "-lpthread", "-o", exe]
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
    build = subprocess.run(base[:1] + san + base[1:], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        build = subprocess.run(base, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    return exe


class _Rows:
    """an ExtractedKmers look-alike on CPU tensors (write_region_tsvs_reference reads .cpu().numpy() of each)"""

    def __init__(self, chrom, regions, W, cols):
        self.chrom, self.regions, self.width = chrom, list(regions), W
        for k, v in cols.items():
            setattr(self, k, torch.from_numpy(v))

    def __len__(self):
        return int(self.kmers.shape[0])

    def region_label(self, r):
        s, e = self.regions[r]
        return f"{self.chrom}:{s}-{e}"


def _oracle_columns(chrom, ref, v, regions, W):
    """the oracle's rows as the columns gfm_graph_emit writes (walk = rank of the row's walk inside its window: rows are
    window-major, forward row then '-' row)"""
    from oracle import extract_oracle as xo
    km, start, stop, strand, freq, is_ref, region, walk = [], [], [], [], [], [], [], []
    for r, (s, e) in enumerate(regions):
        last_p, q = None, 0
        for row in xo.enumerate_region_variants(chrom, ref, v, s, e, W, with_counts=True):
            a, b, sg = int(row[2].split(":")[1][:-1]), int(row[3].split(":")[1][:-1]), row[2][-1]
            p = a if sg == "+" else b
            if sg == "+":
                q = 0 if p != last_p else q + 1
                last_p = p
            km.append(np.frombuffer(row[1].encode(), dtype=np.uint8)); start.append(a); stop.append(b); strand.append(ord(sg))
            freq.append(row[4]); is_ref.append(1 if row[5] == "ref" else 0); region.append(r); walk.append(q)
    n = len(km)
    return dict(kmers=np.array(km, dtype=np.uint8).reshape(n, W), start=np.array(start, np.int64), stop=np.array(stop, np.int64),
                strand=np.array(strand, np.uint8), freq=np.array(freq, np.int64), is_ref=np.array(is_ref, np.uint8),
                region=np.array(region, np.int32), walk=np.array(walk, np.int32))


def _run_native(exe, idx, cols, regions, W, labels, chrom, out_dir, dump, chunk_rows, threads, node_paths=True):
    os.makedirs(dump, exist_ok=True)
    d = os.path.join(out_dir, f"width_{W}")
    os.makedirs(d, exist_ok=True)
    paths = [os.path.join(d, lb.replace(":", "_") + ".tsv") for lb in labels]
    n = len(cols["start"])
    with open(os.path.join(dump, "meta.txt"), "w") as fh:
        fh.write(f"{len(idx.ref)} {len(idx.pos)} {W} {len(regions)} {n} {chrom}\n")
    for name, arr, dt in (("pos.i32", idx.pos, np.int32), ("del_len.i32", idx.del_len, np.int32), ("ins_len.i32", idx.ins_len, np.int32),
                          ("n_alts.u8", idx.n_alts, np.uint8), ("kmers.u8", cols["kmers"], np.uint8), ("start.i64", cols["start"], np.int64),
                          ("stop.i64", cols["stop"], np.int64), ("freq.i64", cols["freq"], np.int64), ("region.i32", cols["region"], np.int32),
                          ("walk.i32", cols["walk"], np.int32), ("strand.u8", cols["strand"], np.uint8), ("is_ref.u8", cols["is_ref"], np.uint8),
                          ("region_stop.i64", np.array([e for _, e in regions]), np.int64)):
        np.ascontiguousarray(arr, dtype=dt).tofile(os.path.join(dump, name))
    open(os.path.join(dump, "labels.txt"), "w").write("".join(lb + "\n" for lb in labels))
    open(os.path.join(dump, "paths.txt"), "w").write("".join(p + "\n" for p in paths))
    run = subprocess.run([exe, dump, str(chunk_rows), str(threads), "1" if node_paths else "0"], capture_output=True, text=True,
                         timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    assert f"rows {n} " in run.stdout
    for p in paths:                     # (the regions without a row: extract_regions.finish_region_tsvs)
        if not os.path.exists(p):
            open(p, "w").close()
    return paths


@pytest.mark.parametrize("W,seed,rich", [(5, 3, True), (19, 4, True), (30, 5, True), (12, 6, False)])
def test_native_writer_equals_the_python_writer_on_rich_graphs(tmp_path, writer_exe, W, seed, rich):
    from grafimo_amd.extract_regions import GraphIndex
    from oracle import extract_oracle as xo
    fasta, vcf = make_graph_files(str(tmp_path), chrom="7", length=1800, n_sites=170, n_samples=12, seed=seed, rich=rich)
    idx = GraphIndex.from_fasta_vcf(fasta, vcf, "7")
    ref, v = xo.read_fasta(fasta)["7"], xo.read_vcf_variants(vcf, "7")
    regions = [(0, 240), (300, 300 + W - 1), (500, 900), (890, 1000), (1650, 1800)]
    cols = _oracle_columns("7", ref, v, regions, W)
    assert len(cols["start"]) > 2000
    rows = _Rows("7", regions, W, cols)
    labels = [f"chr7:{s}-{e}" for s, e in regions]
    exp = write_region_tsvs_reference(idx, rows, str(tmp_path / "py"), labels=labels, chrom="chr7")
    for chunk_rows, threads in ((1 << 30, 1), (997, 3), (64, 4)):          # one chunk; chunks that cut regions and windows
        got = _run_native(writer_exe, idx, cols, regions, W, labels, "chr7", str(tmp_path / f"nat{chunk_rows}"),
                          str(tmp_path / "dump"), chunk_rows, threads)
        for a, b in zip(exp, got):
            ta, tb = open(a, "rb").read(), open(b, "rb").read()
            assert ta == tb, (W, chunk_rows, os.path.basename(a), next((x, y) for x, y in zip(ta.split(b"\n"), tb.split(b"\n")) if x != y))
    # column 7 left empty on request: columns 1-6 unchanged
    got = _run_native(writer_exe, idx, cols, regions, W, labels, "chr7", str(tmp_path / "nopath"), str(tmp_path / "dump"), 500, 2,
                      node_paths=False)
    for a, b in zip(exp, got):
        la, lb = open(a).read().splitlines(), open(b).read().splitlines()
        assert [ln.rsplit("\t", 1)[0] for ln in la] == [ln.rsplit("\t", 1)[0] for ln in lb]
        assert all(ln.endswith("\t") for ln in lb)


@pytest.mark.parametrize("kinds,seed", [("sidmDO", 11), ("sidmDOcS", 12), ("ic", 13), ("sD", 14), ("sO", 15)])
def test_native_writer_on_nested_and_overlapping_deletions_and_complex_alleles(tmp_path, writer_exe, kinds, seed):
    import sys
    from grafimo_amd.extract_regions import GraphIndex
    from oracle import extract_oracle as xo
    fasta, vcf = make_consistent_graph_files(str(tmp_path), chrom="c", length=420, n_samples=8, seed=seed, kinds=kinds)
    stderr = sys.stderr
    with open(os.devnull, "w") as dn:
        sys.stderr = dn
        try:
            idx = GraphIndex.from_fasta_vcf(fasta, vcf, "c")
        finally:
            sys.stderr = stderr
    ref, v = xo.read_fasta(fasta)["c"], xo.read_vcf_variants(vcf, "c")
    for W, regions in ((11, [(0, 150), (140, 300), (380, 420)]), (24, [(100, 330)])):
        cols = _oracle_columns("c", ref, v, regions, W)
        if len(cols["start"]) > 150_000:
            continue
        rows = _Rows("c", regions, W, cols)
        labels = [f"c:{s}-{e}" for s, e in regions]
        exp = write_region_tsvs_reference(idx, rows, str(tmp_path / f"py{W}"), labels=labels, chrom="c")
        got = _run_native(writer_exe, idx, cols, regions, W, labels, "c", str(tmp_path / f"nat{W}"), str(tmp_path / "dump"), 4096, 4)
        for a, b in zip(exp, got):
            assert open(a, "rb").read() == open(b, "rb").read(), (kinds, W, os.path.basename(a))
