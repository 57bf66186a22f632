"""One compute_results call through scan_graph's manifest (CTCF, 10 000 regions, p < 1e-4: 253 rows), taken apart with wall
clocks around its steps (no profiler: cProfile doubles the Python parts), median of 300 calls."""
import contextlib, io, os, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from grafimo_amd import synth
from grafimo_amd import extract_regions as xr
from grafimo_amd.score_sequences import compute_results
from grafimo_amd.workflow import Findmotif

acc = {}


def timed(owner, name, label=None):
    fn = getattr(owner, name)
    label = label or name

    def wrap(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc.setdefault(label, []).append(time.perf_counter() - t)
    setattr(owner, name, wrap)


for name in ("__init__", "enqueue", "fetch", "tables", "close"):
    timed(xr._FusedPass, name, "_FusedPass." + name)
for name in ("_hit_columns", "_frame_of_columns", "_fetch_fused", "_manifest_prep", "read_manifest"):
    timed(xr, name)
timed(xr.RegionLabels, "take", "RegionLabels.take")
timed(xr.DeviceGraph, "score_many", "DeviceGraph.score_many")
timed(xr.DeviceGraph, "annotate", "DeviceGraph.annotate")
timed(xr.DeviceGraph, "fused_zero", "DeviceGraph.fused_zero")
timed(xr.DeviceMotif, "qvalue_table", "DeviceMotif.qvalue_table")

ctcf = bench.load_ctcf()
if "--full-motif" in sys.argv:        # a Motif that carries its score distribution (pval_matrix), as GRAFIMO's own always does
    from grafimo_amd.motif_ops import build_motif_meme
    with contextlib.redirect_stdout(io.StringIO()):
        ctcf = build_motif_meme(os.path.join(bench.ROOT, "tests", "golden", "ref_data", "MA0139.1.meme"), "unfrm_dst", 0.1, False, 1, False, True)[0]
    assert getattr(ctcf, "pval_matrix", None) is not None
idx, regions = synth.make_graph_index(10_000, 19)
tmp = tempfile.mkdtemp(prefix="gfm_brk_")
idx.save(os.path.join(tmp, "chr22"))
bed = os.path.join(tmp, "regions.bed")
open(bed, "w").write("".join(f"chr22\t{s}\t{e}\n" for s, e in regions))
wf = Findmotif(cores=8, threshold=1e-4, graph_genome_dir=tmp, bedfile=bed, chroms_prefix="chr")
os.environ["GRAFIMO_SCAN_OUTPUT"] = "manifest"
with contextlib.redirect_stdout(io.StringIO()):
    loc = xr.scan_graph({19}, wf, False)
    for _ in range(20):
        compute_results(ctcf, loc, False, wf)
    acc.clear()
    ts = []
    for _ in range(300):
        t = time.perf_counter()
        compute_results(ctcf, loc, False, wf)
        ts.append(time.perf_counter() - t)
print(f"compute_results: median {1e6 * np.median(ts):.0f} us (min {1e6 * min(ts):.0f})")
for k, v in sorted(acc.items(), key=lambda kv: -np.median(kv[1])):
    print(f"  {k:28s} median {1e6 * np.median(v):7.1f} us  x{len(v) // 300}")
shutil.rmtree(loc); shutil.rmtree(tmp)
