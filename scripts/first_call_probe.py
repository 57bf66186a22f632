"""The first compute_results call of a process through scan_graph's manifest, taken apart: runtime start, index mapped, graph
uploaded (bitsets from the mapped file), count tables, plan, the first launches of every kernel."""
import contextlib, io, os, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
import torch
import bench
from grafimo_amd import synth
from grafimo_amd import extract_regions as xr
from grafimo_amd.score_sequences import compute_results
from grafimo_amd.workflow import Findmotif
t_import = time.perf_counter() - t0
acc = []


def timed(owner, name, label):
    fn = getattr(owner, name)

    def wrap(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc.append((label, 1e3 * (time.perf_counter() - t)))
    setattr(owner, name, wrap)


timed(xr.GraphIndex, "load", "GraphIndex.load (mapped)")
timed(xr.DeviceGraph, "__init__", "DeviceGraph.__init__ (upload + count tables)")
timed(xr.DeviceGraph, "score_many", "score_many (plan + first launches)")
timed(xr.DeviceGraph, "annotate", "annotate (enqueue)")
timed(xr.DeviceMotif, "lease", "DeviceMotif.lease (handle, DP)")
timed(xr._FusedPass, "fetch", "fetch (first synchronisation)")
timed(xr._FusedPass, "tables", "tables")
ctcf = bench.load_ctcf()
idx, regions = synth.make_graph_index(10_000, 19)
tmp = tempfile.mkdtemp(prefix="gfm_first_")
idx.save(os.path.join(tmp, "chr22"))
bed = os.path.join(tmp, "r.bed")
with open(bed, "w") as fh:
    for s, e in regions:
        fh.write(f"chr22\t{s}\t{e}\n")
torch.cuda.init()
t = time.perf_counter(); torch.zeros(1, device="cuda").cpu(); t_rt = 1e3 * (time.perf_counter() - t)
wf = Findmotif(cores=4, threshold=1e-4, graph_genome_dir=tmp, bedfile=bed, chroms_prefix="chr")
os.environ["GRAFIMO_SCAN_OUTPUT"] = "manifest"
with contextlib.redirect_stdout(io.StringIO()):
    t = time.perf_counter(); loc = xr.scan_graph({19}, wf, True); t_scan = 1e3 * (time.perf_counter() - t)
    t = time.perf_counter(); df = compute_results(ctcf, loc, True, wf); t_first = 1e3 * (time.perf_counter() - t)
    n_first = len(acc)
    t = time.perf_counter(); df = compute_results(ctcf, loc, True, wf); t_second = 1e3 * (time.perf_counter() - t)
print(f"imports {1e3 * t_import:.0f} ms, runtime start (first tensor) {t_rt:.1f} ms, scan_graph {t_scan:.1f} ms, first compute_results {t_first:.2f} ms, "
      f"second {t_second:.2f} ms, {len(df)} rows")
for label, ms in acc[:n_first]:
    print(f"  {label:48s} {ms:8.2f} ms")
shutil.rmtree(tmp, ignore_errors=True)
