"""Fifty single compute_results_from_graph calls (BASELINE configs[4] motifs over 50 000 regions) -- what GRAFIMO's own
unchanged loop makes, one call per motif (grafimo.py:177-183) -- taken apart with wall clocks around the steps of a call."""
import contextlib, io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grafimo_amd import synth
from grafimo_amd import extract_regions as xr
from grafimo_amd.workflow import Findmotif

dev = torch.device("cuda", 0)
motifs = [synth.motif_object(m, f"M{i}") for i, m in enumerate(synth.config_motifs(5))]
idx, regions = synth.make_graph_index(50_000, max(m.width for m in motifs))
g = xr.DeviceGraph(idx, dev)
reg = np.asarray(regions, dtype=np.int64)
wf = Findmotif(threshold=1e-4)
acc = {}


def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
    return w


if "--plain" not in sys.argv:
    xr._hit_columns = timed("gfm_graph_hit_columns (inline)", xr._hit_columns)
    xr._split_lines = timed("split_lines (k-mers; labels once)", xr._split_lines)
    xr.RegionLabels.take = timed("labels.take", xr.RegionLabels.take)
    xr._frame_from_final_columns = timed("DataFrame from final columns", xr._frame_from_final_columns)
    xr._frame_of_columns = timed("frame_of_columns (incl. the two above)", xr._frame_of_columns)
    xr._FusedPass.enqueue = timed("enqueue", xr._FusedPass.enqueue)
    xr._FusedPass.fetch = timed("fetch (wait for the device + copies)", xr._FusedPass.fetch)
    xr._FusedPass.tables = timed("tables (columns + labels + frame + prints)", xr._FusedPass.tables)
    xr._FusedPass.__init__ = timed("pass init", xr._FusedPass.__init__)
    xr._prepare_entries = timed("prepare_entries (per call: graphs, regions, labels)", xr._prepare_entries)
ts = []
with contextlib.redirect_stdout(io.StringIO()):
    for rep in range(8):
        if rep == 2:
            acc.clear()
        t = time.perf_counter()
        rows = 0
        for m in motifs:
            rows += len(xr.compute_results_from_graph(m, g, reg, False, wf))
        ts.append(1e3 * (time.perf_counter() - t))
v = ts[2:]
print(f"fifty single calls, {rows} hit rows: median {np.median(v):.1f} ms   {[round(a, 1) for a in v]}")
for k, x in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:56s} {1e3 * x / len(v):7.2f} ms per fifty calls")
