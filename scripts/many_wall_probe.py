"""compute_results_from_graph_many (BASELINE configs[4] through the graph) taken apart with wall clocks around its steps -- no
profiler (cProfile books tens of ms of a profiled call to trivial pandas functions that a table built alone does not take)."""
import contextlib, io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pandas as pd
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


xr._split_lines = timed("split_lines (k-mers, labels)", xr._split_lines)
xr.RegionLabels.take = timed("labels.take", xr.RegionLabels.take)
xr._frame_of_columns = timed("frame_of_columns (incl. split_lines)", xr._frame_of_columns)
xr._FusedPass.enqueue = timed("enqueue", xr._FusedPass.enqueue)
xr._FusedPass.fetch = timed("fetch", xr._FusedPass.fetch)
xr._FusedPass.columns_start = timed("columns_start (native jobs handed to the library's threads)", xr._FusedPass.columns_start)
xr._FusedPass.columns_wait = timed("columns_wait", xr._FusedPass.columns_wait)
xr._FusedPass.frames = timed("frames (labels.take + frame_of_columns + prints)", xr._FusedPass.frames)
xr._FusedPass.__init__ = timed("pass init (motif handles, cutoffs)", xr._FusedPass.__init__)
xr._FusedPass.close = timed("pass close", xr._FusedPass.close)
_DF = pd.DataFrame.__init__
pd.DataFrame.__init__ = timed("DataFrame.__init__", _DF)

with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(4):
        xr.compute_results_from_graph_many(motifs, g, reg, False, wf)
    reps = 8
    acc.clear()
    t = time.perf_counter()
    for _ in range(reps):
        tabs = xr.compute_results_from_graph_many(motifs, g, reg, False, wf)
    total = time.perf_counter() - t
print(f"compute_results_from_graph_many, 50 PWMs x 50 000 regions, {sum(len(t_) for t_ in tabs)} hit rows: {1e3 * total / reps:.1f} ms per call")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:48s} {1e3 * v / reps:7.2f} ms per call")
