"""Fifty single compute_results_from_graph calls (BASELINE configs[4] motifs over 50 000 regions), wall clocks around the native
columns of each (gfm_graph_hit_columns on records the DMA engine has just written: cold in every cache)."""
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
acc = [0.0, 0]
_hc = xr._hit_columns


def timed(*a, **k):
    t = time.perf_counter()
    try:
        return _hc(*a, **k)
    finally:
        acc[0] += time.perf_counter() - t
        acc[1] += 1


xr._hit_columns = timed
ts = []
with contextlib.redirect_stdout(io.StringIO()):
    for rep in range(8):
        acc[0], acc[1] = 0.0, 0
        t = time.perf_counter()
        rows = 0
        for m in motifs:
            rows += len(xr.compute_results_from_graph(m, g, reg, False, wf))
        ts.append((1e3 * (time.perf_counter() - t), 1e3 * acc[0], acc[1]))
v = ts[2:]
print(f"fifty single calls, {rows} hit rows: median {np.median([a for a, _, _ in v]):.1f} ms, of which gfm_graph_hit_columns "
      f"{np.median([b for _, b, _ in v]):.1f} ms in {v[0][2]} calls   {[round(a, 1) for a, _, _ in v]}")
