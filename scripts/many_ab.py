"""compute_results_from_graph_many (BASELINE configs[4] through the graph), fourteen calls in one process.  (Round 6 used it to
compare the native columns one motif ahead on a helper thread against inline: 65.6 against 65.8 ms; the thread is gone and
xr._COLUMNS_AHEAD with it -- both passes below run the product.)"""
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
ts = []
with contextlib.redirect_stdout(io.StringIO()):
    for rep in range(14):
        t = time.perf_counter()
        tabs = xr.compute_results_from_graph_many(motifs, g, reg, False, wf)
        ts.append(1e3 * (time.perf_counter() - t))
v = ts[2:]
print(f"compute_results_from_graph_many, 50 PWMs x 50 000 regions, {sum(len(t_) for t_ in tabs)} hit rows: median {np.median(v):.1f} ms, "
      f"min {min(v):.1f}, max {max(v):.1f}  {[round(x, 1) for x in v]}")
