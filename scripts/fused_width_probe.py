"""Development aid (GPU box): gfm_graph_score on the bench's synthetic chromosome against the motif width (synthetic
JASPAR-style PWMs: wide motifs have wide score ranges, i.e. big LDS histogram windows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.extract_regions import DeviceGraph

for W in (8, 12, 19, 24, 30, 40, 64):
    idx, regions = synth.make_graph_index(10_000, W)
    g = DeviceGraph(idx)
    rec = synth.synthetic_motif(W, np.random.default_rng(100 + W), np.array([0.3, 0.2, 0.2, 0.3]))
    dm = DeviceMotif(rec["sm"], rec["bg"], rec["min_val"], rec["scale"], rec["offset"])
    reg = np.asarray(regions, dtype=np.int64)
    s0, s1 = np.ascontiguousarray(reg[:, 0]), np.ascontiguousarray(reg[:, 1])
    hist = torch.zeros(dm.L, dtype=torch.int64, device="cuda")
    cut = dm.pvalue_cutoff(1e-4)
    for _ in range(3):
        g.score(dm, s0, s1, cut, hist=hist)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.score(dm, s0, s1, cut, hist=hist)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    rows = int(g.fused_results()[1])
    lo, hi = dm.score_range() if hasattr(dm, "score_range") else (0, 0)
    print(f"W={W:2d}: {rows:9d} rows, gfm_graph_score {ms:.3f} ms = {rows / ms / 1e6:6.1f} G rows/s; score range {hi - lo + 1} bins", flush=True)
    dm.close()
    g.close()
