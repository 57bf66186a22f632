"""gfm_graph_score on the bench's synthetic chromosome, twenty calls: run under `rocprofv3 --kernel-trace --stats` for the
per-kernel times of the fused path (profiles/r04_extract_kernel_stats.csv)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.extract_regions import DeviceGraph
ctcf = bench.load_ctcf()
dev = torch.device("cuda", 0)
n_regions = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
idx, regions = synth.make_graph_index(n_regions, 19)
g = DeviceGraph(idx, dev)
reg = np.asarray(regions, dtype=np.int64)
starts, stops = np.ascontiguousarray(reg[:, 0]), np.ascontiguousarray(reg[:, 1])
dm = DeviceMotif.lease(ctcf)
hist = torch.zeros(dm.L, dtype=torch.int64, device=dev)
qtable = torch.empty(dm.L, dtype=torch.float64, device=dev)
cut = dm.pvalue_cutoff(1e-4)
for _ in range(4):          # a plan's first calls, one at a time: listing; the item count comes back; the walk cache is filled
    g.score(dm, starts, stops, cut, hist=hist)
    torch.cuda.synchronize()
for _ in range(20):
    g.score(dm, starts, stops, cut, hist=hist)
    g.annotate(qtable=None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    g.score(dm, starts, stops, cut, hist=hist)
e1.record()
torch.cuda.synchronize()
print("fused_ms %.4f" % (e0.elapsed_time(e1) / 20), g.fused_results()[:3], "hist sum", int(hist.sum().item()) // 44)
