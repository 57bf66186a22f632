"""Where the time of compute_results_from_graph goes (extraction -> scoring -> table), piece by piece."""
import contextlib, io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.extract_regions import DeviceGraph, compute_results_from_graph
from grafimo_amd.scan import KmerScanner
from grafimo_amd.workflow import Findmotif
ctcf = bench.load_ctcf()
dev = torch.device("cuda", 0)
idx, regions = synth.make_graph_index(10_000, 19)
g = DeviceGraph(idx, dev)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts)), r
ms, rows = t(lambda: g.extract(regions, 19)); print(f"extract            {ms:7.3f} ms  ({len(rows)} rows)")
ms, dm = t(lambda: DeviceMotif.from_motif(ctcf)); print(f"DeviceMotif        {ms:7.3f} ms")
n = len(rows)
ms, sc = t(lambda: KmerScanner(dm, n, hit_capacity=max(4096, n // 16), device=dev, side_stream=False, n_slots=1)); print(f"KmerScanner(1 slot){ms:7.3f} ms")
ms, sc3 = t(lambda: KmerScanner(dm, n, device=dev, side_stream=False)); print(f"KmerScanner(3 slots, full hit list){ms:7.3f} ms")
ms, res = t(lambda: sc.collect(sc.enqueue(rows.kmers, 1e-4))); print(f"enqueue + collect  {ms:7.3f} ms  ({len(res['rows'])} hits)")
wf = Findmotif(cores=1, threshold=1e-4)
with contextlib.redirect_stdout(io.StringIO()):
    ms, df = t(lambda: compute_results_from_graph(ctcf, g, regions, False, wf))
print(f"compute_results_from_graph {ms:7.3f} ms  ({len(df)} table rows)")
