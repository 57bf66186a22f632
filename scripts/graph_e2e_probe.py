"""Where the time of compute_results_from_graph goes, piece by piece: the fused path (gfm_graph_score -> q-table ->
gfm_graph_annotate -> records back -> table) and, for comparison, the materialising one."""
import contextlib, io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.extract_regions import DeviceGraph, compute_results_from_graph
from grafimo_amd.workflow import Findmotif
ctcf = bench.load_ctcf()
dev = torch.device("cuda", 0)
idx, regions = synth.make_graph_index(10_000, 19)
g = DeviceGraph(idx, dev)
reg = np.asarray(regions, dtype=np.int64)
starts, stops = np.ascontiguousarray(reg[:, 0]), np.ascontiguousarray(reg[:, 1])
def t(f, reps=9):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts)), r
def ev(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
dm = DeviceMotif.lease(ctcf)
work = dm.fused_workspace(dev)
L = dm.L
hist, qtable, d_cut = work[:L], work[L:2 * L].view(torch.float64), work[2 * L:2 * L + 1].view(torch.int32)[:1]
cut = dm.pvalue_cutoff(1e-4)
print(f"gfm_graph_score (device, events)   {ev(lambda: g.score(dm, starts, stops, cut, hist=hist)):7.3f} ms")
print(f"gfm_graph_score no hist (device)   {ev(lambda: g.score(dm, starts, stops, cut, hist=None)):7.3f} ms")
ms, _ = t(lambda: g.score(dm, starts, stops, cut, hist=hist)); print(f"g.score wall (enqueue + sync)      {ms:7.3f} ms")
print(f"q-table (device)                   {ev(lambda: dm.qvalue_table(hist, 1e-4, False, qtable, d_cut, None)):7.3f} ms")
hist.zero_(); g.score(dm, starts, stops, cut, hist=hist)
print(f"annotate (device)                  {ev(lambda: g.annotate(qtable=qtable)):7.3f} ms")
ms, res = t(lambda: g.fused_results()); print(f"fused_results (2 D2H copies)       {ms:7.3f} ms  ({res[0]} hits, {res[1]} rows)")
ms, _ = t(lambda: hist.zero_()); print(f"hist.zero_() wall                  {ms:7.3f} ms")
wf = Findmotif(cores=1, threshold=1e-4)
with contextlib.redirect_stdout(io.StringIO()):
    ms, df = t(lambda: compute_results_from_graph(ctcf, g, reg, False, wf), reps=15)
    ms_l, _ = t(lambda: compute_results_from_graph(ctcf, g, regions, False, wf), reps=9)
    ms_m, dfm = t(lambda: compute_results_from_graph(ctcf, g, regions, False, wf, fused=False), reps=5)
print(f"compute_results_from_graph fused, regions as an array {ms:7.3f} ms  ({len(df)} table rows)")
print(f"compute_results_from_graph fused, regions as a list   {ms_l:7.3f} ms")
print(f"compute_results_from_graph materialising              {ms_m:7.3f} ms  ({len(dfm)} table rows)")
