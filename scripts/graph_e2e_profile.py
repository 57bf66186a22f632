import contextlib, io, os, sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from grafimo_amd import synth
from grafimo_amd.extract_regions import DeviceGraph, compute_results_from_graph
from grafimo_amd.workflow import Findmotif
ctcf = bench.load_ctcf()
dev = torch.device("cuda", 0)
idx, regions = synth.make_graph_index(10_000, 19)
g = DeviceGraph(idx, dev)
wf = Findmotif(cores=1, threshold=1e-4)
with contextlib.redirect_stdout(io.StringIO()):
    regions = np.asarray(regions, dtype=np.int64)
    for _ in range(3): compute_results_from_graph(ctcf, g, regions, False, wf)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): compute_results_from_graph(ctcf, g, regions, False, wf)
    pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
