import sys, time, tempfile, shutil
sys.path.insert(0, "/root/repo")
import torch
from grafimo_amd import synth
from grafimo_amd.extract_regions import DeviceGraph, write_region_tsvs
idx, regions = synth.make_graph_index(2000, 19)
g = DeviceGraph(idx, torch.device("cuda", 0))
regs = regions[:300]
rows = g.extract(regs, 19)
tmp = tempfile.mkdtemp()
t = time.perf_counter()
write_region_tsvs(idx, rows, tmp)
dt = time.perf_counter() - t
print(f"write_region_tsvs: {len(rows)} rows, {len(regs)} regions in {dt:.2f} s = {dt / len(rows) * 1e6:.1f} us per row")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); write_region_tsvs(idx, rows, tmp); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
shutil.rmtree(tmp)
