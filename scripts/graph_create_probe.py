import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from grafimo_amd import synth, extract_regions as xr
idx, regions = synth.make_graph_index(10_000, 19)
import tempfile
tmp = tempfile.mkdtemp()
idx.save(os.path.join(tmp, "chr22"))
torch.zeros(1, device="cuda").cpu()
ts = []
for rep in range(4):
    ix = xr.GraphIndex.load(os.path.join(tmp, "chr22" + xr.INDEX_SUFFIX)) if rep % 2 == 0 else idx     # mapped file / heap arrays
    t = time.perf_counter(); g = xr.DeviceGraph(ix); torch.cuda.synchronize(); ts.append((("mapped" if rep % 2 == 0 else "heap"), 1e3 * (time.perf_counter() - t))); g.close()
print("DeviceGraph.__init__ ms:", [(a, round(b, 2)) for a, b in ts], "bitset MB", idx.alt_bits.nbytes / 1e6)
