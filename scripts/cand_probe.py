"""Development aid: candidate and hit counts of the --qvalueT path at config-4 shape."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.scan import KmerScanner
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda:0")
rng = np.random.default_rng(20240139 + 4)
m = bench.synthetic_motif(30, rng, np.full(4, 0.25))
dm = DeviceMotif(m["sm"], m["bg"], m["min_val"], m["scale"], m["offset"])
d = synth.make_device_kmers(n, 30, m["probs"], synth.seed_for(4, 0), dev)
sc = KmerScanner(dm, n, hit_capacity=max(4096, n // 64), device=dev)
slot = sc.enqueue(d, 1e-4, on_qvalue=True)
res = sc.collect(slot)
print("capacity", slot.hit_capacity, "candidates", int(slot.cand[0].item()), "hits", len(res["rows"]), "p-cutoff", dm.pvalue_cutoff(1e-4),
      "q-cutoff", int(slot.cutoff.item()))
