import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from grafimo_amd import _native as nv, synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.scan import KmerScanner
ctcf = bench.load_ctcf()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
probs = np.asarray(ctcf.count_matrix, dtype=np.float64)
n = 20_000_000
first = synth.make_device_kmers(n, 19, probs, 7, dev)
bufs = [first, torch.roll(first, shifts=n // 3, dims=0).contiguous()]
dm = DeviceMotif.from_motif(ctcf)
sc = KmerScanner(dm, n, hit_capacity=n // 64, device=dev, side_stream=True, n_slots=3)
steps, ks = [], []
for i in range(40): sc.enqueue(bufs[i % 2], 1e-4)
torch.cuda.synchronize()
for rep in range(8):
    dm.profile_enable(128, every=4)
    t0 = time.perf_counter()
    for i in range(400): sc.enqueue(bufs[i % 2], 1e-4)
    torch.cuda.synchronize()
    steps.append(1e6 * (time.perf_counter() - t0) / 400)
    ks.append(1e3 * float(np.mean(dm.profile_read()))); dm.profile_enable(0)
print(f"LAB_POST={os.environ.get('GRAFIMO_LAB_POST','0')}: step median {np.median(steps):7.2f} us (min {min(steps):7.2f})  "
      f"kernel median {np.median(ks):7.2f} us (min {min(ks):.2f})", flush=True)
