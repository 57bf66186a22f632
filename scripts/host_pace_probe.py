"""How long does the HOST take to enqueue one step of the resident pipeline (KmerScanner.enqueue: score kernel, post, q-table,
events)?  Measured with a batch so small (2 000 rows) that the device is never the limit: the loop's wall time per step is the
host's; beside it the same loop at the bench's size (2e7 rows), where the device is.  If the first number comes near the second
on some box, the GPU starves there whatever the slots."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.scan import KmerScanner
ctcf = bench.load_ctcf()
dev = torch.device("cuda", 0)
dm = DeviceMotif(ctcf.dense_score_matrix(), ctcf.dense_bg(), ctcf.min_val, ctcf.scale, float(ctcf.offset))
for n in (2_000, 20_000_000):
    d = synth.make_device_kmers(n, 19, np.asarray(ctcf.count_matrix, dtype=np.float64), 11, dev)
    sc = KmerScanner(dm, n, hit_capacity=max(4096, n // 64), device=dev, n_slots=3)
    for _ in range(50):
        sc.enqueue(d, 1e-4, want_qvalues=True)
    torch.cuda.synchronize()
    per = []
    for rep in range(7):
        t = time.perf_counter()
        for _ in range(400):
            sc.enqueue(d, 1e-4, want_qvalues=True)
        host = time.perf_counter() - t
        torch.cuda.synchronize()
        per.append(1e6 * host / 400)
    print(f"rows {n:>10d}: enqueue loop {np.median(per):6.1f} us per step on the host (min {min(per):.1f}, max {max(per):.1f})")
