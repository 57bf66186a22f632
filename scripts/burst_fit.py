"""Development aid: time of a synchronised burst of K pipelined steps against K (fixed cost + per-step time)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.motif_ops import build_motif_meme_host
from grafimo_amd.scan import KmerScanner
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = build_motif_meme_host(os.path.join(ROOT, "tests/golden/ref_data/MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
dm = DeviceMotif(m.dense_score_matrix(), m.dense_bg(), m.min_val, m.scale, m.offset)
n = 20_000_000
dev = torch.device("cuda:0")
ds = [synth.make_device_kmers(n, 19, np.asarray(m.count_matrix), 7 + i, dev) for i in range(2)]
sc = KmerScanner(dm, n, hit_capacity=n // 64, device=dev)
for i in range(10): sc.enqueue(ds[i & 1], 1e-4)
sc.finish(); torch.cuda.synchronize()
pts = []
for K in (1, 2, 3, 5, 10, 20, 40):
    ts = []
    for rep in range(9):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(K): sc.enqueue(ds[i & 1], 1e-4)
        sc.finish(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e6)
    pts.append((K, np.median(ts)))
    print(f"K={K:3d}: burst {np.median(ts):8.1f} us  ({np.median(ts)/K:7.1f} per step)", flush=True)
x = np.array([p[0] for p in pts], float); y = np.array([p[1] for p in pts])
b, a = np.polyfit(x, y, 1)
print(f"fit: fixed {a:.1f} us + {b:.2f} us per step")
