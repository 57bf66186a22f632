"""Development aid: score-kernel time against the number of rows, for both cache policies of the score stores
(GRAFIMO_STORE_POLICY, read per call): fixed cost + marginal rate, and where write-through stops paying."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.motif_ops import build_motif_meme_host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = build_motif_meme_host(os.path.join(ROOT, "tests/golden/ref_data/MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
dm = DeviceMotif(m.dense_score_matrix(), m.dense_bg(), m.min_val, m.scale, m.offset)
nmax = 125_000_000
dev = torch.device("cuda:0")
ds = [synth.make_device_kmers(nmax, m.width, np.asarray(m.count_matrix), 7 + i, dev) for i in range(2)]
scs = [torch.empty(nmax, dtype=torch.int32, device="cuda") for _ in range(3)]
hist = torch.zeros(dm.L, dtype=torch.int64, device="cuda")
hits = torch.zeros(nmax // 32 + 1, dtype=torch.int64, device="cuda")
cut = dm.pvalue_cutoff(1e-4)
reps = 30
for n in [2_500_000, 5_000_000, 10_000_000, 20_000_000, 30_000_000, 40_000_000, 60_000_000, 80_000_000, 125_000_000]:
    row = []
    for pol in ("through", "stream", "through", "stream"):
        os.environ["GRAFIMO_STORE_POLICY"] = pol
        dm.profile_enable(reps)
        for i in range(reps):   # inputs and outputs rotate like the bench's
            dm.score(ds[i & 1][:n], scs[i % 3][:n], hist=hist, select_cutoff=cut, hit_rows=hits[1:], hit_count=hits[:1],
                     reset_hits=True)
        torch.cuda.synchronize()
        ms = np.sort(dm.profile_read()[1:])
        row.append(np.median(ms) * 1e3)
    print(f"n {n:10d} ({n*4/2**20:6.0f} MiB of scores)  through {row[0]:8.2f} {row[2]:8.2f} us   stream {row[1]:8.2f} {row[3]:8.2f} us"
          f"   TB/s {n*23e-6/min(row[0],row[2]):.2f} / {n*23e-6/min(row[1],row[3]):.2f}", flush=True)
