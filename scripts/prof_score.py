"""Tiny driver for rocprofv3: runs the score kernel of BASELINE config 2 a few times."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.motif_ops import build_motif_meme_host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
regions = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
mode = sys.argv[3] if len(sys.argv) > 3 else "select"
m = build_motif_meme_host(os.path.join(ROOT, "tests/golden/ref_data/MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
b = synth.make_batch(regions, 2000, m.width, np.asarray(m.count_matrix), synth.seed_for(2))
dm = DeviceMotif(m.dense_score_matrix(), m.dense_bg(), m.min_val, m.scale, m.offset)
d = torch.from_numpy(b.kmers).cuda()
n = len(b)
sc = torch.empty(n, dtype=torch.int32, device="cuda")
hist = torch.zeros(dm.L, dtype=torch.int64, device="cuda")
hits = torch.zeros(n // 32 + 1, dtype=torch.int64, device="cuda")
cut = dm.pvalue_cutoff(1e-4)
if mode == "selnone":
    cut = 2**31 - 2
    mode = "select"
if mode.startswith("selp"):
    cut = dm.pvalue_cutoff(float(mode[4:]))
    mode = "select"
dm.profile_enable(reps)
for _ in range(reps):
    hits[:1].zero_()
    if mode == "select":
        dm.score(d, sc, hist=hist, select_cutoff=cut, hit_rows=hits[1:], hit_count=hits[:1])
    elif mode == "nohist":
        dm.score(d, sc)
    elif mode == "selonly":
        dm.score(d, sc, select_cutoff=cut, hit_rows=hits[1:], hit_count=hits[:1])
    else:
        dm.score(d, sc, hist=hist)
if os.environ.get("PROF_Q"):
    q = torch.empty(dm.L, dtype=torch.float64, device="cuda")
    cutd = torch.zeros(1, dtype=torch.int32, device="cuda")
    nr = torch.zeros(1, dtype=torch.int64, device="cuda")
    for _ in range(reps):
        dm.qvalue_table(hist, 1e-4, False, q, cutd, nr)
torch.cuda.synchronize()
ms = dm.profile_read()
ms = np.sort(ms[1:]) if len(ms) > 2 else ms
print(f"{os.environ.get('GRAFIMO_HIP_LIB', 'default').split('/')[-1]:28s} {mode:8s} kernel us: min {ms.min()*1e3:6.1f} "
      f"median {np.median(ms)*1e3:6.1f} max {ms.max()*1e3:6.1f}   GB/s alg at median: "
      f"{n * (m.width + 4) / (np.median(ms) * 1e-3) / 1e9:7.1f}")
