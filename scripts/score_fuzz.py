"""Development aid (GPU box): the score kernels against the CPU oracle (oracle/grafimo_oracle.c) on random motifs, widths,
group sizes, row counts (ragged, tiny, around the grid's turn boundaries), N / lowercase rows, histogram and selection flags,
seed after seed for a fixed time.  TEST INFRASTRUCTURE (imports oracle/): not part of the product.
    python scripts/score_fuzz.py [seconds] [first_seed]"""
import os
import sys
import time

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import torch
from test_gpu_parity_sweep import band_matrix, check_against_oracle, random_kmers
from grafimo_amd.device import DeviceMotif, score_multi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
t0 = time.time()
cases = rows = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    mm = int(rng.choice([1, 1, 2, 3, 4, 5]))
    W = int(rng.integers(1, 65 if mm == 1 else 33))
    full_range = rng.random() < 0.3                     # full-range matrices: partial LDS windows + spill counters
    motifs, mats = [], []
    for k in range(mm):
        if full_range:
            sm = rng.integers(0, 1001, size=(4, W)).astype(np.int64)
            sm[rng.integers(0, 4), rng.integers(0, W)] = 0
        else:
            sm = band_matrix(rng, W, 4200 // (W * min(mm, 3)) + 40 // W)
        bg = rng.dirichlet([30, 20, 20, 30])
        mats.append((sm, bg))
        motifs.append(DeviceMotif(sm, bg, int(sm.min()), 40 + k, -9.0 - k))
    n = int(rng.choice([2, 1, 63, 255, 256, 257, 4095, 4096 * 256 - 1, 4096 * 256 + 300, int(rng.integers(1, 300_000)),
                        int(rng.integers(300_000, 2_500_000))]))
    km = random_kmers(rng, n, W, n_frac=float(rng.choice([0.0, 0.01, 0.3])), lower_frac=float(rng.choice([0.0, 0.05])))
    d_k = torch.from_numpy(km).to(dev) if n else torch.zeros((0, W), dtype=torch.uint8, device=dev)
    row_base = int(rng.choice([0, 11, 2 ** 33]))
    scores = [torch.full((n,), -7, dtype=torch.int32, device=dev) for _ in motifs]
    hists = [torch.zeros(m.L, dtype=torch.int64, device=dev) if rng.random() < 0.8 else None for m in motifs]
    cuts = [m.pvalue_cutoff(float(rng.choice([1e-4, 0.03, 0.5]))) if rng.random() < 0.8 else None for m in motifs]
    hits = [torch.zeros(n + 1, dtype=torch.int64, device=dev) for _ in motifs]
    try:
        score_multi(motifs, d_k, scores, hists=hists, cutoffs=cuts, row_base=row_base,
                    hit_rows=[h[1:] for h in hits], hit_counts=[h[:1] for h in hits], reset_hits=True)
        torch.cuda.synchronize()
    except Exception:
        print("FAILED case:", dict(seed=seed, W=W, mm=mm, n=n, full_range=full_range, L=[m.L for m in motifs],
                                   hists=[h is not None for h in hists], cuts=cuts), flush=True)
        raise
    if row_base < 2 ** 20 or True:
        check_against_oracle(dev, motifs, mats, km, scores, hists, hits, cuts, row_base, (seed, W, mm, n, full_range))
    for m in motifs:
        m.close()
    cases += 1
    rows += n * mm
    seed += 1
print(f"score_fuzz: {cases} launches, {rows} (k-mer, motif) pairs in {time.time() - t0:.0f} s: scores, histograms and hit "
      f"lists == oracle; next seed {seed}")
