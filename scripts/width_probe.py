"""Development aid: the score kernel by width and mode (scores only / + histogram / + selection), 1e8 rows of
synthetic JASPAR-style motifs (the bench's config-4 recipe), with the launch geometry the library picked."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif
widths = [int(a) for a in sys.argv[1:]] or [30]
n = 100_000_000
dev = torch.device("cuda:0")
for W in widths:
    rng = np.random.default_rng(20240139 + 4)
    m = bench.synthetic_motif(W, rng, np.full(4, 0.25))
    dm = DeviceMotif(m["sm"], m["bg"], m["min_val"], m["scale"], m["offset"])
    d = synth.make_device_kmers(n, W, m["probs"], 9, dev)
    sc = torch.empty(n, dtype=torch.int32, device=dev)
    hist = torch.zeros(dm.L, dtype=torch.int64, device=dev)
    hits = torch.zeros(n // 32 + 1, dtype=torch.int64, device=dev)
    cut = dm.pvalue_cutoff(1e-4)
    lo, hi = dm.score_range() if hasattr(dm, "score_range") else (0, 0)
    out = []
    for mode in ("nohist", "hist", "select"):
        dm.profile_enable(12)
        for _ in range(12):
            if mode == "nohist":
                dm.score(d, sc)
            elif mode == "hist":
                dm.score(d, sc, hist=hist)
            else:
                dm.score(d, sc, hist=hist, select_cutoff=cut, hit_rows=hits[1:], hit_count=hits[:1], reset_hits=True)
        torch.cuda.synchronize()
        ms = np.sort(dm.profile_read()[1:])
        out.append(f"{mode} {np.median(ms)*1e3:7.1f} us ({n*(W+4)/np.median(ms)/1e9:.2f} TB/s)")
    print(f"W={W:2d} range [{lo},{hi}] " + "  ".join(out), flush=True)
    dm.close(); del d, sc
