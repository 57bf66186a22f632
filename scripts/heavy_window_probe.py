"""Development aid (GPU box): the fused path on windows of millions of walks (n neighbouring biallelic SNPs inside one
30-mer: 2^n walks per window, three such windows) -- time per gfm_graph_score call against n."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.extract_regions import DeviceGraph, GraphIndex

W = 30 if not (len(sys.argv) > 1 and sys.argv[1] == "del") else 60
rec = synth.synthetic_motif(W, np.random.default_rng(5), np.full(4, 0.25))
dm = DeviceMotif(rec["sm"], rec["bg"], rec["min_val"], rec["scale"], rec["offset"])
rng = np.random.default_rng(3)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
ref = acgt[rng.integers(0, 4, 600)]
with_del = len(sys.argv) > 1 and sys.argv[1] == "del"        # ... and a two-base deletion anchored behind the SNPs' middle
for n in (8, 12, 16, 20, 24)[:4 if with_del else 5]:
    pos = np.arange(300, 300 + 2 * n, 2, dtype=np.int32) if with_del else np.arange(300, 300 + n, dtype=np.int32)
    alt = np.zeros((len(pos), 3), np.uint8)
    alt[:, 0] = np.where(ref[pos] == ord("A"), ord("C"), ord("A"))
    n_alts, del_len = np.ones(len(pos), np.uint8), np.zeros(len(pos), np.int32)
    if with_del:
        at = int(pos[len(pos) // 2]) + 1
        pos, alt = np.insert(pos, len(pos) // 2 + 1, at), np.insert(alt, len(alt) // 2 + 1, 0, axis=0)
        n_alts, del_len = np.insert(n_alts, len(n_alts) // 2 + 1, 1), np.insert(del_len, len(del_len) // 2 + 1, 2)
    g = DeviceGraph(GraphIndex("c", ref, pos, n_alts, alt, None, 0, del_len=del_len))
    reg = np.array([(0, 200), (270, 360)], dtype=np.int64)
    s0, s1 = np.ascontiguousarray(reg[:, 0]), np.ascontiguousarray(reg[:, 1])
    hist = torch.zeros(dm.L, dtype=torch.int64, device="cuda")
    ts = []
    for it in range(3):
        hist.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.score(dm, s0, s1, dm.pvalue_cutoff(1e-9), hist=hist)
        count, n_rows, over, recs = g.fused_results()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{n:2d} SNPs in a row: {n_rows:12d} rows, {min(ts):9.3f} ms per call (first {ts[0]:.3f}) -> {n_rows / min(ts) / 1e6:8.1f} G rows/s; "
          f"histogram total {int(hist.sum())}", flush=True)
    g.close()
