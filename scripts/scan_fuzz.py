"""Development aid (GPU box): the resident pipeline (scan.KmerScanner: score kernel -> post -> q-table -> selection, tail on a
side stream, three slots in flight) against the CPU oracle on random motifs, batch sizes and thresholds on p- and q-values,
seed after seed for a fixed time.  TEST INFRASTRUCTURE (imports oracle/): not part of the product.
    python scripts/scan_fuzz.py [seconds] [first_seed]"""
import os
import sys
import time

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import torch
from test_gpu_parity_sweep import band_matrix, random_kmers
from grafimo_amd.device import DeviceMotif
from grafimo_amd.scan import KmerScanner
from oracle import oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
t0 = time.time()
scanners = batches = rows = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    W = int(rng.integers(4, 41))
    sm = band_matrix(rng, W, 4200 // W) if rng.random() < 0.7 else rng.integers(0, 1001, size=(4, W)).astype(np.int64)
    sm[rng.integers(0, 4), rng.integers(0, W)] = 0
    bg = rng.dirichlet([30, 20, 20, 30])
    dm = DeviceMotif(sm, bg, int(sm.min()), 40, -9.0)
    _, pt = dm.tables()
    n_max = int(rng.choice([300, 5000, 70_000, 400_000, 1_500_000]))
    side = bool(rng.random() < 0.7)
    sc = KmerScanner(dm, n_max, device=dev, side_stream=side, n_slots=int(rng.choice([1, 2, 3])) if not side else 3)
    pending = []

    def flush():
        global batches, rows
        while pending:
            slot, km, thr, on_q, want_q, row_base = pending.pop(0)
            res = sc.collect(slot, want_qvalues=want_q)
            exp, p = orc.score_kmers_table(km, sm, pt, int(sm.min()))
            q = orc.fdr_bh(p) if want_q else None
            keep = np.nonzero((q if on_q else p) < thr)[0]
            tag = (seed, W, len(km), thr, on_q, want_q, side)
            assert np.array_equal(res["rows"], keep + row_base), tag
            assert np.array_equal(res["scaled"], exp[keep]), tag
            if want_q and len(keep):
                np.testing.assert_allclose(np.asarray(res["qtable"])[exp[keep]], q[keep], rtol=1e-12, err_msg=str(tag))
            batches += 1
            rows += len(km)

    for b in range(int(rng.integers(2, 7))):
        n = int(rng.integers(max(1, n_max // 3), n_max + 1))
        km = random_kmers(rng, n, W, n_frac=float(rng.choice([0.0, 0.01])))
        on_q = bool(rng.random() < 0.4)
        want_q = on_q or bool(rng.random() < 0.7)
        thr = float(rng.choice([1e-4, 1e-3, 0.02, 0.3] if not on_q else [0.05, 0.5, 0.9]))
        row_base = int(rng.choice([0, 1000, 2 ** 31]))
        slot = sc.enqueue(torch.from_numpy(km).to(dev), thr, on_qvalue=on_q, want_qvalues=want_q, row_base=row_base)
        pending.append((slot, km, thr, on_q, want_q, row_base))
        if len(pending) >= max(1, len(sc.slots) - 1) or b % 2:        # keep up to n_slots - 1 batches in flight
            flush()
    flush()
    dm.close()
    scanners += 1
    seed += 1
print(f"scan_fuzz: {scanners} scanners, {batches} batches, {rows} rows in {time.time() - t0:.0f} s: hit rows, scores and q-values "
      f"== oracle; next seed {seed}")
