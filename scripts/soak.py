"""Soak test of the pipelined scan (score kernel || post + q-table on the tail stream): every step's hit list,
q-table and row count must equal the first step's.  Catches rare races between the streams and in the slot /
workspace / hit-counter rings.   scripts/soak.py [steps] [n_slots]
The q-value threshold used (0.2) makes the p < t candidate list overflow its capacity: the device-side fallback
over the scores runs on every such step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.motif_ops import build_motif_meme_host
from grafimo_amd.scan import KmerScanner
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
m = build_motif_meme_host(os.path.join(ROOT, "tests/golden/ref_data/MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
dm = DeviceMotif(m.dense_score_matrix(), m.dense_bg(), m.min_val, m.scale, m.offset)
batch = synth.make_batch(2000, 2000, 19, np.asarray(m.count_matrix), synth.seed_for(2))
d = torch.from_numpy(batch.kmers).cuda()
n = len(batch)
n_slots = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sc = KmerScanner(dm, n, hit_capacity=n // 32, device=d.device, side_stream=True, n_slots=n_slots)
ref = None
bad = 0
t0 = time.time()
for it in range(steps):
    on_q = (it // 50) % 2 == 1                      # alternate p- and q-value thresholds in blocks of 50
    slot = sc.enqueue(d, 0.2 if on_q else 1e-4, on_qvalue=on_q)
    if it % 7 == 3 or it == steps - 1:              # inspect some steps (this drains the pipeline)
        r = sc.collect(slot)
        key = (on_q, r["rows"].tobytes(), r["scaled"].tobytes(), r["qtable"].tobytes(), r["n_scored"])
        if ref is None:
            ref = {}
        if on_q not in ref:
            ref[on_q] = key
        elif ref[on_q] != key:
            bad += 1
print(f"{steps} steps in {time.time() - t0:.1f} s, mismatching inspected steps: {bad}")
sys.exit(1 if bad else 0)
