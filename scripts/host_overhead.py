import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.motif_ops import build_motif_meme_host
from grafimo_amd.scan import KmerScanner
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = build_motif_meme_host(os.path.join(ROOT, "tests/golden/ref_data/MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
b = synth.make_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 10000, 2000, 19, np.asarray(m.count_matrix), 1)
dm = DeviceMotif(m.dense_score_matrix(), m.dense_bg(), m.min_val, m.scale, m.offset)
d = torch.from_numpy(b.kmers).cuda(); n = len(b)
for slots in (2, 3, 4):
    sc = KmerScanner(dm, n, hit_capacity=n // 32, n_slots=slots)
    for _ in range(5): sc.enqueue(d, 1e-4)
    sc.finish(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): sc.enqueue(d, 1e-4)
    t1 = time.perf_counter()
    sc.finish(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"slots={slots} host enqueue {1e6*(t1-t0)/200:.1f} us/step, total {1e6*(t2-t0)/200:.1f} us/step")
