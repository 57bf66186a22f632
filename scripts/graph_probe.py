"""Is the single-stream path capturable into a HIP graph (the header says: only enqueues)?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.motif_ops import build_motif_meme_host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = build_motif_meme_host(os.path.join(ROOT, "tests/golden/ref_data/MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
b = synth.make_batch(2000, 2000, 19, np.asarray(m.count_matrix), 3)
dm = DeviceMotif(m.dense_score_matrix(), m.dense_bg(), m.min_val, m.scale, m.offset)
n = len(b); d = torch.from_numpy(b.kmers).cuda()
sc = torch.empty(n, dtype=torch.int32, device="cuda"); hist = torch.zeros(dm.L, dtype=torch.int64, device="cuda")
hits = torch.zeros(n // 16 + 1, dtype=torch.int64, device="cuda")
q = torch.empty(dm.L, dtype=torch.float64, device="cuda"); cut = torch.zeros(1, dtype=torch.int32, device="cuda"); nr = torch.zeros(1, dtype=torch.int64, device="cuda")
c = dm.pvalue_cutoff(1e-4)
def step():
    dm.score(d, sc, hist=hist, select_cutoff=c, hit_rows=hits[1:], hit_count=hits[:1], reset_hits=True)
    dm.qvalue_table(hist, 1e-4, False, q, cut, nr, clear_hist=True)
step(); step(); torch.cuda.synchronize()
ref = (sc.clone(), int(hits[0]), q.clone(), int(nr))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step(); step()       # warm both workspace parities on this stream
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        step(); step()
sc.zero_(); hits.zero_(); q.zero_()
for _ in range(3): g.replay()
torch.cuda.synchronize()
ok = torch.equal(sc, ref[0]) and int(hits[0]) == ref[1] and torch.equal(q, ref[2]) and int(nr) == ref[3]
print("graph capture + replay:", "OK" if ok else "MISMATCH", ref[1], int(hits[0]))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"graph replay: {e0.elapsed_time(e1)/100*1e3:.1f} us per step ({n} rows)")
e0.record()
for _ in range(100): step()
e1.record(); torch.cuda.synchronize()
print(f"eager:        {e0.elapsed_time(e1)/100*1e3:.1f} us per step")
