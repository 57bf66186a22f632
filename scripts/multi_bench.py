"""Batched same-width motifs (BASELINE config 5 shape): one shared k-mer read vs separate launches."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif, score_multi
W = int(sys.argv[1]) if len(sys.argv) > 1 else 19
regions = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
rng = np.random.default_rng(5)
motifs = []
for k in range(3):
    # JASPAR-like columns: Dirichlet(0.3) counts -> log-odds -> scaled ints through the library
    from grafimo_amd.device import compute_log_odds_dense, scale_pwm_dense
    probs = (rng.dirichlet([0.3] * 4, size=W).T * 1000 + 0.025) / 1000.1
    bg = rng.dirichlet(50 * synth.BG_NT)
    sm, mn, mx, sc, off = scale_pwm_dense(compute_log_odds_dense(probs, bg))
    motifs.append(DeviceMotif(sm, bg, mn, sc, off))
    print(f"motif {k}: window {motifs[-1].score_hi - motifs[-1].score_lo + 1} bins")
b = synth.make_batch(regions, 2000, W, probs, 11)
n = len(b)
d = torch.from_numpy(b.kmers).cuda()
scores = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in motifs]
hists = [torch.zeros(m.L, dtype=torch.int64, device="cuda") for m in motifs]
hits = [torch.zeros(n // 32 + 1, dtype=torch.int64, device="cuda") for _ in motifs]
cuts = [m.pvalue_cutoff(1e-4) for m in motifs]
def run_multi(M):
    score_multi(motifs[:M], d, scores[:M], hists=hists[:M], cutoffs=cuts[:M], hit_rows=[h[1:] for h in hits[:M]],
                hit_counts=[h[:1] for h in hits[:M]], reset_hits=True)
def run_single(M):
    for j in range(M):
        motifs[j].score(d, scores[j], hist=hists[j], select_cutoff=cuts[j], hit_rows=hits[j][1:], hit_count=hits[j][:1],
                        reset_hits=True)
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M in (1, 2, 3):
    a, s = t(lambda: run_multi(M)), t(lambda: run_single(M))
    print(f"W={W} M={M}: batched {a:7.1f} us ({n*M/a/1e3:6.1f} G pairs/s, {n*(W/M+4)*M/a/1e6:5.2f} TB/s alg)   "
          f"separate {s:7.1f} us ({n*M/s/1e3:6.1f} G pairs/s)   speed-up {s/a:.2f}x")
