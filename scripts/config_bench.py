"""Single-GPU measurements of the other BASELINE configs (parity-test configs, not the bench line):
config 3's per-GPU shard (1.25e8 x W=19), config 4 (W=30, 1e8 windows, --qvalueT 1e-4), config 5
(same-width PWM batches over 1e8 k-mers).  K-mers are generated on the device (i.i.d. bg_nt bases,
1 % planted PWM samples, 0.1 % rows with an N).  Prints one JSON object."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif, compute_log_odds_dense, scale_pwm_dense, score_multi
from grafimo_amd.motif_ops import build_motif_meme_host
from grafimo_amd.scan import KmerScanner

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")


def device_kmers(n, W, probs, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    alpha = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    bg = torch.tensor(synth.BG_NT / synth.BG_NT.sum(), device=dev)
    out = torch.empty((n, W), dtype=torch.uint8, device=dev)
    step = 25_000_000
    for a in range(0, n, step):
        b = min(n, a + step)
        idx = torch.multinomial(bg, (b - a) * W, replacement=True, generator=g).view(b - a, W)
        out[a:b] = alpha[idx]
    k = n // 100
    rows = torch.randint(0, n, (k,), generator=g, device=dev)
    cdf = torch.tensor(np.cumsum(probs / probs.sum(0, keepdims=True), axis=0), device=dev)   # [4, W]
    u = torch.rand((k, W), generator=g, device=dev, dtype=torch.float64)
    code = (u[:, None, :] > cdf[None, :, :]).sum(1).clamp(max=3)
    out[rows] = alpha[code]
    out[torch.randint(0, n, (n // 1000,), generator=g, device=dev), W // 2] = ord("N")
    return out


def torch_scores(d, sm, min_val):
    """plain torch restatement of compute_score_seq's integer part (score_sequences.py:370-388) on the
    device: the full-size cross-check of the HIP scores (exact integers)"""
    n, W = d.shape
    lut = torch.full((256,), 4, dtype=torch.int64, device=dev)
    for ch, c in ((b"Aa", 0), (b"Cc", 1), (b"Gg", 2), (b"Tt", 3)):
        for b_ in ch:
            lut[b_] = c
    smt = torch.tensor(np.asarray(sm, dtype=np.int64), device=dev)            # [4, W]
    smt = torch.cat([smt, torch.zeros((1, W), dtype=torch.int64, device=dev)])  # row 4: N
    out = torch.empty(n, dtype=torch.int32, device=dev)
    step = 10_000_000
    for a in range(0, n, step):
        blk = d[a:a + step]
        acc = torch.zeros(blk.shape[0], dtype=torch.int64, device=dev)
        has_n = torch.zeros(blk.shape[0], dtype=torch.bool, device=dev)
        for j in range(W):
            code = lut[blk[:, j].long()]
            acc += smt[:, j][code]
            has_n |= code == 4
        acc[has_n] = min_val
        out[a:a + step] = acc.to(torch.int32)
    return out


def timed(f, reps):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


res = {}
# ---- config 3 shard: CTCF W=19, 1.25e8 rows
m = build_motif_meme_host(os.path.join(ROOT, "tests/golden/ref_data/MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
probs = np.asarray(m.count_matrix)
dm = DeviceMotif(m.dense_score_matrix(), m.dense_bg(), m.min_val, m.scale, m.offset)
n = 125_000_000
d = device_kmers(n, 19, probs, 3)
sc = KmerScanner(dm, n, hit_capacity=n // 32, device=dev)
dm.profile_enable(64, 1)
dt = timed(lambda: sc.enqueue(d, 1e-4), 20); sc.finish(); torch.cuda.synchronize()
kms = dm.profile_read(); dm.profile_enable(0)
slot = sc.slots[(sc._turn - 1) % 2]
ref = torch_scores(d, m.dense_score_matrix(), m.min_val)
ok = bool(torch.equal(slot.scores[:n], ref))
res["config3_shard"] = dict(rows=n, W=19, step_ms=dt * 1e3, kmers_per_s=n / dt, score_kernel_ms=float(kms.mean()),
                            achieved_GBps=n * 23 / (float(kms.mean()) * 1e-3) / 1e9, scores_equal_torch_reference=ok)
del ref
del d, sc; dm.close(); torch.cuda.empty_cache()

# ---- config 4: W=30 JASPAR-style, 1e8 rows, --qvalueT -t 1e-4
rng = np.random.default_rng(20240139 + 4)
counts = np.rint(rng.dirichlet([0.3] * 4, size=30).T * 1000)
p30 = (counts / counts.sum(0) * counts.sum(0).astype(int) + 0.1 * 0.25) / (counts.sum(0).astype(int) + 0.1)
bgu = np.full(4, 0.25)
sm, mn, mx, scl, off = scale_pwm_dense(compute_log_odds_dense(p30, (bgu + 5e-7) / (1 + 2e-6)))
dm = DeviceMotif(sm, (bgu + 5e-7) / (1 + 2e-6), mn, scl, off)
n = 100_000_000
d = device_kmers(n, 30, p30, 4)
sc = KmerScanner(dm, n, hit_capacity=n // 32, device=dev)
dm.profile_enable(64, 1)
dt = timed(lambda: sc.enqueue(d, 1e-4, on_qvalue=True), 20); sc.finish(); torch.cuda.synchronize()
kms = dm.profile_read(); dm.profile_enable(0)
r = sc.collect(sc.slots[(sc._turn - 1) % 2])
ref = torch_scores(d, sm, mn)
ok4 = bool(torch.equal(sc.slots[(sc._turn - 1) % 2].scores[:n], ref))
del ref
res["config4"] = dict(scores_equal_torch_reference=ok4, rows=n, W=30, window_bins=dm.score_hi - dm.score_lo + 1, threshold="q<1e-4", hits=int(len(r["rows"])),
                      step_ms=dt * 1e3, kmers_per_s=n / dt, score_kernel_ms=float(kms.mean()),
                      achieved_GBps=n * 34 / (float(kms.mean()) * 1e-3) / 1e9)
del d, sc; dm.close(); torch.cuda.empty_cache()

# ---- config 5: three same-width PWMs per launch, widths 8, 12, 19, 25; 1e8 rows per width
cfg5 = []
for W in (8, 12, 19, 25):
    motifs, pl = [], None
    for k in range(3):
        pr = (rng.dirichlet([0.3] * 4, size=W).T * 1000 + 0.025) / 1000.1
        bg = rng.dirichlet(50 * synth.BG_NT)
        smk, mnk, mxk, sck, offk = scale_pwm_dense(compute_log_odds_dense(pr, bg))
        motifs.append(DeviceMotif(smk, bg, mnk, sck, offk)); pl = pr
        motifs[-1].sm_host, motifs[-1].min_val_host = smk, mnk
    n = 100_000_000
    d = device_kmers(n, W, pl, 50 + W)
    scores = [torch.empty(n, dtype=torch.int32, device=dev) for _ in motifs]
    hists = [torch.zeros(mo.L, dtype=torch.int64, device=dev) for mo in motifs]
    hits = [torch.zeros(n // 32 + 1, dtype=torch.int64, device=dev) for _ in motifs]
    cuts = [mo.pvalue_cutoff(1e-4) for mo in motifs]
    def batched():
        score_multi(motifs, d, scores, hists=hists, cutoffs=cuts, hit_rows=[h[1:] for h in hits],
                    hit_counts=[h[:1] for h in hits], reset_hits=True)
    def separate():
        for j, mo in enumerate(motifs):
            mo.score(d, scores[j], hist=hists[j], select_cutoff=cuts[j], hit_rows=hits[j][1:], hit_count=hits[j][:1],
                     reset_hits=True)
    tb, ts = timed(batched, 10), timed(separate, 10)
    batched(); torch.cuda.synchronize()
    ok5 = all(bool(torch.equal(scores[j], torch_scores(d, mo.sm_host, mo.min_val_host))) for j, mo in enumerate(motifs))
    cfg5.append(dict(W=W, motifs=3, rows=n, batched_ms=tb * 1e3, separate_ms=ts * 1e3, pairs_per_s_batched=3 * n / tb,
                     pairs_per_s_separate=3 * n / ts, speedup=ts / tb, scores_equal_torch_reference=ok5))
    for mo in motifs: mo.close()
    del d, scores, hits; torch.cuda.empty_cache()
res["config5_same_width_batches"] = cfg5
print(json.dumps(res))
