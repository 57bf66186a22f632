"""Development aid: time per step of back-to-back score launches against the kernel's own duration, with the pieces
of the scan step added one at a time (post kernel, tail stream, slot events, q-table)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.motif_ops import build_motif_meme_host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = build_motif_meme_host(os.path.join(ROOT, "tests/golden/ref_data/MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
dm = DeviceMotif(m.dense_score_matrix(), m.dense_bg(), m.min_val, m.scale, m.offset)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
dev = torch.device("cuda:0")
ds = [synth.make_device_kmers(n, m.width, np.asarray(m.count_matrix), 7 + i, dev) for i in range(2)]
sc = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]
hist = [torch.zeros(dm.L, dtype=torch.int64, device=dev) for _ in range(2)]
hits = [torch.zeros(n // 32 + 1, dtype=torch.int64, device=dev) for _ in range(2)]
q = torch.empty(dm.L, dtype=torch.float64, device=dev); cutd = torch.zeros(1, dtype=torch.int32, device=dev)
nr = torch.zeros(1, dtype=torch.int64, device=dev)
cut = dm.pvalue_cutoff(1e-4)
main = torch.cuda.current_stream(dev)
tail = torch.cuda.Stream(device=dev, priority=-1)
evs = [torch.cuda.Event() for _ in range(2)]
N = 200


def run(name, body):
    for i in range(10): body(i)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(N): body(i)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / N * 1e3)
    print(f"{name:60s} {best:8.2f} us/step", flush=True)


# the kernel's own duration (hipEvent pair around each launch)
dm.profile_enable(40)
for i in range(40): dm.score(ds[i & 1], sc[i & 1])
torch.cuda.synchronize()
ms = np.sort(dm.profile_read()[1:]); print(f"kernel alone (event pair per launch): median {np.median(ms)*1e3:.2f} us")
dm.profile_enable(0)

run("score only, one stream", lambda i: dm.score(ds[i & 1], sc[i & 1]))
run("score + hist (post kernel), one stream", lambda i: dm.score(ds[i & 1], sc[i & 1], hist=hist[i & 1]))
run("score + hist + select, one stream",
    lambda i: dm.score(ds[i & 1], sc[i & 1], hist=hist[i & 1], select_cutoff=cut, hit_rows=hits[i & 1][1:],
                       hit_count=hits[i & 1][:1], reset_hits=True))
run("score + hist + select, tail stream",
    lambda i: dm.score(ds[i & 1], sc[i & 1], hist=hist[i & 1], select_cutoff=cut, hit_rows=hits[i & 1][1:],
                       hit_count=hits[i & 1][:1], reset_hits=True, tail_stream=tail))


def full(i, slot_events=True, qtab=True):
    k = i & 1
    if slot_events: main.wait_event(evs[k])
    dm.score(ds[k], sc[k], hist=hist[k], select_cutoff=cut, hit_rows=hits[k][1:], hit_count=hits[k][:1],
             reset_hits=True, tail_stream=tail)
    if qtab: dm.qvalue_table(hist[k], 1e-4, False, q, cutd, nr, stream=tail, clear_hist=True)
    if slot_events: evs[k].record(tail)


run("... + q-table on the tail stream", lambda i: full(i, False, True))
run("... + slot events (torch, system-scope release)", lambda i: full(i, True, True))
run("... slot events without q-table", lambda i: full(i, True, False))
