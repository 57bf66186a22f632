"""Extraction kernel throughput at BASELINE config-2 scale: 10 000 regions x 200 bp on a synthetic
chromosome with 1000-Genomes-like variant density (1 site / 32 bp, 6 % of them deletions, 5096 haplotypes =
80 bitset words per allele), W = 19; then the whole extraction -> scoring pipeline on the device."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafimo_amd.extract_regions import DeviceGraph, GraphIndex
from grafimo_amd.device import DeviceMotif
from grafimo_amd.motif_ops import build_motif_meme_host
from grafimo_amd.scan import KmerScanner

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(20240139)
L, H, W = 12_000_000, 5096, 19
n_regions = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L, p=[0.2951, 0.2047, 0.2048, 0.2954])
pos = np.unique(rng.integers(0, L, L // 32)).astype(np.int32)
V = len(pos)
n_alts = np.where(rng.random(V) < 0.97, 1, 2).astype(np.uint8)
alt_bases = np.zeros((V, 3), np.uint8)
for a in range(2):
    alt_bases[:, a] = np.frombuffer(b"ACGT", np.uint8)[(np.searchsorted(np.frombuffer(b"ACGT", np.uint8), ref[pos]) + 1 + a) % 4]
alt_bases[n_alts < 2, 1] = 0
hw = (H + 63) // 64
# allele frequency spectrum skewed to rare variants; haplotype bits drawn per site
af = rng.random(V) ** 4
bits = np.zeros((V, 3, hw), np.uint64)
blk = 20000
for s in range(0, V, blk):
    e = min(V, s + blk)
    carry = rng.random((e - s, hw * 64)) < af[s:e, None]
    carry[:, H:] = False
    bits[s:e, 0, :] = np.packbits(carry, axis=1, bitorder="little").view(np.uint64)
# 6 % of the sites are deletions of 1..8 bases (1000-Genomes-like share), kept apart from each other
del_len = np.zeros(V, np.int32)
cand = np.nonzero(rng.random(V) < 0.06)[0]
last_end = -1
for i in cand:
    ln = int(rng.integers(1, 9))
    if pos[i] > last_end and pos[i] + ln < L - 1:
        del_len[i] = ln
        last_end = int(pos[i]) + ln
n_alts[del_len > 0] = 1
alt_bases[del_len > 0] = 0
bits[del_len > 0, 1:, :] = 0
idx = GraphIndex("22", ref, pos, n_alts, alt_bases, None if os.environ.get("EXTRACT_NO_COUNTS") else bits, H,
                 del_len=None if os.environ.get("EXTRACT_NO_DELS") else del_len)
regions = [(16_000 + 1000 * i, 16_000 + 1000 * i + 200) for i in range(n_regions)]
t = time.perf_counter(); g = DeviceGraph(idx); torch.cuda.synchronize(); t_up = time.perf_counter() - t

def run():
    t0 = time.perf_counter()
    rows = g.extract(regions, W)
    torch.cuda.synchronize()
    return rows, time.perf_counter() - t0

rows, _ = run()
times = []
for _ in range(5):
    rows, dt = run(); times.append(dt)
n = len(rows)
# emit kernel alone (the plan is cached inside the handle after extract)
from grafimo_amd import _native as nv
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(10):
    nv.check(nv.lib().gfm_graph_emit(g._h, rows.kmers.data_ptr(), rows.start.data_ptr(), rows.stop.data_ptr(),
                                     rows.strand.data_ptr(), rows.freq.data_ptr(), rows.is_ref.data_ptr(),
                                     rows.region.data_ptr(), rows.walk.data_ptr(), None))
ev1.record(); torch.cuda.synchronize()
emit_ms = ev0.elapsed_time(ev1) / 10
out_bytes = n * (W + 8 + 8 + 1 + 8 + 1 + 4 + 4)
res = dict(regions=n_regions, region_bp=200, W=W, sites=V, deletions=int((idx.del_len > 0).sum()), haplotypes=H, rows=n, windows=n_regions * (200 - W + 1),
           graph_upload_s=t_up, extract_wall_ms=1e3 * float(np.median(times)), emit_kernel_ms=emit_ms,
           rows_per_s_emit=n / (emit_ms * 1e-3), rows_per_s_wall=n / float(np.median(times)),
           emit_written_GBps=out_bytes / (emit_ms * 1e-3) / 1e9,
           nonref_fraction=float(1 - rows.is_ref.float().mean().item()))
# extraction -> scoring on the device
m = build_motif_meme_host(os.path.join(ROOT, "tests/golden/ref_data/MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
dm = DeviceMotif(m.dense_score_matrix(), m.dense_bg(), m.min_val, m.scale, m.offset)
sc = KmerScanner(dm, n, device=rows.kmers.device, side_stream=False)
def pipeline():
    r = g.extract(regions, W)
    slot = sc.enqueue(r.kmers, 1e-4)
    torch.cuda.synchronize()
    return slot
pipeline()
t0 = time.perf_counter()
for _ in range(5): slot = pipeline()
res["extract_plus_score_ms"] = 1e3 * (time.perf_counter() - t0) / 5
res["hits_p1e-4"] = int(len(sc.collect(slot)["rows"]))
print(json.dumps(res))
