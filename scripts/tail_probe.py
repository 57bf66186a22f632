"""What the per-step tail costs the score kernel that runs beside it (config 2, pipelined, three slots): kernel time
from the dispatch-packet events with (a) the full tail (post + three q-table kernels), (b) the post kernel only,
(c) the tail on the main stream (no overlap), (d) no histogram and no selection at all."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from grafimo_amd import _native as nv, synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.scan import KmerScanner

ctcf = bench.load_ctcf()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
probs = np.asarray(ctcf.count_matrix, dtype=np.float64)
n = 20_000_000
first = synth.make_device_kmers(n, 19, probs, 7, dev)
bufs = [first, torch.roll(first, shifts=n // 3, dims=0).contiguous()]
dm = DeviceMotif.from_motif(ctcf)
lib = nv.lib()

def run(label, side, want_q, hist=True, select=True, steps=300):
    sc = KmerScanner(dm, n, hit_capacity=n // 64, device=dev, side_stream=side, n_slots=3)
    cut = dm.pvalue_cutoff(1e-4)
    main = torch.cuda.current_stream(dev)
    def step(i):
        slot = sc.slots[i % 3]
        if slot.used:
            slot.done.synchronize()
        slot.used = True
        tail = sc.side if side else main
        nv.check(lib.gfm_score_kmers(dm.handle, bufs[i % 2].data_ptr(), n, slot.p_scores, slot.p_hist if hist else None,
                                     cut if select else nv.GFM_NO_SELECT, 0, slot.p_hit_rows if select else None,
                                     slot.hit_capacity if select else 0, slot.p_hit_count if select else None,
                                     nv.GFM_FLAG_RESET_HITS | nv.GFM_FLAG_CALLER_ORDERS_REUSE, main.cuda_stream,
                                     tail.cuda_stream if side else None))
        if want_q and hist:
            nv.check(lib.gfm_qvalue_table(dm.handle, slot.p_hist, 1e-4, 0, slot.p_qtable, slot.p_cutoff, slot.p_nrows,
                                          nv.GFM_FLAG_CLEAR_HIST, tail.cuda_stream))
        elif hist:
            slot.hist.zero_() if not side else None
        slot.done.record(tail)
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    dm.profile_enable(64, every=4)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    k = dm.profile_read()
    dm.profile_enable(0)
    print(f"{label:46s} step {1e6 * el / steps:7.2f} us   kernel {1e3 * float(np.mean(k)):7.2f} us (min {1e3 * float(k.min()):.2f})", flush=True)

for rep in range(2):
    run("full tail beside (post + q-table)", True, True)
    run("post only beside", True, False)
    run("full tail on the main stream", False, True)
    run("post only on the main stream", False, False)
    run("no histogram, selection only, beside", True, False, hist=False)
    run("histogram only, no selection, beside", True, True, select=False)
    run("neither (scores only)", True, False, hist=False, select=False)
