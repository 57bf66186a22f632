"""Inside compute_results_from_graph_many (BASELINE configs[4] through the graph) cProfile books 70-90 ms of a profiled call to
whichever trivial pandas function runs inside DataFrame.__init__ (new_block_2d, ensure_block_shape: a reshape) -- time that a
table built alone does not take (scripts/frame_gc_probe.py: 0.48 ms per table).  Is it the cyclic garbage collector running
over the process's live objects (torch, pandas, the tables kept so far)?  gc.callbacks gives its time exactly; the calls are
repeated with the collector off, with the older generations frozen, and as the product runs them."""
import contextlib, gc, io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grafimo_amd import synth
from grafimo_amd import extract_regions as xr
from grafimo_amd.workflow import Findmotif

dev = torch.device("cuda", 0)
motifs = [synth.motif_object(m, f"M{i}") for i, m in enumerate(synth.config_motifs(5))]
idx, regions = synth.make_graph_index(50_000, max(m.width for m in motifs))
g = xr.DeviceGraph(idx, dev)
reg = np.asarray(regions, dtype=np.int64)
wf = Findmotif(threshold=1e-4)

gc_t = {"t0": 0.0, "sum": 0.0, "n": [0, 0, 0]}


def on_gc(phase, info):
    if phase == "start":
        gc_t["t0"] = time.perf_counter()
    else:
        gc_t["sum"] += time.perf_counter() - gc_t["t0"]
        gc_t["n"][info["generation"]] += 1


gc.callbacks.append(on_gc)


def run(tag, reps=8, keep_tables=False):
    ts, gcs, kept = [], [], []
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(reps):
            gc_t["sum"], gc_t["n"] = 0.0, [0, 0, 0]
            t = time.perf_counter()
            tabs = xr.compute_results_from_graph_many(motifs, g, reg, False, wf)
            ts.append(1e3 * (time.perf_counter() - t))
            gcs.append((1e3 * gc_t["sum"], tuple(gc_t["n"])))
            if keep_tables:
                kept.append(tabs)
    v = ts[2:]
    print(f"{tag:44s} median {np.median(v):6.1f} ms  min {min(v):6.1f}   collector: {np.median([x for x, _ in gcs[2:]]):5.1f} ms per call, "
          f"collections by generation {gcs[-1][1]}   tracked objects {len(gc.get_objects())}")


run("as the product runs it")
run("as the product runs it, results kept", keep_tables=True)
gc.disable()
run("collector off")
gc.enable()
gc.collect()
gc.freeze()
run("older generations frozen (gc.freeze)")
gc.unfreeze()
old = gc.get_threshold()
gc.set_threshold(100_000, 50, 50)
run("threshold 100 000")
gc.set_threshold(*old)
run("as the product runs it (again)")
