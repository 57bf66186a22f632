"""Where the HOST's time goes in the product calls of the fused graph path (VERDICT r5 Weak #4): cProfile of
  (a) compute_results_from_graph_many, BASELINE configs[4] through the graph (50 PWMs, 50 000 regions),
  (b) scan_graph + the first and the later compute_results calls of the manifest route (CTCF, 10 000 regions).
usage: python scripts/host_profile.py [a] [b]   (default: both)"""
import contextlib, cProfile, io, os, pstats, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from grafimo_amd import synth
from grafimo_amd import extract_regions as xr
from grafimo_amd.score_sequences import compute_results
from grafimo_amd.workflow import Findmotif

what = set(sys.argv[1:]) or {"a", "b"}
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
sink = io.StringIO()


def prof(fn, reps=1, top=28):
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(reps):
        fn()
    pr.disable()
    for key in ("cumulative", "tottime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(top)
        print("\n".join(s.getvalue().splitlines()[4:]))


if "a" in what:
    motifs = [synth.motif_object(m, f"M{i}") for i, m in enumerate(synth.config_motifs(5))]
    idx, regions = synth.make_graph_index(50_000, max(m.width for m in motifs))
    g = xr.DeviceGraph(idx, dev)
    reg = np.asarray(regions, dtype=np.int64)
    wf = Findmotif(threshold=1e-4)
    with contextlib.redirect_stdout(sink):
        ts = []
        for _ in range(4):
            t = time.perf_counter()
            tabs = xr.compute_results_from_graph_many(motifs, g, reg, False, wf)
            ts.append(time.perf_counter() - t)
    print(f"=== (a) compute_results_from_graph_many, 50 PWMs x 50 000 regions: {[round(1e3 * x, 1) for x in ts]} ms, "
          f"{sum(len(t_) for t_ in tabs)} hit rows")
    with contextlib.redirect_stdout(sink):
        pr = cProfile.Profile(); pr.enable()
        xr.compute_results_from_graph_many(motifs, g, reg, False, wf)
        pr.disable()
    for key in ("cumulative", "tottime"):
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(30)
        print("\n".join(s.getvalue().splitlines()[4:]))
    g.close()

if "b" in what:
    ctcf = bench.load_ctcf()
    idx, regions = synth.make_graph_index(10_000, 19)
    tmp = tempfile.mkdtemp(prefix="gfm_prof_")
    idx.save(os.path.join(tmp, "chr22"))
    bed = os.path.join(tmp, "regions.bed")
    with open(bed, "w") as fh:
        fh.write("".join(f"chr22\t{s}\t{e}\n" for s, e in regions))
    wf = Findmotif(cores=8, threshold=1e-4, graph_genome_dir=tmp, bedfile=bed, chroms_prefix="chr")
    os.environ["GRAFIMO_SCAN_OUTPUT"] = "manifest"
    with contextlib.redirect_stdout(sink):
        t = time.perf_counter(); loc = xr.scan_graph({19}, wf, False); t_scan = time.perf_counter() - t
        shutil.rmtree(loc)
        pr = cProfile.Profile(); pr.enable(); loc = xr.scan_graph({19}, wf, False); pr.disable()
    print(f"=== (b) scan_graph (manifest) {1e3 * t_scan:.2f} ms")
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print("\n".join(s.getvalue().splitlines()[4:]))
    with contextlib.redirect_stdout(sink):
        pr = cProfile.Profile(); t = time.perf_counter(); pr.enable(); compute_results(ctcf, loc, False, wf); pr.disable()
        t_first = time.perf_counter() - t
    print(f"=== (b) first compute_results {1e3 * t_first:.2f} ms (under cProfile)")
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30); print("\n".join(s.getvalue().splitlines()[4:]))
    with contextlib.redirect_stdout(sink):
        ts = []
        for _ in range(30):
            t = time.perf_counter(); compute_results(ctcf, loc, False, wf); ts.append(time.perf_counter() - t)
        pr = cProfile.Profile(); pr.enable()
        for _ in range(50):
            compute_results(ctcf, loc, False, wf)
        pr.disable()
    print(f"=== (b) later compute_results: median {1e3 * float(np.median(ts)):.3f} ms; profile of 50 calls")
    for key in ("cumulative", "tottime"):
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(32); print("\n".join(s.getvalue().splitlines()[4:]))
    shutil.rmtree(loc); shutil.rmtree(tmp)
