"""One compute_results call through the manifest (CTCF, 10 000 regions), medians of 200 calls, six rounds in one process.  Round 6
used it to compare the q-table as ONE 1024-thread workgroup in one launch against the three small multi-block kernels,
alternating: 351-353 against 333-340 us per call -- the single workgroup is slower, the code is gone (xr._Q_ALONE with it: both
modes below run the product)."""
import contextlib, io, os, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from grafimo_amd import synth
from grafimo_amd import extract_regions as xr
from grafimo_amd.score_sequences import compute_results
from grafimo_amd.workflow import Findmotif
ctcf = bench.load_ctcf()
idx, regions = synth.make_graph_index(10_000, 19)
tmp = tempfile.mkdtemp(prefix="gfm_ab_")
idx.save(os.path.join(tmp, "chr22"))
bed = os.path.join(tmp, "regions.bed")
open(bed, "w").write("".join(f"chr22\t{s}\t{e}\n" for s, e in regions))
wf = Findmotif(cores=8, threshold=1e-4, graph_genome_dir=tmp, bedfile=bed, chroms_prefix="chr")
os.environ["GRAFIMO_SCAN_OUTPUT"] = "manifest"
with contextlib.redirect_stdout(io.StringIO()):
    loc = xr.scan_graph({19}, wf, False)
    for _ in range(20):
        compute_results(ctcf, loc, False, wf)
    for rnd in range(3):
        for mode in (True, False):
            xr._Q_ALONE = mode
            ts = []
            for _ in range(200):
                t = time.perf_counter()
                compute_results(ctcf, loc, False, wf)
                ts.append(time.perf_counter() - t)
            print(f"round {rnd}  q-table in one launch = {mode!s:5s}  compute_results median {1e6 * np.median(ts):.0f} us  min {1e6 * min(ts):.0f}", file=sys.stderr)
shutil.rmtree(loc); shutil.rmtree(tmp)
