"""Streamed scan of 2e7 rows against the parse thread count WITHOUT the library's caps, every run behind 0.15 s of idle (a
fresh CPU-quota period): how many threads pay once a run is charged for its own CPU time only."""
import glob, os, shutil, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GRAFIMO_PARSE_THREADS_EXACT"] = "1"
import bench
ctcf = bench.load_ctcf()
probs = np.asarray(ctcf.count_matrix, dtype=np.float64)
tmp, n = bench.make_tsv_dir(10_000, 2000, 19, probs, min(os.cpu_count(), 128))
import torch
from grafimo_amd.device import DeviceMotif
from grafimo_amd.score_sequences import StreamScan
dm = DeviceMotif.from_motif(ctcf)
try:
    files = sorted(glob.glob(os.path.join(tmp, "width_19", "*.tsv")))
    StreamScan(dm, files, False, 32, 1e-4, False, True)
    for th in (16, 24, 32, 48, 64, 96, 128):
        tot = []
        for _ in range(7):
            time.sleep(0.15)
            sc = StreamScan(dm, files, False, th, 1e-4, False, True)
            tot.append(sc.stats.total_s * 1e3)
        print(f"{n} rows, threads={th:3d} (used {sc.stats.parse_threads:3d}): total median {np.median(tot):7.2f} ms (min {min(tot):7.2f}, max {max(tot):7.2f})", flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
