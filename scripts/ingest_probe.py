"""Streamed scan (gfm_scan_tsv) against the number of parse threads at two sizes: 2e6 rows in 1000 files and 2e7 rows
in 10 000 files (what bench.py's e2e / e2e_config2 blocks run); median of 5 calls after a warm-up call.
usage: ingest_probe.py [trace]  (trace: one more call at 2e7 rows with GRAFIMO_SCAN_TRACE=1)"""
import glob, os, shutil, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ctcf = bench.load_ctcf()
probs = np.asarray(ctcf.count_matrix, dtype=np.float64)
dirs = {}
for regions in (1000, 10_000):
    dirs[regions] = bench.make_tsv_dir(regions, 2000, 19, probs, min(os.cpu_count(), 128))
import torch
from grafimo_amd.device import DeviceMotif
from grafimo_amd.score_sequences import StreamScan
dm = DeviceMotif.from_motif(ctcf)
try:
    for regions, (tmp, n) in dirs.items():
        files = sorted(glob.glob(os.path.join(tmp, "width_19", "*.tsv")))
        StreamScan(dm, files, False, 32, 1e-4, False, True)
        for th in (16, 24, 32, 48, 64, 96, 128):
            tot, par = [], []
            for _ in range(5):
                sc = StreamScan(dm, files, False, th, 1e-4, False, True)
                tot.append(sc.stats.total_s * 1e3); par.append(sc.stats.parse_s * 1e3)
            print(f"{n:9d} rows, threads={th:3d} (used {sc.stats.parse_threads:3d}): total median {np.median(tot):7.2f} ms "
                  f"(min {min(tot):7.2f}), parse median {np.median(par):7.2f} ms -> {n / np.median(tot) / 1e3:.0f} M rows/s", flush=True)
    if len(sys.argv) > 1:
        os.environ["GRAFIMO_SCAN_TRACE"] = "1"
        StreamScan(dm, files, False, 32, 1e-4, False, True)
finally:
    for tmp, _ in dirs.values():
        shutil.rmtree(tmp, ignore_errors=True)
