import glob, os, shutil, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
ctcf = bench.load_ctcf()
probs = np.asarray(ctcf.count_matrix, dtype=np.float64)
n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
tmp, n = bench.make_tsv_dir(n_files, 2000, 19, probs, min(os.cpu_count(), 128))
import torch
from grafimo_amd.device import DeviceMotif
from grafimo_amd.score_sequences import StreamScan
dm = DeviceMotif.from_motif(ctcf)
files = sorted(glob.glob(os.path.join(tmp, "width_19", "*.tsv")))
StreamScan(dm, files, False, 32, 1e-4, False, True)
os.environ["GRAFIMO_SCAN_TRACE"] = "1"
for th in ((32, 64, 64, 64, 96) if n_files > 2000 else (64, 96, 96)):
    print("=== threads", th, flush=True)
    sc = StreamScan(dm, files, False, th, 1e-4, False, True)
    print("total %.1f ms parse %.1f ms" % (sc.stats.total_s*1e3, sc.stats.parse_s*1e3), flush=True)
shutil.rmtree(tmp, ignore_errors=True)
