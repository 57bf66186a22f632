"""Development aid (GPU box): scripts/micro/read_modes on bench.py's 10 000-file TSV directory."""
import os, shutil, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ctcf = bench.load_ctcf()
tmp, n = bench.make_tsv_dir(10_000, 2000, 19, np.asarray(ctcf.count_matrix, dtype=np.float64), min(os.cpu_count(), 128))
try:
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "read_modes")
    for th in (16, 32):
        print(subprocess.run([exe, os.path.join(tmp, "width_19"), str(th)], capture_output=True, text=True).stdout, flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
