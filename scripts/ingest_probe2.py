"""Pure open+read+close scaling of the library's file reader (gfm_tsv_open with a wrong width: every file is read,
then fails on its first row) against the thread count, on /dev/shm and on /tmp."""
import ctypes, glob, os, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from grafimo_amd import _native as nv
ctcf = bench.load_ctcf()
probs = np.asarray(ctcf.count_matrix, dtype=np.float64)
tmp, n = bench.make_tsv_dir(10_000, 2000, 19, probs, min(os.cpu_count(), 128))
alt = tempfile.mkdtemp(prefix="grafimo_probe_", dir="/tmp")
try:
    shutil.copytree(os.path.join(tmp, "width_19"), os.path.join(alt, "width_19"))
    for root in (tmp, alt):
        files = sorted(glob.glob(os.path.join(root, "width_19", "*.tsv")))
        nbytes = sum(os.path.getsize(f) for f in files)
        arr = (ctypes.c_char_p * len(files))(*[f.encode() for f in files])
        for W, what in ((18, "read only"), (19, "read + parse")):
            for th in (8, 16, 32, 48, 64, 96, 128):
                best = []
                for _ in range(3):
                    h, nn = ctypes.c_void_p(), ctypes.c_int64()
                    t = time.perf_counter()
                    rc = nv.lib().gfm_tsv_open(arr, len(files), W, 0, th, ctypes.byref(h), ctypes.byref(nn))
                    best.append(time.perf_counter() - t)
                    if rc == 0:
                        nv.lib().gfm_tsv_close(h)
                print(f"{root[:8]} {what:13s} threads={th:3d}: {[round(b * 1e3, 1) for b in best]} ms ({nbytes / min(best) / 1e9:.1f} GB/s)", flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
    shutil.rmtree(alt, ignore_errors=True)
