"""Development aid: phase times of gfm_scan_tsv (GRAFIMO_SCAN_TRACE) on the bench's e2e directory."""
import glob, os, shutil, sys, tempfile, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif
from grafimo_amd.motif_ops import build_motif_meme_host
from grafimo_amd.score_sequences import StreamScan
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
regions = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m = build_motif_meme_host(os.path.join(ROOT, "tests/golden/ref_data/MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
dm = DeviceMotif(m.dense_score_matrix(), m.dense_bg(), m.min_val, m.scale, m.offset)
tmp = tempfile.mkdtemp(prefix="grafimo_e2e_", dir="/dev/shm")
try:
    batch = synth.make_batch(regions, 2000, 19, np.asarray(m.count_matrix), synth.seed_for(7))
    synth.write_tsv_dir(batch, tmp)
    files = sorted(glob.glob(os.path.join(tmp, "width_19", "*.tsv")))
    counts = [int(a) for a in sys.argv[2:]] or [os.cpu_count() or 1]
    for threads in counts:
        tot = []
        for i in range(8):
            t = time.perf_counter()
            sc = StreamScan(dm, files, False, threads, 1e-4, False, True)
            tot.append((sc.stats.total_s * 1e3, sc.stats.parse_s * 1e3, sc.stats.tail_s * 1e3, 1e3 * (time.perf_counter() - t)))
        tot = sorted(tot[1:])
        print(f"threads {threads:4d} (used {sc.stats.parse_threads}): total ms " + " ".join(f"{x[0]:.2f}" for x in tot)
              + f"   median: total {tot[3][0]:.2f} parse {tot[3][1]:.2f} tail {tot[3][2]:.2f} wall {tot[3][3]:.2f}", flush=True)
    os.environ["GRAFIMO_SCAN_TRACE"] = "1"
    StreamScan(dm, files, False, counts[-1], 1e-4, False, True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
