"""End-to-end probe: TSV ingest rate and PCIe-inclusive scan rate (numbers quoted in DESIGN.md section 6)."""
import os, sys, time, tempfile, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafimo_amd import synth
from grafimo_amd.motif_ops import build_motif_meme_host
from grafimo_amd.score_sequences import KmerTable
from grafimo_amd.device import DeviceMotif
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = build_motif_meme_host(os.path.join(ROOT, "tests/golden/ref_data/MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
regions = int(sys.argv[1]) if len(sys.argv) > 1 else 500
b = synth.make_batch(regions, 2000, 19, np.asarray(m.count_matrix), 7)
d = tempfile.mkdtemp(prefix="gfm_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
t = time.time(); synth.write_tsv_dir(b, d); print(f"wrote {len(b)} rows in {time.time()-t:.1f}s")
files = sorted(os.path.join(d, "width_19", f) for f in os.listdir(os.path.join(d, "width_19")))
size = sum(os.path.getsize(f) for f in files)
for thr in (1, 8, 32, os.cpu_count()):
    t = time.time(); tab = KmerTable(files, 19, False, thr); dt = time.time() - t
    print(f"ingest threads={thr}: {tab.n/dt/1e6:.2f} M rows/s ({size/dt/1e9:.2f} GB/s of text)")
dm = DeviceMotif(m.dense_score_matrix(), m.dense_bg(), m.min_val, m.scale, m.offset)
dm.scan_host(tab.kmers, 1e-4)
t = time.time(); res = dm.scan_host(tab.kmers, 1e-4); dt = time.time() - t
print(f"scan_host (H2D + kernels + D2H hits): {tab.n/dt/1e6:.1f} M k-mers/s, {len(res['rows'])} hits")
import shutil; shutil.rmtree(d)
