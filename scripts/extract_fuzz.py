"""Development aid (GPU box): the extraction kernels against the per-haplotype brute force (oracle/extract_bruteforce.py)
and the walk enumerator (oracle/extract_oracle.py) on random conflict-free graphs of every allele kind, seed after seed
for a fixed time; and the FUSED extraction -> scoring path (compute_results_from_graph at threshold 1 with --recomb: every
row is reported) against those same rows, scored with the motif's integer matrix.  One seed = tests/extract_fuzz_core.py
(`pytest -m gpu` runs a bounded seed set of it).
TEST INFRASTRUCTURE (imports oracle/): not part of the product.
    python scripts/extract_fuzz.py [seconds] [first_seed]"""
import os
import sys
import tempfile
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))

from extract_fuzz_core import fuzz_seed

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time()
stats = dict(graphs=0, rows=0, carried=0, heavy=0, fused=0)
with tempfile.TemporaryDirectory() as tmp:
    while time.time() - t0 < budget:
        fuzz_seed(seed, tmp, stats, log=lambda m: print(m, flush=True))
        seed += 1
print(f"extract_fuzz: {stats['graphs']} graphs, {stats['rows']} rows ({stats['carried']} keys carried by a haplotype) in "
      f"{time.time() - t0:.0f} s: kernels == enumerator == brute force, {stats['fused']} plans also through the fused scoring "
      f"path ({stats['heavy']} plans with a window of more than 2^17 walks: brute force only); next seed {seed}")
