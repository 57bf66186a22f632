"""Development aid (GPU box): the report tables of the product's top-level calls over a graph -- compute_results_from_graph_many
and compute_results_from_graph -- against the CPU oracle end to end on random rich graphs, regions, motif sets and flags, seed
after seed for a fixed time.  One seed = tests/tables_fuzz_core.py (`pytest -m gpu` runs a bounded seed set of it).
TEST INFRASTRUCTURE (imports oracle/): not part of the product.
    python scripts/tables_fuzz.py [seconds] [first_seed]"""
import os
import sys
import tempfile
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))

from tables_fuzz_core import fuzz_seed

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time()
stats = dict(graphs=0, tables=0, rows_scanned=0, rows_reported=0)
with tempfile.TemporaryDirectory() as tmp:
    while time.time() - t0 < budget:
        fuzz_seed(seed, tmp, stats)
        seed += 1
print(f"tables_fuzz: {stats['graphs']} graphs, {stats['tables']} tables ({stats['rows_scanned']} rows scanned, {stats['rows_reported']} "
      f"reported) in {time.time() - t0:.0f} s: motif set == single calls == the oracle's tables; next seed {seed}")
