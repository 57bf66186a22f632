"""Development aid (GPU box): compute_results -- the drop-in entry point (TSV directory -> streamed scan -> table) -- against
the CPU oracle's restatement of the reference's compute_results on random synthetic directories (regions, rows, files per
region, empty files) and flag settings, seed after seed for a fixed time.  TEST INFRASTRUCTURE (imports oracle/).
    python scripts/results_fuzz.py [seconds] [first_seed]"""
import contextlib
import io
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import json
from conftest import GOLDEN
from test_gpu_compute_results import _compare, _ctcf
from grafimo_amd import synth
from grafimo_amd.score_sequences import compute_results
from grafimo_amd.workflow import Findmotif
from oracle import oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
with open(os.path.join(GOLDEN, "motifs.json")) as fh:
    _cases = json.load(fh)
_pmf = np.load(os.path.join(GOLDEN, "pmf.npz"))
flat = {}
for _case in _cases.values():
    for rec in _case["motifs"]:
        rec = dict(rec)
        rec["pmf"] = _pmf[rec["pmf_key"]]
        rec["probs"] = np.array(rec["probs"], dtype=np.float64)
        rec["score_matrix"] = np.array(rec["score_matrix"], dtype=np.int64)
        flat[rec["pmf_key"]] = rec
g = flat["ctcf_meme_unif#0"]
motif = _ctcf(True)
md = dict(score_matrix=g["score_matrix"], pmf=g["pmf"], min_val=g["min_val"], scale=g["scale"], offset=g["offset"], width=19,
          motif_id=g["motif_id"], motif_name=g["motif_name"])
t0 = time.time()
cases = rows = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    with tempfile.TemporaryDirectory() as tmp:
        n_regions = int(rng.integers(1, 60))
        rows_per_region = int(rng.choice([364, 400, 700, 3000]))
        batch = synth.make_batch(n_regions, rows_per_region, 19, g["probs"], synth.seed_for(seed))
        synth.write_tsv_dir(batch, tmp, regions_per_file=int(rng.integers(1, 6)))
        if rng.random() < 0.3:                                    # an empty file among them
            open(os.path.join(tmp, "width_19", "zz_empty.tsv"), "w").close()
        kw = dict(threshold=float(rng.choice([1e-4, 1e-3, 1e-2, 0.3])))
        mode = int(rng.integers(0, 4))
        if mode == 1:
            kw.update(qval_t=True, threshold=float(rng.choice([0.05, 0.5, 0.9])))
        elif mode == 2:
            kw.update(no_reverse=True, recomb=bool(rng.random() < 0.5))
        elif mode == 3:
            kw.update(no_qvalue=True, recomb=True)
        wf = Findmotif(cores=int(rng.choice([1, 3, 16, 64])), **kw)
        ref = orc.compute_results(md, tmp, threshold=kw["threshold"], qval_t=kw.get("qval_t", False),
                                  no_qvalue=kw.get("no_qvalue", False), no_reverse=kw.get("no_reverse", False),
                                  recomb=kw.get("recomb", False))
        exp = pd.DataFrame({c: ref[c] for c in ref if not c.startswith("_")})
        with contextlib.redirect_stdout(io.StringIO()):
            try:
                df = compute_results(motif, tmp, True, wf)
            except Exception:
                print("FAILED case:", seed, n_regions, rows_per_region, kw, file=sys.stderr)
                raise
        if len(exp) == 0:
            assert df is None or len(df) == 0, (seed, kw)
        else:
            _compare(df, exp)
        cases += 1
        rows += batch.n if hasattr(batch, "n") else n_regions * rows_per_region
    seed += 1
print(f"results_fuzz: {cases} directories, {rows} rows in {time.time() - t0:.0f} s: compute_results == the oracle's table; "
      f"next seed {seed}")
