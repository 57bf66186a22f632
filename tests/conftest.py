import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_DATA = os.path.join(GOLDEN, "ref_data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_motifs():
    with open(os.path.join(GOLDEN, "motifs.json")) as fh:
        cases = json.load(fh)
    pmf = np.load(os.path.join(GOLDEN, "pmf.npz"))
    flat = {}
    for name, case in cases.items():
        for k, rec in enumerate(case["motifs"]):
            rec = dict(rec)
            rec["pmf"] = pmf[rec["pmf_key"]]
            rec["case"] = name
            rec["probs"] = np.array(rec["probs"], dtype=np.float64)
            rec["bg"] = np.array(rec["bg"], dtype=np.float64)
            rec["logodds"] = np.array(rec["logodds"], dtype=np.float64)
            rec["score_matrix"] = np.array(rec["score_matrix"], dtype=np.int64)
            flat[rec["pmf_key"]] = rec
    return cases, flat


@pytest.fixture(scope="session")
def golden_json():
    def load(name):
        with open(os.path.join(GOLDEN, name)) as fh:
            return json.load(fh)
    return load


def kmers_from_strings(seqs):
    w = len(seqs[0])
    return np.frombuffer("".join(seqs).encode(), dtype=np.uint8).reshape(len(seqs), w).copy()
