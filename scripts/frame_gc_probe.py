"""Why pandas' new_block_2d shows 1.8 ms per 8 800-row table inside the product call and 0.2 ms alone: garbage collection?"""
import gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pandas as pd
n, W, M, NR = 8800, 19, 50, 50000
rng = np.random.default_rng(1)
labels = np.array([f"chr22:{16000000 + 1000 * i}-{16000200 + 1000 * i}" for i in range(NR)], dtype=object)
STR = np.array(["+", "-"], dtype=object)

def one():
    km = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, W + 1))].copy(); km[:, W] = 10
    return rng.integers(0, NR, n), rng.integers(0, 1 << 30, n), rng.integers(0, 2, n).astype(np.uint8), km, rng.random(n)

inputs = [one() for _ in range(M)]

def build(inp):
    reg, st, sidx, km, sc = inp
    seqs = km.tobytes().decode().split("\n"); seqs.pop()
    ids = np.empty(n, dtype=object); ids[:] = "M1"
    data = {"motif_id": ids, "motif_alt_id": ids.copy(), "sequence_name": labels[reg], "start": st, "stop": st + W, "strand": STR[sidx],
            "score": sc, "p-value": sc, "q-value": sc, "matched_sequence": np.array(seqs, dtype=object), "haplotype_frequency": st, "reference": STR[sidx]}
    return pd.DataFrame(data, copy=False)

def run(tag):
    ts = []
    for _ in range(5):
        t = time.perf_counter(); dfs = [build(i) for i in inputs]; ts.append(time.perf_counter() - t); del dfs
    print(f"{tag:34s} 50 tables: min {1e3 * min(ts):.1f} ms  median {1e3 * sorted(ts)[2]:.1f} ms   gc counts {gc.get_count()} tracked {len(gc.get_objects())}")

run("plain python + pandas")
import torch
torch.cuda.init() if torch.cuda.is_available() else None
import grafimo_amd.extract_regions, grafimo_amd.score_sequences
run("after import torch + grafimo_amd")
gc.disable(); run("gc disabled"); gc.enable()
gc.freeze(); run("gc.freeze()"); gc.unfreeze()
gc.set_threshold(100000, 50, 50); run("threshold 100000"); gc.set_threshold(700, 10, 10)
keep = [build(i) for i in inputs * 4]          # 200 live tables: what a caller that keeps its results holds
run("with 200 live tables")
gc.disable(); run("with 200 live tables, gc disabled"); gc.enable()
