"""Extraction kernel throughput at BASELINE config-2 scale (bench.py's `extract` block on its own): 10 000 regions x
200 bp on a synthetic chromosome with 1000-Genomes-like variant density (grafimo_amd.synth.make_graph_index), W = 19;
then the whole extraction -> scoring pipeline.  EXTRACT_NO_COUNTS=1 / EXTRACT_NO_DELS=1: without haplotype bitsets /
without deletions.  Under rocprofv3 --kernel-trace --stats this gives profiles/r03_extract_kernel_stats.csv."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from grafimo_amd import synth

if os.environ.get("EXTRACT_NO_COUNTS") or os.environ.get("EXTRACT_NO_DELS"):
    orig = synth.make_graph_index
    synth.make_graph_index = lambda n, w, **kw: orig(n, w, with_counts=not os.environ.get("EXTRACT_NO_COUNTS"),
                                                     with_dels=not os.environ.get("EXTRACT_NO_DELS"), **kw)
n_regions = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
torch.cuda.set_device(0)
print(json.dumps(bench.extract_block(bench.load_ctcf(), torch.device("cuda", 0), n_regions)))
