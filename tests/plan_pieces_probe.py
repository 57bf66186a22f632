"""Helper of tests/test_gpu_extract.py::test_regions_beyond_one_plan_are_cut_into_pieces: runs in a process of its own because
the plan's row cap (GRAFIMO_PLAN_MAX_WALKS, a test aid of gfm_graph_plan_windows) is read once per process.
    python plan_pieces_probe.py rows OUT.npz [DIR]   extract() over five regions -> the rows, the number of pieces; with DIR also
                                                  the TSV files, written piece by piece the way scan_graph does
    python plan_pieces_probe.py single            a region whose densest window alone exceeds the cap -> the error text"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grafimo_amd import _native as nv                                  # noqa: E402
from grafimo_amd.extract_regions import (DeviceGraph, GraphIndex, finish_region_tsvs, region_file_names,     # noqa: E402
                                         write_region_tsvs)

rng = np.random.default_rng(3)
ref = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 900)]
pos = np.concatenate([np.arange(300, 316), np.arange(600, 614)]).astype(np.int32)      # 2^16 and 2^14 walks per full window
alt = np.zeros((len(pos), 3), np.uint8)
alt[:, 0] = np.where(ref[pos] == ord("A"), ord("C"), ord("A"))
g = DeviceGraph(GraphIndex("c", ref, pos, np.ones(len(pos), np.uint8), alt, None, 0))
cap = int(os.environ.get("GRAFIMO_PLAN_MAX_WALKS", 0x3fffffff))
if sys.argv[1] == "rows":
    regions = [(0, 120), (280, 340), (100, 130), (590, 640), (700, 900)]
    parts = list(g.extract_chunks(regions, 24))
    rows = g.extract(regions, 24)
    assert all(len(p) <= 2 * cap for p in parts) and sum(len(p) for p in parts) == len(rows)
    print(f"pieces {len(parts)} rows {len(rows)}")
    np.savez(sys.argv[2], km=rows.kmers.cpu().numpy(), st=rows.start.cpu().numpy(), sp=rows.stop.cpu().numpy(),
             rg=rows.region.cpu().numpy(), wk=rows.walk.cpu().numpy(), fr=rows.freq.cpu().numpy(),
             sd=rows.strand.cpu().numpy(), rf=rows.is_ref.cpu().numpy())
    if len(sys.argv) > 3:           # a region cut into several plans is appended to, the rowless regions get their empty files
        labels = [f"c:{s}-{e}" for s, e in regions]
        seen = np.zeros(len(regions), dtype=np.uint8)
        for part in parts:
            write_region_tsvs(g.index, part, sys.argv[3], labels=labels, seen=seen, threads=3)
        finish_region_tsvs([os.path.join(sys.argv[3], "width_24", f) for f in region_file_names(labels)], seen)
else:
    try:
        g.extract([(0, 120), (280, 340)], 24)
        print("NOT REFUSED")
    except nv.NativeError as e:
        print("REFUSED", e.code, str(e))
g.close()
