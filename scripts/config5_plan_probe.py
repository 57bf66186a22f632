"""Development aid: how the 50 PWMs of BASELINE configs[4] are grouped into batched launches (group size, waves per
workgroup) and how wide their score ranges are -- what decides whether three motifs of a width share one k-mer read."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from grafimo_amd import synth
from grafimo_amd.device import DeviceMotif, multi_plan

mots = synth.config_motifs(5)
dms = [DeviceMotif(m["sm"], m["bg"], m["min_val"], m["scale"], m["offset"]) for m in mots]
for w in sorted({m["width"] for m in mots}):
    ix = [j for j, m in enumerate(mots) if m["width"] == w]
    sizes, waves = multi_plan([dms[j] for j in ix])
    rng = []
    for j in ix:
        sm = np.asarray(mots[j]["sm"]).reshape(4, -1)
        rng.append(int(sm.max(0).sum() - sm.min(0).sum()) + 1)
    print(f"W={w:2d} motifs={len(ix)} groups={list(map(int, sizes))} waves={list(map(int, waves))} score ranges={rng} "
          f"strips16={16 * 256 * w // 1024} KiB")
