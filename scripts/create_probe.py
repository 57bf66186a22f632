import os, sys, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafimo_amd.device import DeviceMotif, comp_pval_mat_dense
cases = json.load(open("tests/golden/motifs.json"))
torch.cuda.init(); torch.zeros(1, device="cuda")
for name in ["ctcf_meme_unif", "syn30_jaspar_unif", "multi_meme_bg1"]:
    m = cases[name]["motifs"][0]
    sm, bg = np.array(m["score_matrix"]), np.array(m["bg"])
    DeviceMotif(sm, bg, m["min_val"], m["scale"], m["offset"]).close()
    t = time.perf_counter()
    for _ in range(20):
        dm = DeviceMotif(sm, bg, m["min_val"], m["scale"], m["offset"]); dm.close()
    dt = (time.perf_counter() - t) / 20
    t = time.perf_counter()
    for _ in range(20): comp_pval_mat_dense(sm, bg)
    dt2 = (time.perf_counter() - t) / 20
    print(f"{name} W={m['width']}: motif create+destroy {dt*1e3:.2f} ms, comp_pval_mat (DP + D2H) {dt2*1e3:.2f} ms")
