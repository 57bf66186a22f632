import sys, os, json, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from grafimo_amd.device import DeviceMotif
from oracle import oracle as orc
from test_gpu_parity import random_kmers
cases=json.load(open("tests/golden/motifs.json")); pmfs=np.load("tests/golden/pmf.npz")
m=cases["ctcf_meme_unif"]["motifs"][0]
sm=np.array(m["score_matrix"]); 
dm=DeviceMotif(sm, np.array(m["bg"]), m["min_val"], m["scale"], m["offset"], pmfs[m["pmf_key"]])
_,pt=dm.tables()
rng=np.random.default_rng(19)
for n in [0,1,63,64,65,127,128,129,1000]:
    km=random_kmers(rng,n,19,n_frac=0.01)
    if n==0: continue
    d_km=torch.from_numpy(km).cuda(); d_sc=torch.full((n,),-7,dtype=torch.int32,device="cuda")
    dm.score(d_km,d_sc); torch.cuda.synchronize()
    got=d_sc.cpu().numpy(); exp,_=orc.score_kmers_table(km,sm,pt,m["min_val"])
    bad=np.nonzero(got!=exp)[0]
    print(n, "bad rows", bad[:20], [ (km[i].tobytes(), int(got[i]), int(exp[i])) for i in bad[:5]])
