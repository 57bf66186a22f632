import sys, time, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from test_hit_table_host import _records
from grafimo_amd import extract_regions as xr
rng=np.random.default_rng(1)
W=19; NR=50000
pt=np.linspace(1,0,19001)
eo=[np.zeros(NR,np.int64)]; rb=np.array([0,NR],np.int64)
for n,k in ((8800,1),(8800,3),(8800,6),(88000,3)):
    specs=[(pt,62,-14.0,W,eo,rb,[_records(rng,n,W,NR,n_win=150000,dup_scores=False)],False,False) for _ in range(k)]
    s=[];a=[];st=[]
    for _ in range(60):
        t=time.perf_counter(); [xr._hit_columns(*x) for x in specs]; s.append(time.perf_counter()-t)
        t=time.perf_counter(); r=xr._ColumnsRun(specs); st.append(time.perf_counter()-t); r.wait(); a.append(time.perf_counter()-t)
    print(f"{n} rows x {k} jobs: sync {1e6*np.median(s):.0f} us, async start {1e6*np.median(st):.0f} us, start+wait {1e6*np.median(a):.0f} us")
