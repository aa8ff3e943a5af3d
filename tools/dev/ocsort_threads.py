import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from clearcam_amd.ocsort import OCSort
rng=np.random.default_rng(0)
N=64; D=262
base=[]
for c in range(N):
    xy=rng.uniform(0,1800,(D,2)); wh=rng.uniform(30,300,(D,2))
    d=np.zeros((300,6),np.float32); d[:D,:2]=xy; d[:D,2:4]=xy+wh; d[:D,4]=np.sort(rng.uniform(0.26,0.9,D))[::-1]; d[:D,5]=rng.integers(0,80,D)
    base.append(d)
base=np.stack(base)
for nt in (1,4,8,16,32,64):
    trk=[OCSort(max_age=100) for _ in range(N)]
    ts=[]
    for f in range(12):
        b=base.copy(); b[:,:D,:4]+=rng.normal(0,1.0,(N,D,4)).astype(np.float32)
        t0=time.perf_counter(); rows=OCSort.update_many(trk,b,0.25,n_threads=nt); ts.append(time.perf_counter()-t0)
    print("threads",nt,"median ms %.2f" % (sorted(ts)[6]*1e3))
