import numpy as np, sys
sys.path.insert(0,'.')
from oracle import som_oracle as O
from tests.conftest import load_golden
from xpysom_dask_amd.engine import HipEngine
g=load_golden("g6_iris"); z=g["iris_z"].astype(np.float32)
decay="linear"; init="random"
w=g[f"{decay}_{init}_w0"].astype(np.float32)
e=HipEngine(6,6,4)
e.set_data(z)
f=O.DECAYS[decay]
for t in range(100):
    eta=f(0.5,0.01,t,100); sig=f(3.0,1,t,100)
    e.set_weights(w)
    e.epoch_accumulate(sig,eta,False)
    num,den,bmu=e.epoch_fetch()
    obmu,onum,oden,wn=O.epoch(z,w,eta,sig,wide=False,n_parallel=4000)
    bad=np.flatnonzero(bmu!=obmu)
    if len(bad):
        wf=w.reshape(-1,4).astype(np.float64)
        d=((z[bad,None,:].astype(np.float64)-wf[None])**2).sum(-1)
        print("epoch",t,"mismatch",len(bad),[(int(b),int(bmu[b]),int(obmu[b]),d[i,bmu[b]],d[i,obmu[b]]) for i,b in enumerate(bad[:5])])
    e.epoch_merge()
    wg=e.get_weights().reshape(6,6,4)
    err=np.abs(wg-wn).max()
    if err>1e-5 or len(bad): print("epoch",t,"werr",err)
    w=wn
print("done")
