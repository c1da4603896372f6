import sys, time, numpy as np
sys.path.insert(0,'.')
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.synthetic import gaussian_blobs
def run(X,Y,D,N,prec,epochs=3):
    e=HipEngine(X,Y,D,precision=prec)
    rs=np.random.RandomState(1234); w=rs.rand(X,Y,D)*2-1; w/=np.linalg.norm(w,axis=-1,keepdims=True)
    e.set_weights(w.astype(np.float32)); e.set_data(gaussian_blobs(N,D))
    e.epoch(min(X,Y)/2,0.5,True); e.sync()
    e.profile_reset(); e.profile_enable(True)
    t0=time.perf_counter()
    for i in range(epochs): e.epoch(min(X,Y)/2*0.8**i,0.4,True)
    e.sync(); dt=(time.perf_counter()-t0)/epochs
    e.profile_enable(False)
    parts={k:round(e.profile_get(k)[0]/epochs,3) for k in ("prep","bmu","segsum","kron","merge")}
    fl=2.0*N*X*Y*D
    print(f"{X}x{Y}x{D} N={N} {prec}: {dt*1e3:.3f} ms/epoch  {N/dt/1e6:.2f} Msamples/s  bmu {fl/(parts['bmu']*1e-3)/1e12:.1f} TF/s  {parts}")
run(64,64,32,100000,"f32"); run(64,64,32,100000,"bf16")
run(256,256,128,131072,"f32")
run(6,6,4,150,"f32",epochs=20)
t0=time.perf_counter()
from xpysom_dask_amd import XPySom
from tests.conftest import load_golden
z=load_golden("g6_iris")["iris_z"]
s=XPySom(6,6,4,random_seed=10); s.train(z,100)
t0=time.perf_counter(); s=XPySom(6,6,4,random_seed=10); s.train(z,100); print("iris 100 epochs (C1) wall ms:", (time.perf_counter()-t0)*1e3)
