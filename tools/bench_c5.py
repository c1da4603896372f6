import sys, time, numpy as np
sys.path.insert(0,'.')
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.synthetic import gaussian_blobs
def run(X,Y,D,N,prec,dist="euclidean",neigh="gaussian",epochs=2):
    e=HipEngine(X,Y,D,precision=prec,distance=dist,neighborhood=neigh)
    rs=np.random.RandomState(1234); w=np.abs(rs.rand(X,Y,D)); w/=np.linalg.norm(w,axis=-1,keepdims=True)
    e.set_weights(w.astype(np.float32)); data=np.abs(gaussian_blobs(N,D)); e.set_data(data)
    e.epoch(min(X,Y)/2,0.5,True); e.sync()
    e.profile_reset(); e.profile_enable(True)
    t0=time.perf_counter()
    for i in range(epochs): e.epoch(min(X,Y)/2*0.8**i,0.4,True)
    e.sync(); dt=(time.perf_counter()-t0)/epochs
    e.profile_enable(False)
    parts={k:round(e.profile_get(k)[0]/epochs,3) for k in ("prep","bmu","segsum","kron","merge")}
    fl=2.0*N*X*Y*D
    print(f"{X}x{Y}x{D} N={N} {prec} {dist}/{neigh}: {dt*1e3:.2f} ms/epoch  {N/dt/1e6:.3f} Msamples/s  bmu {fl/(parts['bmu']*1e-3)/1e12:.1f} TF/s  {parts}", flush=True)
run(128,128,784,65536,"bf16")
run(256,256,784,65536,"bf16","cosine","mexican_hat")
run(512,512,784,32768,"bf16","cosine","mexican_hat",epochs=1)
run(128,128,784,16384,"f32",epochs=1)
if "--full" in sys.argv:      # the per-GPU shard of BASELINE configs[4]: 2M rows / 8 GPUs
    run(512,512,784,250000,"bf16","cosine","mexican_hat",epochs=2)
