import sys, time, numpy as np
sys.path.insert(0,'.')
from xpysom_dask_amd import XPySom
from xpysom_dask_amd.synthetic import gaussian_blobs
for (X,Y,D,N,T) in ((64,64,32,100000,10),(128,128,64,200000,10)):
    data=gaussian_blobs(N,D,seed=5)
    res={}
    for prec in ("f32","bf16"):
        s=XPySom(X,Y,D,random_seed=1234,precision=prec)
        t0=time.perf_counter(); s.train(data,T); dt=time.perf_counter()-t0
        res[prec]=(s, s.quantization_error(data[:20000]), s.topographic_error(data[:20000]), dt)
    a=res["f32"][0]; b=res["bf16"][0]
    wa=np.array(a.winner(data[:20000])); 
    b32=XPySom(X,Y,D,random_seed=1234,precision="f32"); b32._weights=b._weights
    agree_same_map=(np.array(b.winner(data[:20000]))==np.array(b32.winner(data[:20000]))).all(axis=1).mean()
    print(f"{X}x{Y}x{D} N={N} T={T}: QE f32 {res['f32'][1]:.5f} bf16 {res['bf16'][1]:.5f} (rel {abs(res['bf16'][1]-res['f32'][1])/res['f32'][1]:.2e});"
          f" TE f32 {res['f32'][2]:.4f} bf16 {res['bf16'][2]:.4f}; bf16-vs-f32 BMU agreement on the bf16-trained map {agree_same_map:.4f};"
          f" codebook rel diff {np.abs(a._weights-b._weights).max()/np.abs(a._weights).max():.3e}; train s f32 {res['f32'][3]:.2f} bf16 {res['bf16'][3]:.2f}")
