"""Do reduced-precision BMUs change the map?  Train the same data / seed in f32, f16 and bf16 and compare
quantization error, topographic error, and the BMU agreement of each mode with f32 on ITS OWN trained map."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from xpysom_dask_amd import XPySom
from xpysom_dask_amd.synthetic import gaussian_blobs
for (X, Y, D, N, T) in ((64, 64, 32, 100000, 10), (128, 128, 64, 200000, 10)):
    data = gaussian_blobs(N, D, seed=5)
    q = data[:20000]
    res = {}
    for prec in ("f32", "exact", "f16", "bf16"):
        s = XPySom(X, Y, D, random_seed=1234, precision=prec)
        t0 = time.perf_counter(); s.train(data, T); dt = time.perf_counter() - t0
        chk = XPySom(X, Y, D, random_seed=1234, precision="f32"); chk._weights = s._weights
        agree = (np.array(s.winner(q)) == np.array(chk.winner(q))).all(axis=1).mean()
        res[prec] = (s, s.quantization_error(q), s.topographic_error(q), dt, agree)
    a = res["f32"][0]
    for prec in ("exact", "f16", "bf16"):
        b, qe, te, dt, agree = res[prec]
        print(f"{X}x{Y}x{D} N={N} T={T} {prec:7s}: QE {qe:.5f} vs f32 {res['f32'][1]:.5f} (rel {abs(qe-res['f32'][1])/res['f32'][1]:.2e}); "
              f"TE {te:.4f} vs {res['f32'][2]:.4f}; BMU agreement with f32 on its own map {agree:.5f}; "
              f"codebook rel diff to the f32-trained map {np.abs(a._weights-b._weights).max()/np.abs(a._weights).max():.3e}; "
              f"train s {dt:.2f} vs {res['f32'][3]:.2f}", flush=True)
