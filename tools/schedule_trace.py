"""Every epoch of a schedule from a fresh engine, one by one (host sync after each): ms, executed share, and -- SOM_DEBUG=1 --
the library's own plan lines (scout estimate, level-1 / level-2 shares, sorts).
    SOM_DEBUG=1 ST_ROWS=1048576 ST_T=25 python tools/schedule_trace.py [blobs|normal|overlap|manifold|heavy]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpysom_dask_amd.decays import exponential_decay
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd import synthetic

X = Y = int(os.environ.get("ST_SIDE", "256")); D = int(os.environ.get("ST_D", "128"))
N = int(os.environ.get("ST_ROWS", str(1 << 20))); T = int(os.environ.get("ST_T", "25"))
kind = sys.argv[1] if len(sys.argv) > 1 else "blobs"
data = synthetic.variant(kind, N, D) if hasattr(synthetic, "variant") else synthetic.gaussian_blobs(N, D)
rs = np.random.RandomState(1234)
w = rs.rand(X, Y, D) * 2 - 1; w /= np.linalg.norm(w, axis=-1, keepdims=True)
e = HipEngine(X, Y, D, precision=os.environ.get("ST_PREC", "exact")); e.set_weights(w.astype(np.float32)); e.set_data(data); e.sync()
tot = 0.0
for t in range(T):
    sig, eta = exponential_decay(min(X, Y) / 2, 1, t, T), exponential_decay(0.5, 0.01, t, T)
    s0 = e.exact_skip_stats(); t0 = time.perf_counter()
    e.epoch(sig, eta, True); e.sync()
    ms = 1e3 * (time.perf_counter() - t0); s1 = e.exact_skip_stats(); tot += ms
    print("epoch %2d sigma %6.1f: %7.3f ms share %.4f" % (t, sig, ms, (s1[0] - s0[0]) / max(1, s1[1] - s0[1])), flush=True)
print("whole schedule: %.3f ms per epoch; scout stats %s" % (tot / T, e.exact_scout_stats()))
