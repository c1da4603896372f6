import sys, numpy as np
sys.path.insert(0, '.')
from xpysom_dask_amd.engine import HipEngine
e = HipEngine(4, 4, 4, precision="f32")
rs = np.random.RandomState(0)
worst = {}
for trial in range(3000):
    kind = trial % 6
    a = (rs.randn(16, 32) * 2.0 ** rs.randint(-6, 12)).astype(np.float16)
    b = (rs.randn(32, 16) * 2.0 ** rs.randint(-6, 12)).astype(np.float16)
    if kind == 1: a[:, 2:] = 0; 
    if kind == 2: a = np.abs(a); b = np.abs(b)
    cs = [0.0, 1.0, 2.0 ** 10, 2.0 ** 20, 2.0 ** 30, 2.0 ** 37][rs.randint(0, 6)]
    c = (rs.rand(16, 16).astype(np.float32) + 0.5) * np.float32(cs) * (1 if kind != 3 else -1)
    d = e.debug_mfma16(a, b, c)
    ex = a.astype(np.float64) @ b.astype(np.float64) + c.astype(np.float64)
    mag = np.maximum(np.maximum(np.abs(c.astype(np.float64)), np.abs(ex)), np.abs(a.astype(np.float64)) @ np.abs(b.astype(np.float64)))
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(mag, 1e-300))) - 23)
    err = np.abs(d.astype(np.float64) - ex) / ulp
    key = (kind, cs)
    worst[key] = max(worst.get(key, 0), err.max())
for k in sorted(worst): print(k, "max err %.3f ulp of the largest magnitude" % worst[k])
print("overall", max(worst.values()))
