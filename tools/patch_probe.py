"""how many candidate groups a row would have if a group were an a x b patch of the map instead of a 1 x 64 strip
(run with SOM_EXACT_PATCH=0: the engine's strip counts calibrate the per-row band)"""
import os, sys, numpy as np, torch
os.environ.setdefault("SOM_TEST_HOOKS", "1")   # (the library reads its developer switches only under this one)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.decays import exponential_decay
from xpysom_dask_amd.synthetic import gaussian_blobs
X = Y = int(os.environ.get("EX_SIDE", "256")); D = int(os.environ.get("EX_D", "128")); N = int(os.environ.get("EX_ROWS", "65536")); T = 10
DIST = os.environ.get("EX_DIST", "euclidean"); NEIGH = os.environ.get("EX_NEIGH", "gaussian")
rs = np.random.RandomState(1234)
w = rs.rand(X, Y, D) * 2 - 1; w /= np.linalg.norm(w, axis=-1, keepdims=True); w = w.astype(np.float32)
data = gaussian_blobs(N, D)
if DIST == "cosine": data = np.abs(data); w = np.abs(w)
tr = HipEngine(X, Y, D, precision="f32", distance=DIST, neighborhood=NEIGH); tr.set_data(data); tr.set_weights(w)
ex = HipEngine(X, Y, D, precision="exact", distance=DIST, neighborhood=NEIGH); ex.set_data(data)
S = 2048
xs = torch.from_numpy(data[:S]).cuda()
for t in range(T + 1):
    if t in (0, 1, 2, 5, 10):
        wt = tr.get_weights(); ex.set_weights(wt)
        sig, eta = exponential_decay(min(X, Y) / 2, 1, min(t, T - 1), T), exponential_decay(0.5, 0.01, min(t, T - 1), T)
        ex.epoch_accumulate(sig, eta, True); ex.sync()
        c = ex.exact_last_counts(S).astype(np.int64)
        W = torch.from_numpy(wt.reshape(-1, D)).cuda()
        xx = xs
        if DIST == "cosine":
            W = W / W.norm(dim=1, keepdim=True); xx = xs / xs.norm(dim=1, keepdim=True)
            d = -(xx.double() @ W.double().T)
        else:
            d = 0.5 * (W.double() ** 2).sum(1)[None, :] - xx.double() @ W.double().T
        d = d.reshape(S, X, Y)
        strips = d.reshape(S, X, Y // 64, 64).amin(-1).reshape(S, -1)
        srt, _ = strips.sort(dim=1)
        ci = torch.from_numpy(np.clip(c, 1, strips.shape[1]) - 1).cuda()
        tau = srt.gather(1, ci[:, None])                      # the band that gives the engine's count on strips
        out = ["state %2d: engine strips mean %.2f" % (t, c.mean())]
        for a, b in ((1, 64), (2, 32), (4, 16), (8, 8), (16, 4), (1, 32), (4, 8), (8, 4), (2, 16)):
            g = d.reshape(S, X // a, a, Y // b, b).amin(dim=(2, 4)).reshape(S, -1)
            n = (g <= tau).sum(1).double()
            out.append("%dx%d: %.2f (units %.0f)" % (a, b, n.mean().item(), n.mean().item() * a * b))
        print("  ".join(out), flush=True)
    if t < T:
        sig, eta = exponential_decay(min(X, Y) / 2, 1, t, T), exponential_decay(0.5, 0.01, t, T)
        tr.epoch(sig, eta, True)
