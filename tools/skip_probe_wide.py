"""Feasibility probe: block skipping beyond 128 features for the ORDINARY wide case -- euclidean distance + gaussian neighbourhood
at 784 features (MNIST-shaped rows; the G17 family), where tools/skip_probe_c5.py looked at configs[4]'s cosine + mexican_hat only.
Share of (256-row tile, block) pairs a centroid / radius bound cannot prove empty, rows in the order of their last BMU's patch,
for blocks of 64 / 32 / 16 units; float32 torch arithmetic without the margins of a rigorous version (an upper bound of what
could be skipped).  U = distance to last epoch's BMU under the current codebook.
    SP_SIDE=512 SP_ROWS=250000 SP_EPOCHS=12 python3 tools/skip_probe_wide.py
"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.decays import exponential_decay
from xpysom_dask_amd.synthetic import gaussian_blobs

SIDE = int(os.environ.get("SP_SIDE", "512")); D = int(os.environ.get("SP_D", "784")); N = int(os.environ.get("SP_ROWS", "250000"))
T = int(os.environ.get("SP_EPOCHS", "12")); TILE = 256
NONNEG = os.environ.get("SP_NONNEG", "0") == "1"           # |x|, as configs[4]'s rows (MNIST-like)
data = gaussian_blobs(N, D, seed=1234, centre_seed=1234)
if NONNEG:
    data = np.abs(data)
rs = np.random.RandomState(1234)
w = rs.rand(SIDE, SIDE, D) * 2 - 1; w /= np.linalg.norm(w, axis=-1, keepdims=True); w = w.astype(np.float32)
eng = HipEngine(SIDE, SIDE, D, precision="exact", distance="euclidean", neighborhood="gaussian")
eng.set_data(data); eng.set_weights(w)
xs = torch.from_numpy(data).cuda(); x2 = (xs * xs).sum(1)
prev = None
NT = N // TILE
print("map %d x %d x %d, %d rows, euclidean + gaussian, %d-epoch schedule%s" % (SIDE, SIDE, D, N, T, ", non-negative rows" if NONNEG else ""), flush=True)
for t in range(T):
    wt = eng.get_weights()
    sig, eta = exponential_decay(SIDE / 2, 1, t, T), exponential_decay(0.5, 0.01, t, T)
    t0 = time.perf_counter(); eng.epoch_accumulate(sig, eta, True); eng.sync(); ms = (time.perf_counter() - t0) * 1e3
    bmu = eng.epoch_fetch()[2].astype(np.int64)
    if prev is not None:
        W = torch.from_numpy(wt.reshape(SIDE, SIDE, D)).cuda()
        pu = torch.from_numpy(prev).cuda()
        U = (xs - W.reshape(-1, D)[pu]).norm(dim=1)                                  # distance to last epoch's BMU, now
        patch = (pu // SIDE // 8) * (SIDE // 8) + (pu % SIDE) // 8
        order = torch.argsort(patch * (SIDE * SIDE) + pu)[: NT * TILE]
        out = []
        for name, (a, b) in (("64 (8x8)", (8, 8)), ("32 (4x8)", (4, 8)), ("16 (4x4)", (4, 4))):
            G = W.reshape(SIDE // a, a, SIDE // b, b, D).permute(0, 2, 1, 3, 4).reshape(-1, a * b, D)
            c = G.mean(1); r = (G - c[:, None, :]).norm(dim=2).amax(1); c2 = (c * c).sum(1)
            kept = 0; pairs = 0
            for lo in range(0, NT * TILE, 16 * TILE):
                idx = order[lo: lo + 16 * TILE]
                x = xs[idx]
                d2 = (x2[idx][:, None] + c2[None, :] - 2.0 * (x @ c.T)).clamp_min(0)
                can = (d2.sqrt() - r[None, :]) <= U[idx][:, None]
                kept += can.reshape(-1, TILE, can.shape[1]).any(1).sum().item(); pairs += can.sum().item()
            out.append("%s: tiles %.2f %% rows %.3f %% (mean r %.3f)" % (name, 100.0 * kept / (NT * c.shape[0]), 100.0 * pairs / (NT * TILE * c.shape[0]), r.mean().item()))
            del G
        print("epoch %2d (sigma %.1f, %.1f ms): BMU unchanged %.1f %%, mean sqrt(U) %.3f; blocks that must run -- %s" % (
            t, sig, ms, 100 * float((bmu == prev).mean()), U.mean().item(), "; ".join(out)), flush=True)
    prev = bmu
    eng.epoch_merge()
