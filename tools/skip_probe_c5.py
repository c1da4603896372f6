"""Feasibility probe: block skipping for the configs[4] shard (512 x 512 x 784, cosine): the share of (256-row tile, block)
pairs a centroid / radius bound on the unit-length operands cannot prove empty, rows visited in the order of their last BMU's
patch, for blocks of 64 units (an 8 x 8 patch = a group of the wide screen) and of 32 units (one of its stages, 4 x 8).
Float32 torch arithmetic without the margins of a rigorous version: an upper bound of what could be skipped.

    python3 tools/skip_probe_c5.py            (SP_ROWS, SP_EPOCHS, SP_SIDE, SP_D)
"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.decays import exponential_decay

SIDE = int(os.environ.get("SP_SIDE", "512")); D = int(os.environ.get("SP_D", "784")); N = int(os.environ.get("SP_ROWS", "250000"))
T = int(os.environ.get("SP_EPOCHS", "12")); TILE = 256
B.WORKLOADS["c5"]["features"] = D
data = B.workload_rows("c5", N, 1234)
rs = np.random.RandomState(1234)
w = rs.rand(SIDE, SIDE, D) * 2 - 1; w /= np.linalg.norm(w, axis=-1, keepdims=True); w = np.abs(w).astype(np.float32)
eng = HipEngine(SIDE, SIDE, D, precision="exact", distance="cosine", neighborhood="mexican_hat")
eng.set_data(data); eng.set_weights(w)
xs = torch.from_numpy(data).cuda(); xs = xs / xs.norm(dim=1, keepdim=True)
prev = None
NT = N // TILE
for t in range(T):
    wt = eng.get_weights()
    sig, eta = exponential_decay(SIDE / 2, 1, t, T), exponential_decay(0.5, 0.01, t, T)
    t0 = time.perf_counter(); eng.epoch_accumulate(sig, eta, True); eng.sync(); ms = (time.perf_counter() - t0) * 1e3
    bmu = eng.epoch_fetch()[2].astype(np.int64)
    if prev is not None:
        W = torch.from_numpy(wt.reshape(SIDE, SIDE, D)).cuda()
        W = W / W.norm(dim=-1, keepdim=True).clamp_min(1e-30)
        pu = torch.from_numpy(prev).cuda()
        U = (xs - W.reshape(-1, D)[pu]).norm(dim=1)                                  # distance to last epoch's BMU, now
        patch = (pu // SIDE // 8) * (SIDE // 8) + (pu % SIDE) // 8
        order = torch.argsort(patch * (SIDE * SIDE) + pu)[: NT * TILE]
        out = []
        for name, (a, b) in (("64 (8x8)", (8, 8)), ("32 (4x8)", (4, 8)), ("16 (2x8)", (2, 8))):
            G = W.reshape(SIDE // a, a, SIDE // b, b, D).permute(0, 2, 1, 3, 4).reshape(-1, a * b, D)
            c = G.mean(1); r = (G - c[:, None, :]).norm(dim=2).amax(1); c2 = (c * c).sum(1)
            kept = 0; pairs = 0
            for lo in range(0, NT * TILE, 16 * TILE):
                idx = order[lo: lo + 16 * TILE]
                x = xs[idx]
                d2 = (1.0 + c2[None, :] - 2.0 * (x @ c.T)).clamp_min(0)
                can = (d2.sqrt() - r[None, :]) <= U[idx][:, None]
                kept += can.reshape(-1, TILE, can.shape[1]).any(1).sum().item(); pairs += can.sum().item()
            out.append("%s: tiles %.2f %% rows %.3f %% (mean r %.3f)" % (name, 100.0 * kept / (NT * c.shape[0]), 100.0 * pairs / (NT * TILE * c.shape[0]), r.mean().item()))
            del G
        print("epoch %2d (sigma %.1f, %.1f ms): BMU unchanged %.1f %%, mean sqrt(U) %.3f; blocks that must run -- %s" % (
            t, sig, ms, 100 * float((bmu == prev).mean()), U.mean().item(), "; ".join(out)), flush=True)
    prev = bmu
    eng.epoch_merge()
