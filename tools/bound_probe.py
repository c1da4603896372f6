"""What limits block skipping late in the schedule, and what would a finer bound or a stale row order cost?

Float64 count (torch on the GPU) over the benchmark's schedule at full size, per epoch:
  pairs        share of (row, 64-unit group) pairs the centroid/radius bound cannot rule out (the per-row need)
  tiles<T>     share of (T-row tile, group) blocks that must run, rows sorted by (last BMU's patch, last BMU)
  patch-only   the same with rows sorted by the patch only (what csrc/exact_skip.hpp sorts by)
  stale<k>     ... rows sorted by the BMUs of k epochs ago (a resident sorted pass that is not re-sorted every epoch)
  sub2x8 / sub4x4   a group is needed only if one of its four 16-unit sub-balls is (2 x 8 strips = the stage's t16 tiles in
               today's order; 4 x 4 blocks); `t16` = the share of (tile, 16-unit sub-block) blocks needed
    BP_T=25 BP_ROWS=1048576 python tools/bound_probe.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpysom_dask_amd.decays import exponential_decay
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.synthetic import gaussian_blobs

X = Y = 256
D = 128
N = int(os.environ.get("BP_ROWS", str(1 << 20)))
T = int(os.environ.get("BP_T", "25"))
EPOCHS = [int(v) for v in os.environ.get("BP_EPOCHS", "3,6,10,14,17,19,21,23,24").split(",")]
rs = np.random.RandomState(1234)
w = rs.rand(X, Y, D) * 2 - 1
w /= np.linalg.norm(w, axis=-1, keepdims=True)
data = gaussian_blobs(N, D, seed=1234, centre_seed=1234)
tr = HipEngine(X, Y, D, precision="exact")
tr.set_data(data)
tr.set_weights(w.astype(np.float32))
xs = torch.from_numpy(data).cuda()
GY = Y // 8


def patch_of(u):
    return (u // Y // 8) * GY + (u % Y) // 8


def balls(W, shape):
    """centroid / radius of every sub-block of `shape` = (a, b) units of the map, as [X/a * Y/b] in row-major block order"""
    a, b = shape
    G = W.reshape(X // a, a, Y // b, b, D).permute(0, 2, 1, 3, 4).reshape(-1, a * b, D)
    c = G.mean(1)
    r = (G - c[:, None, :]).norm(dim=2).amax(1)
    return c, r


hist = []
for t in range(T):
    wt = tr.get_weights()
    sig, eta = exponential_decay(min(X, Y) / 2, 1, t, T), exponential_decay(0.5, 0.01, t, T)
    if t in EPOCHS and len(hist) >= 1:
        W = torch.from_numpy(wt.reshape(X, Y, D)).cuda().double()
        Wf = W.reshape(-1, D)
        c8, r8 = balls(W, (8, 8))
        c28, r28 = balls(W, (2, 8))         # [X/2][Y/8]: sub-block (i2, gy) belongs to patch (i2 // 4, gy)
        c44, r44 = balls(W, (4, 4))         # [X/4][Y/4]: sub-block (i4, j4) belongs to patch (i4 // 2, j4 // 2)
        prev = torch.from_numpy(hist[-1]).cuda()
        res = {}
        chunk = 1 << 16
        need8 = torch.empty((N, c8.shape[0]), dtype=torch.bool, device="cuda")
        need28 = torch.empty((N, c28.shape[0]), dtype=torch.bool, device="cuda")
        need44 = torch.empty((N, c44.shape[0]), dtype=torch.bool, device="cuda")
        for lo in range(0, N, chunk):
            xd = xs[lo:lo + chunk].double()
            U = (xd - Wf[prev[lo:lo + chunk]]).norm(dim=1)[:, None]
            need8[lo:lo + chunk] = (torch.cdist(xd, c8) - r8[None, :]) <= U
            need28[lo:lo + chunk] = (torch.cdist(xd, c28) - r28[None, :]) <= U
            need44[lo:lo + chunk] = (torch.cdist(xd, c44) - r44[None, :]) <= U
        # sub-blocks -> their group: [N][groups][4]
        n28 = need28.reshape(N, X // 8, 4, GY).permute(0, 1, 3, 2).reshape(N, -1, 4)
        n44 = need44.reshape(N, X // 8, 2, GY, 2).permute(0, 1, 3, 2, 4).reshape(N, -1, 4)
        g28, g44 = n28.any(2), n44.any(2)
        res["pairs"] = need8.float().mean().item()
        res["pairs_sub2x8"] = (need8 & g28).float().mean().item()
        res["pairs_sub4x4"] = (need8 & g44).float().mean().item()

        def tiles(mask, order, tile):
            return mask[order].reshape(N // tile, tile, -1).any(1).float().mean().item()

        key = patch_of(prev) * (X * Y) + prev
        o_full = torch.argsort(key, stable=True)
        o_patch = torch.argsort(patch_of(prev), stable=True)
        for tile in (256, 128, 64):
            res["tiles%d" % tile] = tiles(need8, o_full, tile)
        res["patch-only256"] = tiles(need8, o_patch, 256)
        res["sub2x8_256"] = tiles(need8 & g28, o_full, 256)
        res["sub4x4_256"] = tiles(need8 & g44, o_full, 256)
        res["t16_2x8_256"] = (n28 & need8[:, :, None])[o_full].reshape(N // 256, 256, -1).any(1).float().mean().item()
        res["t16_4x4_256"] = (n44 & need8[:, :, None])[o_full].reshape(N // 256, 256, -1).any(1).float().mean().item()
        for k in (2, 3, 4):
            if len(hist) >= k:
                old = torch.from_numpy(hist[-k]).cuda()
                res["stale%d_256" % (k - 1)] = tiles(need8, torch.argsort(patch_of(old) * (X * Y) + old, stable=True), 256)
        res["radius_mean"] = r8.mean().item()
        res["radius_p90"] = r8.quantile(0.9).item()
        print("epoch %2d sigma %6.2f: " % (t, sig) + "  ".join("%s %.4f" % kv for kv in res.items()), flush=True)
        del need8, need28, need44, n28, n44, g28, g44
    tr.epoch_accumulate(sig, eta, True)
    hist.append(tr.epoch_fetch()[2].astype(np.int64))
    hist = hist[-4:]
    tr.epoch_merge()
