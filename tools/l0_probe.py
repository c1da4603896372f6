"""Would a level above the groups pay?  Float64 count (torch on the GPU) over the benchmark's schedule at full size: per epoch,
for super-blocks of a x b units (32 x 32 = 4 x 4 patches: one 16-slot MFMA tile of a level-1 image ordered block by block;
8 x 128 = today's tile, half a band of patches; 64 x 64 / 16 x 256 = a stage of 64 groups in either order), the share of
(128-row tile, super-block) pairs the centroid / radius bound cannot rule out, rows sorted by their last BMU's patch.
    L0_T=25 python tools/l0_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpysom_dask_amd.decays import exponential_decay
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.synthetic import variant

X = Y = 256
D = 128
N = 1 << 20
T = int(os.environ.get("L0_T", "25"))
KIND = os.environ.get("L0_DATA", "blobs")
EPOCHS = [int(v) for v in os.environ.get("L0_EPOCHS", "3,6,10,14,18,21,24").split(",")]
rs = np.random.RandomState(1234)
w = rs.rand(X, Y, D) * 2 - 1
w /= np.linalg.norm(w, axis=-1, keepdims=True)
data = variant(KIND, N, D, seed=1234)
tr = HipEngine(X, Y, D, precision="exact")
tr.set_data(data)
tr.set_weights(w.astype(np.float32))
xs = torch.from_numpy(data).cuda()
GY = Y // 8


def balls(W, shape):
    a, b = shape
    G = W.reshape(X // a, a, Y // b, b, D).permute(0, 2, 1, 3, 4).reshape(-1, a * b, D)
    c = G.mean(1)
    r = (G - c[:, None, :]).norm(dim=2).amax(1)
    return c, r


prev = None
for t in range(T):
    wt = tr.get_weights()
    sig, eta = exponential_decay(min(X, Y) / 2, 1, t, T), exponential_decay(0.5, 0.01, t, T)
    if t in EPOCHS and prev is not None:
        W = torch.from_numpy(wt.reshape(X, Y, D)).cuda().double()
        Wf = W.reshape(-1, D)
        pv = torch.from_numpy(prev).cuda()
        shapes = [(8, 8), (32, 32), (8, 128), (64, 64), (16, 256), (16, 16)]
        bl = {s: balls(W, s) for s in shapes}
        need = {s: torch.empty((N, bl[s][0].shape[0]), dtype=torch.bool, device="cuda") for s in shapes}
        chunk = 1 << 16
        for lo in range(0, N, chunk):
            xd = xs[lo:lo + chunk].double()
            U = (xd - Wf[pv[lo:lo + chunk]]).norm(dim=1)[:, None]
            for s in shapes:
                c, r = bl[s]
                need[s][lo:lo + chunk] = (torch.cdist(xd, c) - r[None, :]) <= U
        # rows in the order of their last BMU's patch, patches block by block (4 x 4 patches, then 2 x 2 of those)
        px, py = pv // Y // 8, (pv % Y) // 8
        key_rowmajor = px * GY + py
        key_block = ((px // 8) * (GY // 8) + py // 8) * 64 + (((px // 4) & 1) * 2 + ((py // 4) & 1)) * 16 + (px & 3) * 4 + (py & 3)
        out = []
        for name, key in (("row-major", key_rowmajor), ("blocked", key_block)):
            o = torch.argsort(key, stable=True)
            for s in shapes:
                sh = need[s][o].reshape(N // 128, 128, -1).any(1).float().mean().item()
                out.append("%s %dx%d %.4f" % (name, s[0], s[1], sh))
        # groups that survive when their 32 x 32 block must be needed too (the finer bound also holds: an AND of the two)
        o = torch.argsort(key_block, stable=True)
        n8 = need[(8, 8)].reshape(N, X // 8, GY)
        n32 = need[(32, 32)].reshape(N, X // 32, 1, Y // 32, 1).expand(N, X // 32, 4, Y // 32, 4).reshape(N, X // 8, GY)
        both = (n8 & n32).reshape(N, -1)
        out.append("groups&L0 %.4f" % both[o].reshape(N // 128, 128, -1).any(1).float().mean().item())
        print("epoch %2d sigma %6.2f: " % (t, sig) + "  ".join(out), flush=True)
        del need, n8, n32, both
    tr.epoch_accumulate(sig, eta, True)
    prev = tr.epoch_fetch()[2].astype(np.int64)
    tr.epoch_merge()
