"""Would a bound that knows the map is a SHEET pay?  The ball bound charges a row's whole distance to a group's centroid against
the group's radius: |x - w| >= |x - c| - r.  Most of that distance is noise ORTHOGONAL to the sheet (the benchmark's rows sit
~10.5 from their BMU, the units of a patch within 2..5 of each other).  With an orthonormal basis B_g of the k leading directions
of the group's offsets w - c (residual radius rho = max |(I - B B^T)(w - c)|):
    |x - w|^2 >= |x - c|^2 + m(a, r) - 2 |x - c| rho,   a = |B^T (x - c)|,  m = -a^2 (a <= r), r^2 - 2 a r (a > r)
Float64 count over the benchmark's schedule at full size: share of (128-row tile, group) pairs kept by the ball bound, by the
projected bound (k = 2, 3), and by both; the same for the 16-unit sub-blocks (what level 2 tests today).
    python tools/proj_probe.py"""
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
T = int(os.environ.get("PP_T", "25"))
KIND = os.environ.get("PP_DATA", "blobs")
EPOCHS = [int(v) for v in os.environ.get("PP_EPOCHS", "3,6,10,14,18,21,24").split(",")]
rs = np.random.RandomState(1234)
w = rs.rand(X, Y, D) * 2 - 1
w /= np.linalg.norm(w, axis=-1, keepdims=True)
data = variant(KIND, N, D, seed=1234)
tr = HipEngine(X, Y, D, precision="exact")
tr.set_data(data)
tr.set_weights(w.astype(np.float32))
xs = torch.from_numpy(data).cuda()
GY = Y // 8


def groups_of(W, shape):
    a, b = shape
    return W.reshape(X // a, a, Y // b, b, D).permute(0, 2, 1, 3, 4).reshape(-1, a * b, D)


prev = None
for t in range(T):
    wt = tr.get_weights()
    sig, eta = exponential_decay(min(X, Y) / 2, 1, t, T), exponential_decay(0.5, 0.01, t, T)
    if t in EPOCHS and prev is not None:
        W = torch.from_numpy(wt.reshape(X, Y, D)).cuda().double()
        Wf = W.reshape(-1, D)
        pv = torch.from_numpy(prev).cuda()
        o = torch.argsort((pv // Y // 8) * GY + (pv % Y) // 8, stable=True)
        out = []
        for shape in ((8, 8), (4, 4)):
            G = groups_of(W, shape)
            c = G.mean(1)
            off = G - c[:, None, :]
            r = off.norm(dim=2).amax(1)
            _, _, Vh = torch.linalg.svd(off, full_matrices=False)       # [G][min(m, D)][D]
            res = {}
            for k in (2, 3):
                B = Vh[:, :k, :]                                         # [G][k][D]
                inp = torch.einsum("gmd,gkd->gmk", off, B)
                rho = (off - torch.einsum("gmk,gkd->gmd", inp, B)).norm(dim=2).amax(1)
                rin = inp.norm(dim=2).amax(1)                            # in-subspace radius
                keep_ball = torch.empty((N, c.shape[0]), dtype=torch.bool, device="cuda")
                keep_proj = torch.empty_like(keep_ball)
                cb = torch.einsum("gd,gkd->gk", c, B)
                for lo in range(0, N, 1 << 15):
                    xd = xs[lo:lo + (1 << 15)].double()
                    U2 = ((xd - Wf[pv[lo:lo + (1 << 15)]]) ** 2).sum(1)[:, None]
                    dc = torch.cdist(xd, c)
                    keep_ball[lo:lo + (1 << 15)] = (dc - r[None, :]) <= U2.sqrt()
                    p = torch.einsum("nd,gkd->ngk", xd, B) - cb[None]
                    a = p.norm(dim=2)
                    m = torch.where(a <= rin[None, :], -a * a, rin[None, :] ** 2 - 2 * a * rin[None, :])
                    keep_proj[lo:lo + (1 << 15)] = (dc * dc + m - 2 * dc * rho[None, :]) <= U2
                def sh(mask):
                    return mask[o].reshape(N // 128, 128, -1).any(1).float().mean().item()
                res["ball"] = sh(keep_ball)
                res["proj%d" % k] = sh(keep_proj)
                res["both%d" % k] = sh(keep_ball & keep_proj)
                res["rho%d/r" % k] = (rho / r).mean().item()
                del keep_ball, keep_proj
            out.append("%dx%d: " % shape + " ".join("%s %.4f" % kv for kv in res.items()))
        print("epoch %2d sigma %6.2f: " % (t, sig) + " | ".join(out), flush=True)
    tr.epoch_accumulate(sig, eta, True)
    prev = tr.epoch_fetch()[2].astype(np.int64)
    tr.epoch_merge()
