"""Would a TILE-level gate in front of level 1 of the exact mode's plan pay?  For the benchmark's schedule, per epoch: the share
of (256-row tile, group) pairs that survive  |mu_T - c_g| <= rho_T + max_x sqrt(U(x)) + r_g  (mu_T, rho_T: mean and radius of the
tile's rows; rows in the order of their last BMU's patch), the share of 16-group MFMA tiles of level 1's centroid image that hold a
survivor, and the per-row level-1 share for comparison.  Float32 torch arithmetic without margins.
    TG_T=25 TG_ROWS=1048576 python tools/tile_gate_probe.py
"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpysom_dask_amd.decays import exponential_decay
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.synthetic import gaussian_blobs

X = Y = 256; D = 128
N = int(os.environ.get("TG_ROWS", str(1 << 20))); T = int(os.environ.get("TG_T", "25"))
EPOCHS = [int(v) for v in os.environ.get("TG_EPOCHS", "4,8,12,16,20,24").split(",")]
rs = np.random.RandomState(1234)
w = rs.rand(X, Y, D) * 2 - 1; w /= np.linalg.norm(w, axis=-1, keepdims=True)
data = gaussian_blobs(N, D, seed=1234, centre_seed=1234)
tr = HipEngine(X, Y, D, precision="exact"); tr.set_data(data); tr.set_weights(w.astype(np.float32))
xs = torch.from_numpy(data).cuda()
GY = Y // 8
prev = None
for t in range(T):
    wt = tr.get_weights()
    sig, eta = exponential_decay(min(X, Y) / 2, 1, t, T), exponential_decay(0.5, 0.01, t, T)
    if t in EPOCHS and prev is not None:
        W = torch.from_numpy(wt.reshape(X, Y, D)).cuda()
        G = W.reshape(X // 8, 8, GY, 8, D).permute(0, 2, 1, 3, 4).reshape(-1, 64, D)       # groups in band-major order
        c = G.mean(1); r = (G - c[:, None, :]).norm(dim=2).amax(1); c2 = (c * c).sum(1)
        pu = torch.from_numpy(prev).cuda()
        U = (xs - W.reshape(-1, D)[pu]).norm(dim=1)
        patch = (pu // Y // 8) * GY + (pu % Y) // 8
        order = torch.argsort(patch, stable=True)
        xo = xs[order].reshape(N // 256, 256, D); Uo = U[order].reshape(N // 256, 256)
        mu = xo.mean(1); rho = (xo - mu[:, None, :]).norm(dim=2).amax(1); umax = Uo.amax(1)
        d = torch.cdist(mu, c)
        keep = d <= (rho + umax)[:, None] + r[None, :]
        tiles16 = keep.reshape(N // 256, -1, 16).any(2)
        rows = 0
        for lo in range(0, N // 256, 256):
            xx = xo[lo:lo + 256].reshape(-1, D); uu = Uo[lo:lo + 256].reshape(-1)
            need = (torch.cdist(xx, c) - r[None, :]) <= uu[:, None]
            rows += need.reshape(-1, 256, need.shape[1]).any(1).sum().item()
        print("epoch %2d sigma %6.2f: gate keeps %.3f of the (tile, group) pairs, %.3f of level 1's 16-group MFMA tiles; per-row level 1 keeps %.4f; "
              "rho mean %.1f max %.1f, max sqrt(U) mean %.1f, radius mean %.1f" % (
                  t, sig, keep.float().mean().item(), tiles16.float().mean().item(), rows / (N // 256 * c.shape[0]),
                  rho.mean().item(), rho.max().item(), umax.mean().item(), r.mean().item()), flush=True)
    tr.epoch_accumulate(sig, eta, True)
    prev = tr.epoch_fetch()[2].astype(np.int64)
    tr.epoch_merge()
