"""Feasibility probe for a NEXT step (not built): how many (256-row workgroup tile, 64-unit group) blocks of the exact
mode's screen could be skipped outright if the resident rows were kept sorted by their last BMU's patch and every
group carried a centroid c_g and a radius r_g = max |w - c_g|:  |x - w| >= |x - c_g| - r_g for every unit of the group, so
a block whose rows ALL have (|x - c_g| - r_g)^2 > |x - w_prev(x)|^2 (the distance to last epoch's BMU under the current
codebook: an upper bound of the row's best distance) holds no row's BMU.  Real arithmetic in float64 here: the count
is an upper bound of what a rigorous float32/half version with margins could skip."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.decays import exponential_decay
from xpysom_dask_amd.synthetic import gaussian_blobs
X = Y = int(os.environ.get("EX_SIDE", "256")); D = int(os.environ.get("EX_D", "128")); N = int(os.environ.get("EX_ROWS", "65536")); T = 10
TILE = int(os.environ.get("SP_TILE", "256"))
rs = np.random.RandomState(1234)
w = rs.rand(X, Y, D) * 2 - 1; w /= np.linalg.norm(w, axis=-1, keepdims=True); w = w.astype(np.float32)
data = gaussian_blobs(N, D)
tr = HipEngine(X, Y, D, precision="f32"); tr.set_data(data); tr.set_weights(w)
xs = torch.from_numpy(data).cuda().double()
prev = None
for t in range(T + 1):
    wt = tr.get_weights()
    sig, eta = exponential_decay(min(X, Y) / 2, 1, min(t, T - 1), T), exponential_decay(0.5, 0.01, min(t, T - 1), T)
    tr.epoch_accumulate(sig, eta, True); bmu = tr.epoch_fetch()[2].astype(np.int64)
    if prev is not None:
        W = torch.from_numpy(wt.reshape(X, Y, D)).cuda().double()
        G = W.reshape(X // 8, 8, Y // 8, 8, D).permute(0, 2, 1, 3, 4).reshape(-1, 64, D)      # 8 x 8 patches
        c = G.mean(1); r = (G - c[:, None, :]).norm(dim=2).amax(1)                           # centroid, radius
        pu = torch.from_numpy(prev).cuda()
        U = (xs - W.reshape(-1, D)[pu]).norm(dim=1)                                          # distance to last epoch's BMU, now
        patch = (pu // Y // 8) * (Y // 8) + (pu % Y) // 8
        order = torch.argsort(patch * (X * Y) + pu)                                          # rows sorted by last BMU's patch
        dc = torch.cdist(xs[order], c)                                                       # |x - c_g|
        can = (dc - r[None, :]) <= U[order][:, None]                                         # group may hold the row's BMU
        keep_sorted = can.reshape(N // TILE, TILE, -1).any(1).double().mean().item()
        keep_rows = can.double().mean().item()
        inv = torch.argsort(order)
        keep_unsorted = can[inv].reshape(N // TILE, TILE, -1).any(1).double().mean().item()
        same = float((bmu == prev).mean())
        print("state %2d (sigma %.1f): BMU unchanged on %.1f %% of rows; blocks that must run: rows sorted %.1f %%, unsorted %.1f %%; (row, group) pairs %.2f %%" % (
            t, sig, 100 * same, 100 * keep_sorted, 100 * keep_unsorted, 100 * keep_rows), flush=True)
    prev = bmu
    if t < T:
        tr.epoch_merge()
