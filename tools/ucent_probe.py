"""Would an upper bound of the BMU distance taken from the CURRENT codebook's group centroids -- U_c(x) = min_g (|x - c_g| + r_g)^2,
no last BMU needed -- let the plan skip in a schedule's first epochs, where last epoch's BMU says little (rows still travel
across a smooth map)?  Float64 count over the benchmark's schedule: share of (256-row tile, group) blocks that must run with
U = U_last (today), U = min(U_last, U_c), and U_c in the LINEARISED form the MFMA can evaluate,
U_lin(x) = min_g [ |x - c_g|^2 + r_g (2 Dhat(x) + r_g) ],  Dhat(x) = |x| + max |c|.
    UC_ROWS=262144 python tools/ucent_probe.py
"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpysom_dask_amd.decays import exponential_decay
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.synthetic import gaussian_blobs

X = Y = 256; D = 128
N = int(os.environ.get("UC_ROWS", str(1 << 18))); T = int(os.environ.get("UC_T", "25"))
EPOCHS = [int(v) for v in os.environ.get("UC_EPOCHS", "1,2,3,4,6,10,16,24").split(",")]
rs = np.random.RandomState(1234)
w = rs.rand(X, Y, D) * 2 - 1; w /= np.linalg.norm(w, axis=-1, keepdims=True)
data = gaussian_blobs(N, D, seed=1234, centre_seed=1234)
tr = HipEngine(X, Y, D, precision="exact"); tr.set_data(data); tr.set_weights(w.astype(np.float32))
xs = torch.from_numpy(data).cuda().double()
GY = Y // 8
prev = None


def balls(W, a, b):
    G = W.reshape(X // a, a, Y // b, b, D).permute(0, 2, 1, 3, 4).reshape(-1, a * b, D)
    c = G.mean(1)
    return c, (G - c[:, None, :]).norm(dim=2).amax(1)


for t in range(T):
    wt = tr.get_weights()
    sig, eta = exponential_decay(min(X, Y) / 2, 1, t, T), exponential_decay(0.5, 0.01, t, T)
    if t in EPOCHS and prev is not None:
        W = torch.from_numpy(wt.reshape(X, Y, D)).cuda().double()
        c8, r8 = balls(W, 8, 8)
        c4, r4 = balls(W, 4, 4)                               # [X/4][Y/4] -> group (i4 // 2, j4 // 2)
        pu = torch.from_numpy(prev).cuda()
        patch = (pu // Y // 8) * GY + (pu % Y) // 8
        order = torch.argsort(patch, stable=True)
        xo = xs[order]
        Ulast = (xo - W.reshape(-1, D)[pu[order]]).norm(dim=1)
        dc = torch.cdist(xo, c8)
        Ucent = (dc + r8[None, :]).amin(1)
        Dhat = xo.norm(dim=1) + c8.norm(dim=1).max()
        Ulin = (dc * dc + r8[None, :] * (2 * Dhat[:, None] + r8[None, :])).amin(1).sqrt()
        dc4 = torch.cdist(xo, c4)
        out = []
        for name, U in (("last", Ulast), ("min(last, cent)", torch.minimum(Ulast, Ucent)), ("min(last, lin)", torch.minimum(Ulast, Ulin)), ("cent alone", Ucent)):
            need = (dc - r8[None, :]) <= U[:, None]
            g = need.reshape(N // 256, 256, -1).any(1).double().mean().item()
            need4 = ((dc4 - r4[None, :]) <= U[:, None]).reshape(N, X // 8, 2, GY, 2).permute(0, 1, 3, 2, 4).reshape(N, -1, 4) & need[:, :, None]
            b = need4.reshape(N // 256, 256, -1).any(1).double().mean().item()
            out.append("%s: groups %.4f blocks %.4f" % (name, g, b))
        print("epoch %2d sigma %6.1f: mean sqrt(U) last %.2f cent %.2f lin %.2f, radius mean %.3f | %s" % (
            t, sig, Ulast.mean().item(), Ucent.mean().item(), Ulin.mean().item(), r8.mean().item(), " | ".join(out)), flush=True)
    tr.epoch_accumulate(sig, eta, True)
    prev = tr.epoch_fetch()[2].astype(np.int64)
    tr.epoch_merge()
