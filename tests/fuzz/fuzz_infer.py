"""Inference fuzz: winner / quantization / quantization_error / topographic_error / activate / distance_map of
XPySom (all three precisions) against the oracle on random maps, data and distances."""
import os, sys, time, warnings, numpy as np
sys.path.insert(0, '.')
from oracle import som_oracle as O
from xpysom_dask_amd import XPySom

warnings.filterwarnings("ignore")
F32 = np.float32
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
t0 = time.time()
for case in range(n_cases):
    side = int(os.environ.get("FUZZ_MAXSIDE", "30"))        # > 64 reaches the wide bf16 kernel (>= 4096 units, 130 / 300 features)
    X, Y = int(rs.randint(2, side)), int(rs.randint(2, side))
    D = int(rs.choice([1, 2, 5, 16, 33, 64, 128, 130, 300]))
    n = int(rs.choice([1, 2, 17, 256, 1000, 3001]))
    prec = str(rs.choice(["f32", "f32", "exact", "bf16", "f16", "exact"]))
    dist = str(rs.choice(["euclidean", "cosine"])) if prec not in ("f32", "exact") else str(rs.choice(["euclidean", "cosine", "euclidean_no_opt", "manhattan"]))
    data = O.gaussian_blobs(n, D, seed=case + 5)
    if dist == "cosine":
        data = np.abs(data)
    msgs = []
    try:
        som = XPySom(X, Y, D, activation_distance=dist, random_seed=case, precision=prec, n_parallel=int(rs.choice([0, 7, 500])))
        w = (rs.rand(X, Y, D).astype(F32) * 4 - (0 if dist == "cosine" else 2))
        som._weights = w
        wf = w.reshape(-1, D)
        x64, w64 = data.astype(np.float64), wf.astype(np.float64)
        tol = {"f32": 2.0 ** -18, "exact": 2.0 ** -18, "bf16": 2.0 ** -6, "f16": 2.0 ** -9}[prec]
        # winner: configured distance
        ids = np.array([i * Y + j for i, j in som.winner(data)])
        if dist == "cosine":
            with np.errstate(all="ignore"):
                dd = 1 - np.nan_to_num((x64 @ w64.T) / np.sqrt((x64 ** 2).sum(1)[:, None] * (w64 ** 2).sum(1)[None, :]))
            scale = np.ones(n)
        elif dist == "manhattan":
            dd = np.abs(x64[:, None, :] - w64[None, :, :]).sum(-1); scale = dd.max(1) + 1e-30
        else:
            dd = (x64 ** 2).sum(1)[:, None] - 2 * x64 @ w64.T + (w64 ** 2).sum(1)[None, :]
            scale = ((x64 ** 2).sum(1) + (w64 ** 2).sum(1).max())
        if not (dd[np.arange(n), ids] <= dd.min(1) + tol * scale).all(): msgs.append("winner not near-best")
        one = som.winner(data[0])
        if prec in ("f32", "exact"):
            if one != (ids[0] // Y, ids[0] % Y): msgs.append("winner(1-D) != winner(2-D)[0]")
        elif not dd[0, one[0] * Y + one[1]] <= dd[0].min() + tol * scale[0]:   # bf16: the offset B belongs to the launch
            msgs.append("winner(1-D) not near-best")
        # quantization error (always Euclidean): exact distance to the chosen unit, BMU near-best
        qe, oqe = som.quantization_error(data), O.quantization_error(data, w)
        # bf16 modes: the pick is near-best in d^2 to eps |x||w|; a dense codebook (1 feature, hundreds of units) turns
        # that into a visible relative change of the tiny distances themselves
        qtol = 1e-5 if prec in ("f32", "exact") else 5e-2
        if (prec in ("f32", "exact") or n >= 17) and abs(qe - oqe) > qtol * max(oqe, 1e-6): msgs.append("QE %.7f vs %.7f" % (qe, oqe))
        q = som.quantization(data)
        if q.shape != data.shape or not np.isfinite(q).all(): msgs.append("quantization shape/finite")
        if abs(np.linalg.norm(data.astype(np.float64) - q, axis=1).mean() - qe) > 1e-5 * max(qe, 1e-6): msgs.append("quantization != QE")
        # topographic error (rectangular): top-2 by value, ties may swap ids -> compare the rate loosely
        if n > 1:
            te, ote = som.topographic_error(data), O.topographic_error(data, w)
            if abs(te - ote) > 0.02 + 2.0 / n: msgs.append("TE %.4f vs %.4f" % (te, ote))
        # analysis calls
        if dist in ("euclidean", "cosine", "euclidean_no_opt") and n * X * Y < 2e6:
            a = som.activate(data[:5])
            ref = {"euclidean": O.dist_euclid_part, "euclidean_no_opt": O.dist_euclid_sq, "cosine": O.dist_cosine}[dist](data[:5], wf)
            if np.abs(np.asarray(a).reshape(len(data[:5]), -1) - ref).max() > 1e-4 * max(np.abs(ref).max(), 1.0): msgs.append("activate")
        dm = som.distance_map()
        if dm.shape != (X, Y) or not np.isfinite(dm).all() or dm.max() > 1.0 + 1e-6: msgs.append("distance_map")
    except Exception as ex:                      # noqa: BLE001
        msgs.append("EXC " + repr(ex)[:300])
    if msgs:
        bad += 1
        print(f"FAIL case {case}: {X}x{Y}x{D} n={n} {prec} {dist}: {'; '.join(msgs)}", flush=True)
print(f"{n_cases} cases, {bad} failures, {time.time()-t0:.1f} s")
