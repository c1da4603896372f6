"""Class-level fuzz: XPySom.train (f32 precision) against the oracle's train() on random small configurations --
map shape, features, rows, decay function, neighbourhood, topology, GEMM-form distance, sigma, learning rate,
one epoch at a random point of the schedule (iter_beg / iter_end).  The merged codebook must agree to 2e-4
relative on every unit whose denominator is clear of the underflow / cancellation zone."""
import sys, time, warnings, numpy as np
sys.path.insert(0, '.')
from oracle import som_oracle as O
from xpysom_dask_amd import XPySom

warnings.filterwarnings("ignore")
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
t0 = time.time()
for case in range(n_cases):
    X, Y = int(rs.randint(2, 25)), int(rs.randint(2, 25))
    D = int(rs.choice([1, 2, 3, 8, 17, 32, 64, 129, 200]))
    n = int(rs.choice([5, 64, 257, 1000, 2500]))
    decay = str(rs.choice(["linear", "exponential", "asymptotic"]))
    neigh = str(rs.choice(["gaussian", "mexican_hat", "bubble", "triangle"]))
    topo = "rectangular" if neigh == "triangle" else str(rs.choice(["rectangular", "hexagonal"]))
    dist = str(rs.choice(["euclidean", "euclidean", "cosine", "euclidean_no_opt"]))
    sigma = float(rs.choice([0, 1.5, 3.0])) or min(X, Y) / 2
    lr = float(rs.choice([0.5, 0.1, 1.0]))
    compact = bool(neigh in ("gaussian", "triangle") and rs.rand() < 0.4)      # (hexagonal gaussian: the four-class mask)
    if neigh == "mexican_hat" and rs.rand() < 0.4:            # the reference's double mask on px: hexagonal, or square maps
        compact = True
        if topo == "rectangular":
            Y = X
    std_coeff = float(rs.choice([0.5, 0.5, 0.25, 1.0]))
    T = int(rs.choice([1, 3, 10]))                 # schedule length; ONE epoch of it is run and compared
    t_at = int(rs.randint(0, T))
    data = O.gaussian_blobs(n, D, seed=case + 1000)
    if dist == "cosine":
        data = np.abs(data)
    try:
        som = XPySom(X, Y, D, sigma=sigma, learning_rate=lr, decay_function=decay, neighborhood_function=neigh,
                     topology=topo, activation_distance=dist, random_seed=case, compact_support=compact, std_coeff=std_coeff)
        w0 = som._weights.copy()
        ids = som._upload_weights().bmu(data)               # the engine's BMUs from the initial codebook
        ref = O.bmu_ids(data, w0.astype(np.float32).reshape(-1, D), dist)
        diff = np.flatnonzero(ids != ref)                   # float32 near-ties may fall either way on this host's BLAS
        if len(diff):
            x64, w64 = data[diff].astype(np.float64), w0.reshape(-1, D).astype(np.float64)
            if dist == "cosine":
                with np.errstate(all="ignore"):
                    dd = 1 - np.nan_to_num((x64 @ w64.T) / np.sqrt((x64 ** 2).sum(1)[:, None] * (w64 ** 2).sum(1)[None, :]))
                scale = 1.0
            else:
                dd = (x64 ** 2).sum(1)[:, None] - 2 * x64 @ w64.T + (w64 ** 2).sum(1)[None, :]
                scale = ((x64 ** 2).sum(1) + (w64 ** 2).sum(1).max())[:, None]
            r = np.arange(len(diff))
            assert (np.abs(dd[r, ids[diff]] - dd[r, ref[diff]]) <= 4e-6 * np.squeeze(scale)).all(), "BMU mismatch beyond a near-tie"
        som.train(data, T, iter_beg=t_at, iter_end=t_at + 1)
        f = O.DECAYS[decay]
        sig_t, eta_t = f(sigma, 1, t_at, T), f(lr, 0.01, t_at, T)
        bmu, num, den, want = O.epoch(data, w0.astype(np.float32), eta_t, sig_t, wide=O.decay_is_wide(decay),
                                      n_parallel=max(n, 1), distance=dist, forced_bmu=ids, compact=compact, std_coeff=std_coeff,
                                      neighbourhood=neigh + ("_hex" if topo == "hexagonal" else ""))
        # units whose denominator is far from the float32 underflow / cancellation zone (elsewhere
        # `den != 0` and num/den are decided by rounding noise in the reference itself)
        # (mexican_hat denominators cancel: num and den are each good to ~1e-7 of their maxima, their ratio only
        #  where the denominator is not small against its maximum)
        live = np.abs(den[..., 0]) > 1e-3 * np.abs(den).max()
        got = som._weights
        err = np.abs(got[live] - want[live]).max() / max(np.abs(want[live]).max(), 1e-30) if live.any() else 0.0
        qe, oqe = err, err
        ok = err < (5e-4 if neigh == "mexican_hat" else 2e-4)      # (its denominators cancel: x1e3 on 2e-7 at the mask edge)
    except Exception as ex:                      # noqa: BLE001
        ok, err, qe, oqe = False, -1, -1, -1
        print("EXC", repr(ex)[:300])
    if not ok:
        bad += 1
        print(f"FAIL case {case}: {X}x{Y}x{D} n={n} {decay} {neigh} {topo} {dist} sigma={sigma} lr={lr} compact={compact} std={std_coeff}: w err {err:.2e} qe {qe:.6f} vs {oqe:.6f}", flush=True)
print(f"{n_cases} cases, {bad} failures, {time.time()-t0:.1f} s")
