"""Randomised shape fuzz of the C ABI against the oracle: map sides 1..40, 1..300 features, 1..3000 rows, every
precision, every distance / neighbourhood / topology the engine implements.  BMUs must be the oracle's or near-best within the precision's bound; the
accumulators must match the oracle's update from the engine's own BMUs to 1e-5."""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
from oracle import som_oracle as O
from xpysom_dask_amd.engine import HipEngine

F32 = np.float32
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bad = 0
t0 = time.time()
for case in range(n_cases):
    side = int(os.environ.get("FUZZ_MAXSIDE", "40"))      # > 64 reaches the 256 x 256 tiles of the tiled kernel (K >= 4096)
    X, Y = int(rs.randint(1, side + 1)), int(rs.randint(1, side + 1))
    D = int(rs.choice([1, 2, 3, 7, 16, 31, 32, 33, 64, 100, 128, 129, 130, 200, 257, 300]))
    if os.environ.get("FUZZ_BIGD") and rs.rand() < 0.6:       # with FUZZ_MAXSIDE > 64: the wide kernel's instances (<= 800) and past them
        D = int(rs.choice([160, 224, 333, 416, 512, 640, 784, 800, 801, 900]))
    n = int(rs.choice([1, 2, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1000, 3000]))
    prec = str(rs.choice(["f32", "bf16", "exact", "f16", "exact"]))
    dist = str(rs.choice(["euclidean", "cosine"]))
    neigh = str(rs.choice(["gaussian", "gaussian", "mexican_hat", "bubble", "triangle"]))
    topo = str(rs.choice(["rectangular", "rectangular", "hexagonal"]))
    if neigh == "triangle":
        topo = "rectangular"                                  # (the hexagonal registry has no triangle, xpysom.py:271-279)
    compact = bool(neigh in ("gaussian", "triangle") and rs.rand() < 0.3)
    if neigh == "mexican_hat" and rs.rand() < 0.3:            # the reference's double mask on px: hexagonal, or square maps
        compact = True
        if topo == "rectangular":
            Y = X
    std_coeff = float(rs.choice([0.5, 0.5, 0.25, 1.0]))
    p_norm = 2
    if prec in ("f32", "exact") and rs.rand() < 0.3:          # the VALU distances exist in f32 only (and serve 'exact')
        dist = str(rs.choice(["manhattan", "norm_p", "norm_p_no_opt", "euclidean_no_opt"]))
        p_norm = int(rs.choice([1, 2, 3, 4]))
        D = min(D, 64)
    data = O.gaussian_blobs(n, D, seed=case)
    w = O.default_codebook(X, Y, D, case + 1).astype(F32) * 3
    # magnitudes, zero rows, duplicated units
    half = prec.startswith("f16")                           # IEEE half operands: norms must stay below 65504
    data = data * F32(rs.choice([1e-3, 1.0, 1.0, 10.0 if half else 1e3]))
    w = w * F32(rs.choice([1e-2, 1.0, 1.0, 10.0 if half else 1e2]))
    if n > 4 and rs.rand() < 0.3:
        data[rs.randint(0, n, size=max(1, n // 50))] = 0
    if X * Y > 3 and rs.rand() < 0.3:
        wf0 = w.reshape(-1, D)
        wf0[rs.randint(0, X * Y)] = wf0[rs.randint(0, X * Y)]
    if dist == "cosine":
        data, w = np.abs(data), np.abs(w)
    try:
        e = HipEngine(X, Y, D, precision=prec, distance=dist, neighborhood=neigh, topology=topo, norm_p=p_norm,
                      compact_support=compact, std_coeff=std_coeff)
        e.set_weights(w); e.set_data(data)
        # sigma: half the map, or a value on / one ulp off the unit lattice (the support masks' boundary cases)
        sig = float(rs.choice([max(min(X, Y) / 2, 1.0), 1.0, 2.5, 3.0, 5 / (1 + 2 / 3), 1.7320508]))
        eta = 0.5
        e.epoch_accumulate(sig, eta, True)
        num, den, bmu = e.epoch_fetch()
        q = e.bmu(data)
        wf = w.reshape(-1, D)
        x64, w64 = data.astype(np.float64), wf.astype(np.float64)
        if dist in ("manhattan", "norm_p", "norm_p_no_opt"):
            pp = 1 if dist == "manhattan" else p_norm
            dd = (np.abs(x64[:, None, :] - w64[None, :, :]) ** pp).sum(-1)
            scale = dd.max(1) + 1e-30
        elif dist in ("euclidean", "euclidean_no_opt"):
            # squared distances: the operand rounding perturbs x.w by eps |x||w|, i.e. d^2 by that much
            dd = np.maximum((x64 ** 2).sum(1)[:, None] - 2 * x64 @ w64.T + (w64 ** 2).sum(1)[None, :], 0)
            scale = (np.linalg.norm(x64, axis=1) + np.linalg.norm(w64, axis=1).max()) ** 2
        else:
            with np.errstate(all="ignore"):
                dd = 1 - np.nan_to_num((x64 @ w64.T) / np.sqrt((x64 ** 2).sum(1)[:, None] * (w64 ** 2).sum(1)[None, :]))
            scale = np.ones(n)
        # (two units are compared, each with its own operand and norm rounding: 2 * (2^-8 |x||w| + 2^-9 |w|^2) in d^2 for
        #  bf16 -- up to 2^-6.4 of the scale when |w| >> |x|; seed 32 case 377 sits at 1.11 x 2^-7)
        tol = {"f32": 2.0 ** -18, "exact": 2.0 ** -18, "bf16": 2.0 ** -6, "f16": 2.0 ** -9}[prec]
        ok_bmu = (dd[np.arange(n), bmu] <= dd.min(1) + tol * scale).all() and (dd[np.arange(n), q] <= dd.min(1) + tol * scale).all()
        _, onum, oden = O.update(data, w, np.float64(eta), np.float64(sig), wide=True, forced_bmu=bmu, compact=compact,
                                 std_coeff=std_coeff, neighbourhood=neigh + ("_hex" if topo == "hexagonal" else ""))
        en = np.abs(num - onum.reshape(-1, D)).max() / max(np.abs(onum).max(), 1e-30)
        ed = np.abs(den - oden.reshape(-1)).max() / max(np.abs(oden).max(), 1e-30)
        ok = ok_bmu and en < 1e-5 and ed < 1e-5 and (q != bmu).mean() <= 0.02
    except Exception as ex:                      # noqa: BLE001
        ok, en, ed = False, -1, -1
        print("EXC", repr(ex)[:200])
    if not ok:
        bad += 1
        print(f"FAIL case {case}: {X}x{Y}x{D} n={n} {prec} {dist}(p={p_norm}) {neigh} {topo} compact={compact} std={std_coeff} sigma={sig!r} num {en:.2e} den {ed:.2e} bmu_ok {ok_bmu if en >= 0 else None}", flush=True)
        if en >= 0 and not ok_bmu:                            # how far outside the bound, in units of the bound
            ex = (dd[np.arange(n), bmu] - dd.min(1)) / (tol * scale)
            exq = (dd[np.arange(n), q] - dd.min(1)) / (tol * scale)
            print(f"   worst excess: resident {ex.max():.3f}x the bound (row {ex.argmax()}), query {exq.max():.3f}x; |x| range {np.linalg.norm(x64, axis=1).min():.3g}..{np.linalg.norm(x64, axis=1).max():.3g}, |w| max {np.linalg.norm(w64, axis=1).max():.3g}; q!=bmu {np.mean(q != bmu):.4f}", flush=True)
print(f"{n_cases} cases, {bad} failures, {time.time()-t0:.1f} s")
