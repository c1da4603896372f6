#!/usr/bin/env python3
"""Differential fuzz of the product's HOST class against the reference class -- runs ONLY in the build container.

`xpysom_dask_amd.XPySom` (validation, seeded initialisation, schedules, epoch loop, tuple formatting, the batched
analysis helpers, pickling) is driven here over `tests/oracle_engine.OracleEngine`, the NumPy test double with the
HipEngine interface, and compared method by method with the reference class imported from /root/reference (never
copied, never shipped) on random configurations.  The kernels are not involved: this pins the Python above the C ABI.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/diff_host_reference.py [seed] [cases]
"""
import contextlib
import io
import os
import pickle
import sys
import warnings

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True
with contextlib.redirect_stdout(io.StringIO()):     # silence the CuPy/Dask import warnings
    sys.path.insert(0, "/root/reference")
    from xpysom_dask import XPySom as RefSom          # noqa: E402

from oracle import som_oracle as O                    # noqa: E402
import xpysom_dask_amd.engine as engine_mod           # noqa: E402
from tests.oracle_engine import OracleEngine          # noqa: E402
from xpysom_dask_amd import XPySom                    # noqa: E402

engine_mod.HipEngine = OracleEngine                   # the host class over the NumPy double (no GPU here)
warnings.filterwarnings("ignore")
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0


def same(a, b, tol=0.0):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype.kind == "f" and not np.array_equal(np.isnan(a), np.isnan(b)):
        return False                                       # (a NaN must be a NaN on both sides: 0/0 in distance_map)
    if tol == 0.0:
        return bool(np.array_equal(a, b, equal_nan=a.dtype.kind == "f"))
    d = np.nan_to_num(np.abs(a.astype(np.float64) - b.astype(np.float64)))
    return bool(d.max() <= tol * max(np.nanmax(np.abs(b)) if np.isfinite(b).any() else 0.0, 1e-30))


for case in range(n_cases):
    X, Y = int(rs.randint(2, 11)), int(rs.randint(2, 11))
    D = int(rs.choice([1, 2, 4, 7]))
    n = int(rs.choice([5, 30, 120]))
    decay = str(rs.choice(["linear", "exponential", "asymptotic"]))
    neigh = str(rs.choice(["gaussian", "mexican_hat", "bubble", "triangle"]))
    topo = "rectangular" if neigh == "triangle" else str(rs.choice(["rectangular", "hexagonal"]))
    dist = str(rs.choice(["euclidean", "euclidean", "cosine", "euclidean_no_opt"]))
    sigma = float(rs.choice([0, 1.0, 1.5, 3.0])) or min(X, Y) / 2
    lr = float(rs.choice([0.5, 0.1]))
    T = int(rs.choice([1, 2, 5]))
    init = str(rs.choice(["default", "random", "pca"]))
    data = O.gaussian_blobs(n, D, seed=case + 9000)
    if dist == "cosine":
        data = np.abs(data) + 0.01
    labels = [int(v) for v in rs.randint(0, 4, size=n)]
    kw = dict(sigma=sigma, learning_rate=lr, decay_function=decay, neighborhood_function=neigh, topology=topo,
              activation_distance=dist, random_seed=case, n_parallel=4000, sigmaN=float(rs.choice([1, 1, 0.5, 0])),
              learning_rateN=float(rs.choice([0.01, 0.01, 0])), std_coeff=float(rs.choice([0.5, 0.25, 1.0])),
              compact_support=bool(neigh in ("gaussian", "triangle") and rs.rand() < 0.3))
    msgs = []
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            ref = RefSom(X, Y, D, xp=np, **kw)
        som = XPySom(X, Y, D, **kw)
        if not same(ref._weights, som._weights): msgs.append("default codebook")
        if init == "random":
            ref.random_weights_init(data); som.random_weights_init(data)
        elif init == "pca" and D > 1 and n > 2:
            ref.pca_weights_init(data); som.pca_weights_init(data)
        if not same(ref._weights, som._weights, 1e-12): msgs.append(init + " init")
        som._weights = ref._weights.copy()                    # (pca: LAPACK sign noise aside, continue from one state)
        exc = []
        for m in (ref, som):                                   # an exception must be the same exception on both sides
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    m.train(data, T)
                exc.append(None)
            except Exception as ex:                            # noqa: BLE001
                exc.append((type(ex).__name__, str(ex)))
        if exc[0] != exc[1]:
            msgs.append("train raises %r vs %r" % (exc[0], exc[1]))
        if exc[0] is not None or exc[1] is not None:
            if msgs:
                bad += 1
                print(f"FAIL case {case}: {X}x{Y}x{D} {decay} {neigh} {topo} {dist}: " + "; ".join(msgs), flush=True)
            continue
        if not same(ref._weights, som._weights, 2e-6): msgs.append("train %.2e" % np.abs(ref._weights - som._weights).max())
        if ref._weights.dtype != som._weights.dtype: msgs.append("weights dtype")
        som._weights = ref._weights.copy()
        q = data[: min(n, 40)]
        rw, sw = ref.winner(q), som.winner(q)
        if [tuple(map(int, t)) for t in rw] != [tuple(map(int, t)) for t in sw] or type(rw[0][0]) is not type(sw[0][0]): msgs.append("winner")
        if not same(ref.quantization(q), som.quantization(q), 1e-7): msgs.append("quantization")
        a, b = ref.quantization_error(q), som.quantization_error(q)
        if type(a) is not type(b) or abs(a - b) > 1e-6 * max(abs(a), 1e-30): msgs.append("QE %r %r" % (a, b))
        if topo == "rectangular" or X == Y:
            a, b = ref.topographic_error(q), som.topographic_error(q)
            if abs(a - b) > 1e-12: msgs.append("TE %r %r" % (a, b))
        if not same(ref.distance_map(), som.distance_map(), 1e-6): msgs.append("distance_map")
        if not same(ref.activation_response(q), som.activation_response(q)): msgs.append("activation_response")
        rwm, swm = ref.win_map(q), som.win_map(q)
        if list(rwm) != list(swm) or any(not same(np.array(rwm[k]), np.array(swm[k])) for k in rwm): msgs.append("win_map")
        rlm, slm = ref.labels_map(q, labels[: len(q)]), som.labels_map(q, labels[: len(q)])
        if list(rlm) != list(slm) or any(rlm[k] != slm[k] for k in rlm): msgs.append("labels_map")
        if dist in ("euclidean", "cosine", "euclidean_no_opt"):
            if not same(ref.activate(q[0]), som.activate(q[0]), 2e-6): msgs.append("activate")
        if not same(ref.distance_from_weights(q, None), som.distance_from_weights(q, None), 2e-6): msgs.append("distance_from_weights")
        if not same(ref.get_weights(), som.get_weights()): msgs.append("get_weights")
        if topo == "rectangular" or X == Y:                   # (the reference's convert_map_to_euclidean indexes (Y, X) grids)
            ij = (int(rs.randint(0, X)), int(rs.randint(0, Y)))
            if not same(ref.convert_map_to_euclidean(ij), som.convert_map_to_euclidean(ij)): msgs.append("convert_map_to_euclidean")
        for ra, sa in zip(ref.get_euclidean_coordinates(), som.get_euclidean_coordinates()):
            if not same(ra, sa): msgs.append("get_euclidean_coordinates")
        back = pickle.loads(pickle.dumps(som))
        if not same(back._weights, som._weights) or back.winner(q) != sw and [tuple(map(int, t)) for t in back.winner(q)] != [tuple(map(int, t)) for t in sw]:
            msgs.append("pickle")
    except Exception as ex:                               # noqa: BLE001
        msgs.append("EXC " + repr(ex)[:300])
    if msgs:
        bad += 1
        print(f"FAIL case {case}: {X}x{Y}x{D} n={n} {decay} {neigh} {topo} {dist} sigma={sigma} lr={lr} T={T} init={init}: " + "; ".join(msgs), flush=True)
print(f"{n_cases} cases, {bad} failures")
