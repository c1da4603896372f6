#!/usr/bin/env python3
"""Differential fuzz of the oracle against the reference itself -- runs ONLY in the build container.

The golden fixtures pin the oracle on fixed inputs; this script pins it on random ones: it imports the reference
NumPy path from /root/reference (never copied, never shipped), draws random configurations (map, features, rows,
decay, neighbourhood, topology, distance, compact_support, std_coeff, sigma, learning rate, schedule position) and
compares, for each, the reference's own outputs with the oracle's restatement:
  * one epoch of `train(..., iter_beg=t, iter_end=t+1)` from the same codebook  -> merged codebook (and the BMUs)
  * `winner`, `quantization_error`, `topographic_error` on the trained state
Both run the same NumPy on the same host, so the comparison is bit-level except where the reference's mini-batch
split (`n_parallel`) changes the float32 accumulation order.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/diff_reference.py [seed] [cases]
"""
import contextlib
import io
import os
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

warnings.filterwarnings("ignore")
F32 = np.float32
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0
worst = 0.0
for case in range(n_cases):
    # DIFF_BIG=1: maps up to 48 x 48, up to 300 features, up to 3000 rows -- where OpenBLAS threads the GEMMs and (past
    # 448 features only) splits K; the default stays small and fast (thousands of cases per minute)
    big = bool(os.environ.get("DIFF_BIG"))
    X, Y = int(rs.randint(2, 49 if big else 13)), int(rs.randint(2, 49 if big else 13))
    D = int(rs.choice([1, 2, 3, 5, 8, 17] + ([32, 64, 100, 128, 200, 300] if big else [])))
    n = int(rs.choice([1, 7, 40, 150, 400] + ([1000, 3000] if big else [])))
    decay = str(rs.choice(["linear", "exponential", "asymptotic"]))
    neigh = str(rs.choice(["gaussian", "mexican_hat", "bubble", "triangle"]))
    topo = "rectangular" if neigh == "triangle" else str(rs.choice(["rectangular", "hexagonal"]))
    dist = str(rs.choice(["euclidean", "euclidean", "cosine", "euclidean_no_opt", "manhattan", "norm_p"]))
    p_norm = int(rs.choice([1, 2, 3, 4]))
    compact = bool(neigh in ("gaussian", "triangle") and rs.rand() < 0.4)
    if neigh == "mexican_hat" and rs.rand() < 0.4:        # the reference's double mask on px: hexagonal, or square maps
        compact = True
        if topo == "rectangular":
            Y = X
    sigma = float(rs.choice([0, 1.0, 1.5, 3.0, 5.0])) or min(X, Y) / 2
    lr = float(rs.choice([0.5, 0.1, 1.0]))
    std_coeff = float(rs.choice([0.5, 0.25, 1.0]))
    T = int(rs.choice([1, 3, 10]))
    t_at = int(rs.randint(0, T))
    n_par = int(rs.choice([0, 0, 7, 64]))
    data = O.gaussian_blobs(n, D, seed=case + 5000)
    if dist == "cosine":
        data = np.abs(data)
    kw = {"p": p_norm} if dist == "norm_p" else {}
    msgs = []
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            ref = RefSom(X, Y, D, sigma=sigma, learning_rate=lr, decay_function=decay, neighborhood_function=neigh,
                         topology=topo, activation_distance=dist, activation_distance_kwargs=kw, random_seed=case,
                         compact_support=compact, std_coeff=std_coeff, n_parallel=n_par, xp=np)
        w0 = ref._weights.copy()
        assert np.array_equal(w0, O.default_codebook(X, Y, D, case)), "default codebook"
        npar = ref._n_parallel
        ref.train(data, T, iter_beg=t_at, iter_end=t_at + 1)
        want = ref._weights
        f = O.DECAYS[decay]
        sig_t, eta_t = f(sigma, 1, t_at, T), f(lr, 0.01, t_at, T)
        okw = dict(distance=dist, compact=compact, std_coeff=std_coeff, neighbourhood=neigh + ("_hex" if topo == "hexagonal" and neigh != "triangle" else ""))
        forced = None
        if dist in ("manhattan", "norm_p"):                # the pairwise distances: BMUs from their own restatement
            forced = O.bmu_ids_pairwise(data.astype(F32), w0.astype(F32).reshape(-1, D), dist, p_norm)
            okw["distance"] = "euclidean"
        _, _, _, got = O.epoch(data.astype(F32), w0.astype(F32), eta_t, sig_t, wide=O.decay_is_wide(decay), n_parallel=npar,
                               forced_bmu=forced, **okw)
        scale = max(np.abs(want).max(), 1e-30)
        err = np.abs(got - want).max() / scale
        worst = max(worst, err)
        if not err <= 1e-6:
            msgs.append("epoch %.2e" % err)
        # inference on the trained state
        q = data[: min(n, 50)]
        rw = ref.winner(q) if len(q) > 0 else []
        if dist in ("manhattan", "norm_p"):
            # (mini-batches of n_parallel rows as the reference's winner() forms them: norm_p with an even p is a sum of
            #  GEMMs, and a GEMM's float32 rounding depends on its shape)
            ow = np.concatenate([O.bmu_ids_pairwise(np.asarray(q)[s:s + npar], want.reshape(-1, D), dist, p_norm)
                                 for s in range(0, len(q), npar)]) if len(q) else []
        else:
            ow = O.winner_ids(q, want, dist, n_parallel=npar)    # (the same chunks: a GEMM's rounding depends on its shape)
        if [tuple(map(int, t)) for t in rw] != [(int(k) // Y, int(k) % Y) for k in ow]:
            msgs.append("winner")
        rq, oq = ref.quantization_error(q), O.quantization_error(q, want, n_parallel=npar)
        if abs(rq - oq) > 1e-6 * max(abs(rq), 1e-30):
            msgs.append("QE %.8g vs %.8g" % (rq, oq))
        if X * Y > 1 and (topo == "rectangular" or X == Y):
            rt, ot = ref.topographic_error(q), O.topographic_error(q, want, topo, n_parallel=npar)
            if abs(rt - ot) > 1e-12:
                msgs.append("TE %.6f vs %.6f" % (rt, ot))
    except Exception as ex:                               # noqa: BLE001
        msgs.append("EXC " + repr(ex)[:200])
    if msgs:
        bad += 1
        print(f"FAIL case {case}: {X}x{Y}x{D} n={n} {decay} {neigh} {topo} {dist}(p={p_norm}) sigma={sigma} lr={lr} compact={compact} "
              f"std={std_coeff} T={T} t={t_at} n_parallel={n_par}: " + "; ".join(msgs), flush=True)
print(f"{n_cases} cases, {bad} failures, worst epoch deviation {worst:.2e}")
