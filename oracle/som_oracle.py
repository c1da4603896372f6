"""CPU oracle for the batch-SOM hot path -- TEST INFRASTRUCTURE ONLY.

This module is a from-scratch NumPy restatement of the algorithm the
reference (jcfaracco/xpysom-dask @ 2024-10-08) runs on its NumPy path.  It is
the *checker* for the HIP engine: only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under
``xpysom_dask_amd/`` imports it, and the product never falls back to it.

Parity status: PINNED.  ``oracle/make_golden.py`` imported the reference in
the build container and wrote ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against those vectors (bit-exact where the
arithmetic is the same NumPy expression, 1e-6 where only BLAS summation order
may differ).

Conventions
-----------
* The codebook is handled flat: ``W`` is ``(K, D)`` with ``k = i*Y + j``
  (reference: ``xpysom.py:240`` unravel table, ``distances.py:185`` reshape).
* dtype flow follows SURVEY.md section 3.4: data/codebook/accumulators are
  float32; the neighbourhood is float64 when ``sigma`` is a ``numpy.float64``
  (what ``exponential_decay`` returns) and float32 when it is a Python float
  (``linear``/``asymptotic``) -- the NumPy >= 2 promotion rule.  Here the
  choice is the explicit argument ``wide``.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32
F64 = np.float64


# --------------------------------------------------------------------------
# schedules -- reference xpysom_dask/decays.py:4-65
# --------------------------------------------------------------------------
def asymptotic_decay(v0, vN, t, T):
    """decays.py:4-20 -- ``v0 / (1 + 2t/T)`` (vN unused). Python float in, Python float out."""
    return v0 / (1 + 2 * t / T)


def exponential_decay(v0, vN, t, T):
    """decays.py:23-43 -- geometric interpolation v0 -> vN; returns numpy.float64
    because numpy's exp/log are used (this is what makes the neighbourhood float64)."""
    rate = (-np.log(0.1) if vN == 0 else -np.log(vN / v0)) / T
    return v0 * np.exp(-t * rate)


def linear_decay(v0, vN, t, T):
    """decays.py:46-65 -- straight line reaching vN at t = T-1 (T == 1 -> v0)."""
    if T == 1:
        return v0
    return v0 + (vN - v0) * t / (T - 1)


DECAYS = {
    "exponential": exponential_decay,
    "asymptotic": asymptotic_decay,
    "linear": linear_decay,
}


def decay_is_wide(name):
    """True when the schedule hands back numpy.float64."""
    return name == "exponential"


# --------------------------------------------------------------------------
# initial codebook -- reference xpysom.py:167,189-190
# --------------------------------------------------------------------------
def default_codebook(X, Y, D, seed):
    """float64 (X,Y,D): uniform(-1,1) rows scaled to unit L2 norm."""
    rs = np.random.RandomState(seed)
    w = rs.rand(X, Y, D) * 2 - 1
    w /= np.linalg.norm(w, axis=-1, keepdims=True)
    return w


# --------------------------------------------------------------------------
# distances -- reference xpysom_dask/distances.py:11-59
# --------------------------------------------------------------------------
def row_sq(a):
    return np.power(a, 2).sum(axis=1, keepdims=True)


def dist_euclid_part(x, w, w_sq=None):
    """distances.py:11-23: -2 x.w^T + |w|^2 (the |x|^2 term is dropped)."""
    if w_sq is None:
        w_sq = row_sq(w)
    return -2 * np.dot(x, w.T) + w_sq.T


def dist_euclid_sq(x, w, w_sq=None):
    """distances.py:25-31."""
    return dist_euclid_part(x, w, w_sq) + row_sq(x)


def dist_euclid(x, w, w_sq=None):
    """distances.py:33-43: sqrt with NaN (negative by cancellation) mapped to 0."""
    with np.errstate(invalid="ignore"):
        return np.nan_to_num(np.sqrt(dist_euclid_sq(x, w, w_sq)))


def dist_cosine(x, w, w_sq=None):
    """distances.py:45-59: 1 - nan_to_num(x.w / sqrt(|x|^2 |w|^2))."""
    if w_sq is None:
        w_sq = row_sq(w)
    x_sq = row_sq(x)
    with np.errstate(invalid="ignore", divide="ignore"):
        sim = np.nan_to_num(np.dot(x, w.T) / np.sqrt(x_sq * w_sq.T))
    return 1 - sim


def dist_norm_p_generic(x, w, p=2):
    """distances.py:61-75: sum_d |x-w|^p through the (n, K, D) difference tensor."""
    return np.sum(np.power(np.abs(x[:, None, :] - w[None, :, :]), p), axis=2)


def dist_norm_p_even(x, w, p=2):
    """distances.py:77-96: binomial expansion, p+1 dot products accumulated in float64."""
    acc = np.zeros((len(x), len(w)))
    k = 1
    for e in range(p + 1):
        acc += (-1 if e % 2 == 1 else 1) * k * np.dot(x ** (p - e), (w ** e).T)
        k = (k * (p - e)) // (e + 1)
    return acc


def dist_norm_p(x, w, p=2):
    """distances.py:98-107."""
    return dist_norm_p_even(x, w, p) if p % 2 == 0 else dist_norm_p_generic(x, w, p)


def dist_manhattan(x, w):
    """distances.py:138-158 (NumPy path: the generic form with p = 1)."""
    return dist_norm_p_generic(x, w, 1)


DISTANCES = {
    "euclidean": dist_euclid_part,       # distances.py:163
    "euclidean_no_opt": dist_euclid_sq,  # distances.py:164
    "cosine": dist_cosine,               # distances.py:167
}


def bmu_ids_pairwise(x, w, name, p=2):
    """BMU ids for the distances that take no cached w_sq (distances.py:165-169)."""
    if name in ("manhattan", "manhattan_no_opt"):
        d = dist_manhattan(x, w)
    elif name == "norm_p":
        d = dist_norm_p(x, w, p)
    else:
        d = dist_norm_p_generic(x, w, p)
    return np.argmin(d, axis=1)


def bmu_ids(x, w, distance="euclidean", w_sq=None):
    """Raveled best-matching-unit ids, first minimum wins (xpysom.py:410-417)."""
    return np.argmin(DISTANCES[distance](x, w, w_sq), axis=1)


# --------------------------------------------------------------------------
# neighbourhoods (rectangular) -- reference xpysom_dask/neighborhoods.py:14-74
# --------------------------------------------------------------------------
def _support(n, c, sigma):
    return np.logical_and(n > c - sigma, n < c + sigma)


def neigh_gaussian(X, Y, std_coeff, compact, ci, cj, sigma, wide):
    """neighborhoods.py:14-33: separable gaussian around (ci, cj); (n, X, Y).

    ``wide`` selects float64 (sigma is numpy.float64) or float32 evaluation of
    ``exp(-delta^2/d)``; ``d = 2*std_coeff^2*sigma^2`` is always formed in
    double (Python arithmetic) first."""
    sigma = F64(sigma) if wide else float(sigma)
    d = 2 * std_coeff ** 2 * sigma ** 2
    ni = np.arange(X)[None, :]
    nj = np.arange(Y)[None, :]
    ci = np.asarray(ci)[:, None]
    cj = np.asarray(cj)[:, None]
    ax = np.exp(-np.power(ni - ci, 2, dtype=F32) / d)
    ay = np.exp(-np.power(nj - cj, 2, dtype=F32) / d)
    if compact:
        ax *= _support(ni, ci, sigma)
        ay *= _support(nj, cj, sigma)
    return ax[:, :, None] * ay[:, None, :]


def neigh_mexican_hat(X, Y, std_coeff, compact, ci, cj, sigma, wide):
    """neighborhoods.py:57-74.  With compact_support the reference multiplies px by the row mask AND by the
    column mask -- the latter an (n, Y) array against px's (n, X), so square maps only (NumPy refuses to broadcast
    otherwise, as in the reference), where it compares the ROW index with the BMU's COLUMN -- and leaves py
    unmasked (:69-71).  Restated literally: that is what a drop-in has to reproduce."""
    sigma = F64(sigma) if wide else float(sigma)
    d = 2 * std_coeff ** 2 * sigma ** 2
    ni = np.arange(X)[None, :]
    nj = np.arange(Y)[None, :]
    ci = np.asarray(ci)[:, None]
    cj = np.asarray(cj)[:, None]
    px = np.power(ni - ci, 2, dtype=F32)
    py = np.power(nj - cj, 2, dtype=F32)
    if compact:
        px *= _support(ni, ci, sigma)
        px *= _support(nj, cj, sigma)
    p = px[:, :, None] + py[:, None, :]
    return np.exp(-p / d) * (1 - 2 / d * p)


def neigh_bubble(X, Y, std_coeff, compact, ci, cj, sigma, wide):
    """neighborhoods.py:99-112: 1 inside the open box |di| < sigma, |dj| < sigma (float32)."""
    ni, nj = np.arange(X)[None, :], np.arange(Y)[None, :]
    ci, cj = np.asarray(ci)[:, None], np.asarray(cj)[:, None]
    return (_support(ni, ci, sigma)[:, :, None] * _support(nj, cj, sigma)[:, None, :]).astype(F32)


def neigh_triangle(X, Y, std_coeff, compact, ci, cj, sigma, wide):
    """neighborhoods.py:114-130: (sigma - |di|)+ * (sigma - |dj|)+, optionally masked to the open box.
    ``-abs(int64) + sigma`` is float64 for a Python-float sigma and for a numpy.float64 one alike, so the
    result is float64 whatever ``wide`` says (and so are g, num and den of the update that uses it)."""
    ni, nj = np.arange(X)[None, :], np.arange(Y)[None, :]
    ci, cj = np.asarray(ci)[:, None], np.asarray(cj)[:, None]
    sigma = F64(sigma) if wide else float(sigma)
    tx = (-np.abs(ci - ni)) + sigma
    ty = (-np.abs(cj - nj)) + sigma
    tx[tx < 0] = 0.
    ty[ty < 0] = 0.
    if compact:
        tx *= _support(ni, ci, sigma)
        ty *= _support(nj, cj, sigma)
    return tx[:, :, None] * ty[:, None, :]


def hex_coords(X, Y):
    """Euclidean unit coordinates of the hexagonal topology (xpysom.py:201-206):
    meshgrid (shape (Y, X)), every second row counted from the LAST one shifted by -0.5."""
    xx, yy = np.meshgrid(np.arange(X), np.arange(Y))
    xx, yy = xx.astype(float), yy.astype(float)
    xx[::-2] -= 0.5
    return xx, yy


def _generic_terms(X, Y, ci, cj):
    xx, yy = hex_coords(X, Y)
    nx, ny = xx[None, :, :], yy[None, :, :]
    cx = xx.T[(ci, cj)][:, None, None]
    cy = yy.T[(ci, cj)][:, None, None]
    return nx, ny, cx, cy


def neigh_gaussian_hex(X, Y, std_coeff, compact, ci, cj, sigma, wide):
    """gaussian_generic on the hexagonal grid, neighborhoods.py:35-55; (n, X, Y)."""
    sigma = F64(sigma) if wide else float(sigma)
    d = 2 * std_coeff ** 2 * sigma ** 2
    nx, ny, cx, cy = _generic_terms(X, Y, np.asarray(ci), np.asarray(cj))
    ax = np.exp(-np.power(nx - cx, 2, dtype=F32) / d)
    ay = np.exp(-np.power(ny - cy, 2, dtype=F32) / d)
    if compact:
        ax *= np.logical_and(nx > cx - sigma, nx < cx + sigma)
        ay *= np.logical_and(ny > cy - sigma, ny < cy + sigma)
    return (ax * ay).transpose((0, 2, 1))


def neigh_mexican_hat_hex(X, Y, std_coeff, compact, ci, cj, sigma, wide):
    """mexican_hat_generic on the hexagonal grid, neighborhoods.py:76-97; compact_support masks px by the x box
    and by the y box and leaves py unmasked (:91-93), restated literally."""
    sigma = F64(sigma) if wide else float(sigma)
    d = 2 * std_coeff ** 2 * sigma ** 2
    nx, ny, cx, cy = _generic_terms(X, Y, np.asarray(ci), np.asarray(cj))
    px = np.power(nx - cx, 2, dtype=F32)
    py = np.power(ny - cy, 2, dtype=F32)
    if compact:
        px *= np.logical_and(nx > cx - sigma, nx < cx + sigma)
        px *= np.logical_and(ny > cy - sigma, ny < cy + sigma)
    p = px + py
    return (np.exp(-p / d) * (1 - 2 / d * p)).transpose((0, 2, 1))


NEIGHBOURHOODS = {
    "gaussian": neigh_gaussian,
    "mexican_hat": neigh_mexican_hat,
    "bubble": neigh_bubble,
    "triangle": neigh_triangle,
    "gaussian_hex": neigh_gaussian_hex,          # topology='hexagonal' registry, xpysom.py:271-279
    "mexican_hat_hex": neigh_mexican_hat_hex,
    "bubble_hex": neigh_bubble,
}


# --------------------------------------------------------------------------
# one mini-batch: BMUs -> g -> (numerator, denominator); xpysom.py:420-443
# --------------------------------------------------------------------------
def update(x, W3, eta, sigma, *, wide, distance="euclidean", neighbourhood="gaussian",
           std_coeff=0.5, compact=False, w_sq=None, forced_bmu=None):
    """Returns (bmu, num, den): bmu (n,) int64 raveled, num (X,Y,D), den (X,Y,1).

    num/den carry the dtype the reference's arithmetic produces (float64 when
    ``wide``).  ``forced_bmu`` teacher-forces the BMU ids (tests use it to
    isolate the accumulate path from near-tie BMU noise)."""
    X, Y, D = W3.shape
    w = W3.reshape(-1, D)
    if forced_bmu is None:
        bmu = bmu_ids(x, w, distance, w_sq)
    else:
        bmu = np.asarray(forced_bmu, dtype=np.int64)
    ci, cj = bmu // Y, bmu % Y
    eta = F64(eta) if wide else float(eta)
    g = NEIGHBOURHOODS[neighbourhood](X, Y, std_coeff, compact, ci, cj, sigma, wide) * eta
    den = g.sum(axis=0)[:, :, None]
    num = np.dot(g.reshape(len(x), -1).T, x).reshape(W3.shape)
    return bmu, num, den


def merge(W3, num, den):
    """xpysom.py:446-455: where(den != 0, num/den, W)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(den != 0, num / den, W3)


def epoch(data, W3, eta, sigma, *, wide, n_parallel, forced_bmu=None, **kw):
    """One full epoch from codebook W3 (float32): xpysom.py:515-577.
    Returns (bmu (N,), num f32 (X,Y,D), den f32 (X,Y,1), W_new f32)."""
    X, Y, D = W3.shape
    num = np.zeros(W3.shape, dtype=F32)
    den = np.zeros((X, Y, 1), dtype=F32)
    w_sq = None
    if kw.get("distance", "euclidean") in ("euclidean", "cosine"):   # can_cache, distances.py:179-182
        w_sq = row_sq(W3.reshape(-1, D))
    out_bmu = []
    for s in range(0, len(data), n_parallel):
        fb = None if forced_bmu is None else forced_bmu[s:s + n_parallel]
        b, a_num, a_den = update(data[s:s + n_parallel], W3, eta, sigma, wide=wide,
                                 w_sq=w_sq, forced_bmu=fb, **kw)
        num += a_num      # float64 -> float32 downcast on += when wide (xpysom.py:568-569)
        den += a_den
        out_bmu.append(b)
    return np.concatenate(out_bmu), num, den, merge(W3, num, den)


def train(data, W3, num_epochs, *, sigma0, sigmaN=1, lr0=0.5, lrN=0.01, decay="exponential",
          n_parallel=4000, iter_beg=0, iter_end=None, **kw):
    """xpysom.py:458-594 local branch. ``W3`` any float dtype; returns float32 codebook."""
    if iter_end is None:
        iter_end = num_epochs
    W3 = np.asarray(W3, dtype=F32)
    data = np.asarray(data, dtype=F32)
    f = DECAYS[decay]
    wide = decay_is_wide(decay)
    for t in range(iter_beg, iter_end):
        eta = f(lr0, lrN, t, num_epochs)
        sig = f(sigma0, sigmaN, t, num_epochs)
        _, _, _, W3 = epoch(data, W3, eta, sig, wide=wide, n_parallel=n_parallel, **kw)
    return W3


# --------------------------------------------------------------------------
# inference -- xpysom.py:370-408 (winner), :632-707 (quantization error)
# --------------------------------------------------------------------------
def winner_ids(x, W3, distance="euclidean", n_parallel=4000):
    """Raveled BMU ids exactly as ``winner`` finds them: configured distance,
    w_sq recomputed per chunk, no dtype coercion of x."""
    x = np.asarray(x)
    w = np.asarray(W3).reshape(-1, W3.shape[2])
    out = [bmu_ids(x[s:s + n_parallel], w, distance) for s in range(0, len(x), n_parallel)]
    return np.concatenate(out) if out else np.zeros(0, dtype=np.int64)


def quantization_ids(x32, W3, n_parallel=4000):
    """BMU ids as ``_quantization`` finds them: always the full Euclidean
    distance (sqrt + nan_to_num) regardless of the configured one (xpysom.py:640,670)."""
    w = W3.reshape(-1, W3.shape[2])
    out = [np.argmin(dist_euclid(x32[s:s + n_parallel], w), axis=1)
           for s in range(0, len(x32), n_parallel)]
    return np.concatenate(out)


def quantization_error(data, W3, n_parallel=4000):
    """xpysom.py:673-707 local branch: mean_n |x_n - W[bmu_n]| -> Python float."""
    x = np.array(data, dtype=F32)
    W3 = np.asarray(W3)
    ids = quantization_ids(x, W3, n_parallel)
    x = x - W3.reshape(-1, W3.shape[2])[ids]
    return np.linalg.norm(x, axis=1).mean().item()


def top2_ids(x32, W3, n_parallel=4000):
    """Best and second-best unit under the full Euclidean distance (xpysom.py:727-734).  The distances come in
    mini-batches of `n_parallel` rows as the reference forms them (_distance_from_weights, xpysom.py:660-671): a
    GEMM's float32 summation order depends on its shape once K is in the hundreds (oracle/diff_reference.py with
    DIFF_BIG=1 found a 200-feature case where one chunk of 150 rows and chunks of 7 rows order a near-tie differently)."""
    x32 = np.asarray(x32, dtype=F32)
    w = W3.reshape(-1, W3.shape[2])
    d = np.vstack([dist_euclid(x32[s:s + n_parallel], w) for s in range(0, len(x32), n_parallel)]) if len(x32) else \
        np.zeros((0, len(w)), F32)
    return np.argsort(d, axis=1)[:, :2]


def topographic_error(data, W3, topology="rectangular", n_parallel=4000):
    """xpysom.py:709-746: share of samples whose two best units are not adjacent -- rectangular: |di| > 1 or
    |dj| > 1; hexagonal (:739-746): farther apart than 1.5 in the coordinates ``_xx[i, j], _yy[i, j]``.  The
    reference indexes its UNtransposed (Y, X) meshgrids with the map index (i, j), i.e. it reads
    ``x = j - s(i)/2, y = i`` (s: the shifted rows of xpysom.py:201-206); only square maps are well defined."""
    X, Y = W3.shape[:2]
    b = top2_ids(np.array(data, dtype=F32), np.asarray(W3), n_parallel)
    i, j = b // Y, b % Y
    if topology == "hexagonal":
        xx, yy = hex_coords(X, Y)
        dx, dy = np.diff(xx[i, j]), np.diff(yy[i, j])
        return (np.linalg.norm(np.hstack([dx, dy]), axis=1) > 1.5).mean().item()
    di, dj = np.abs(i[:, 0] - i[:, 1]), np.abs(j[:, 0] - j[:, 1])
    return ((di > 1) | (dj > 1)).mean().item()


# --------------------------------------------------------------------------
# synthetic workload shared by bench.py and the tests
# --------------------------------------------------------------------------
def gaussian_blobs(N, D, seed=1234, centres=64, spread=3.0):
    rng = np.random.default_rng(seed)
    c = rng.normal(0.0, spread, size=(centres, D))
    lab = rng.integers(0, centres, size=N)
    x = c[lab] + rng.normal(0.0, 1.0, size=(N, D))
    return x.astype(F32)


def smooth_sheet_codebook(X, Y, D, seed, anchors=9, amplitude=0.5, centre=None):
    """A smooth codebook that any host reproduces BIT FOR BIT from its arguments: bilinear interpolation between a coarse
    grid of seeded anchor vectors, evaluated with elementwise float64 products and sums only (no BLAS, no fused
    operations), then rounded to float32.  Neighbouring units differ by ~amplitude/side -- the state a SOM is in
    early in its schedule, where every row has hundreds of near-best units -- without a 33 MB trained codebook having
    to be stored next to the reference's answers for it (tests/golden/g18)."""
    rs = np.random.RandomState(seed)
    A = rs.randn(anchors, anchors, D) * amplitude
    if centre is not None:
        A = A + np.asarray(centre, dtype=F64)[None, None, :]
    gi = np.arange(X, dtype=F64) * ((anchors - 1) / max(X - 1, 1))
    gj = np.arange(Y, dtype=F64) * ((anchors - 1) / max(Y - 1, 1))
    i0 = np.minimum(gi.astype(np.int64), anchors - 2)
    j0 = np.minimum(gj.astype(np.int64), anchors - 2)
    fi = (gi - i0)[:, None, None]
    fj = (gj - j0)[None, :, None]
    a00 = A[i0][:, j0]
    a10 = A[i0 + 1][:, j0]
    a01 = A[i0][:, j0 + 1]
    a11 = A[i0 + 1][:, j0 + 1]
    top = a00 * (1.0 - fi)
    top = top + a10 * fi
    bot = a01 * (1.0 - fi)
    bot = bot + a11 * fi
    w = top * (1.0 - fj)
    w = w + bot * fj
    return w.astype(F32)


def sheet_step(w0, w_other, mix):
    """(1 - mix) * w0 + mix * w_other, elementwise in float64, rounded to float32: a "next" codebook state any host
    evaluates bit for bit (tests/golden/g20)."""
    a = np.asarray(w0, dtype=F64) * (1.0 - mix)
    a = a + np.asarray(w_other, dtype=F64) * mix
    return a.astype(F32)


def rows_on_codebook(w, n, seed, noise):
    """n rows scattered around randomly chosen units of a codebook: unit + noise * N(0, I), elementwise float64, rounded to
    float32 (RandomState streams are reproducible across hosts and NumPy versions)."""
    rs = np.random.RandomState(seed)
    wf = np.asarray(w, dtype=F64).reshape(-1, np.shape(w)[-1])
    pick = rs.randint(0, wf.shape[0], size=n)
    x = wf[pick] + rs.standard_normal((n, wf.shape[1])) * noise
    return x.astype(F32)

