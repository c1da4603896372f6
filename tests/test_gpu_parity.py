"""Parity of the HIP hot path (through the C ABI) against the golden vectors captured from
the reference and against the pinned NumPy oracle.  GPU only (`-m gpu`).

Tolerances (north_star): BMU ids bit-exact wherever the arithmetic is exact or the top-2 gap
is above float32 noise; codebook / numerator / denominator within 1e-5 relative."""
import numpy as np
import pytest

from oracle import som_oracle as O
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
F32 = np.float32


def engine(X, Y, D, **kw):
    from xpysom_dask_amd.engine import HipEngine
    return HipEngine(X, Y, D, **kw)


def near_tie_mask(x, w, tol=2e-6):
    """True for samples whose best and second-best squared distance (float64) differ by
    less than tol * (|x|^2 + |w|^2 scale): there the float32 BMU is summation-order noise."""
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    d = -2 * x64 @ w64.T + (w64 ** 2).sum(1)[None, :]
    if d.shape[1] < 2:
        return np.zeros(len(x), dtype=bool)
    part = np.partition(d, 1, axis=1)
    scale = (x64 ** 2).sum(1) + np.abs(part[:, 0]) + 1e-30
    return (part[:, 1] - part[:, 0]) < tol * scale


def bf16_misses_are_near_best(x, w, bmu, bad):
    """bf16 mode measures |x~ - w~| on operands rounded to 8 significant bits: a unit it picks
    instead of the float32 BMU must be no farther than the best one plus that rounding
    (|delta x| + |delta w| <= 2^-8 (|x| + |w|))."""
    x64, w64 = x[bad].astype(np.float64), w.astype(np.float64)
    dd = np.sqrt(((x64[:, None, :] - w64[None, :, :]) ** 2).sum(-1))
    got = dd[np.arange(len(bad)), bmu[bad]]
    slack = 2.0 ** -8 * (np.linalg.norm(x64, axis=1) + np.linalg.norm(w64, axis=1).max())
    return bool((got <= dd.min(1) + slack).all())


def rel_err(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def rel_elem(a, b, mag):
    """Largest ELEMENTWISE error relative to the element's own magnitude sum `mag` = (|g|^T |x|)[k, d], the quantity a
    float32 sum's rounding error is proportional to: a unit far from the data has a numerator orders of magnitude below
    the map's largest one, and a normwise bound (rel_err) says nothing about it.  (Relative to the element's VALUE the
    measure would punish cancellation inside a sum, which no summation order avoids.)"""
    a, b, mag = np.asarray(a, np.float64), np.asarray(b, np.float64), np.asarray(mag, np.float64)
    return float((np.abs(a - b) / np.maximum(mag, 1e-300)).max())


def abs_numerator(data, w3, eta, sig, wide, bmu, **kw):
    """(|g|^T |x|) for a non-negative neighbourhood: the oracle's numerator of the rows' absolute values."""
    return O.update(np.abs(data), w3, eta, sig, wide=wide, forced_bmu=bmu, **kw)[1].reshape(-1, data.shape[1])


# golden comparisons a float32 near-tie took away (the reference's accumulators belong to ITS BMUs): counted, and
# bounded by the last test of this module, instead of silently skipped
LOST = []
COMPARED = []


# ----------------------------------------------------------------------------- G1 exact ties
@pytest.mark.parametrize("precision", ["f32", "exact", "bf16"])
def test_g1_exact_ties_lowest_index(precision):
    g = load_golden("g1_ties")
    x, w = g["x"].astype(F32), g["w"].astype(F32)
    X, Y, D = w.shape
    e = engine(X, Y, D, precision=precision)
    e.set_weights(w)
    assert np.array_equal(e.bmu(x), g["ids"])            # small integers: exact in f32 AND bf16
    e.set_weights(np.zeros_like(w))
    assert np.array_equal(e.bmu(x), g["ids_zero"])
    e.set_weights(np.ones_like(w))
    assert np.array_equal(e.bmu(x), g["ids_same"])


def test_g2_binary_vectors_argmin_exact():
    """All binary-vector cases of the reference's distance tests: arithmetic is exact, so the
    argmin of every distance flavour must equal numpy's on the golden matrices."""
    g = load_golden("g2_distances")
    n = int(g["n_cases"])
    for c in range(n - 8):                                   # the last 8 are the fuzzy float cases
        x, w = g[f"c{c:03d}_x"].astype(F32), g[f"c{c:03d}_w"].astype(F32)
        K, D = w.shape
        for dist, key in (("euclidean", "part"), ("euclidean_no_opt", "sq"), ("cosine", "cos")):
            e = engine(K, 1, D, distance=dist)
            e.set_weights(w)
            assert np.array_equal(e.bmu(x), np.argmin(g[f"c{c:03d}_{key}"], axis=1)), (c, dist)
        e = engine(K, 1, D)
        e.set_weights(w)
        assert np.array_equal(e.bmu(x, quantization=True), np.argmin(g[f"c{c:03d}_l2"], axis=1)), c


# ----------------------------------------------------------------------------- G4/G5 update + epoch
SHAPES = ["6x6x4", "8x8x3", "24x24x16", "20x30x12"]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("decay", ["linear", "exponential"])
def test_g4_g5_update_and_epoch(shape, decay):
    g = load_golden("g4_update_" + shape)
    X, Y, D, n = (int(v) for v in g["shape"])
    T = int(g["T"])
    data = O.gaussian_blobs(n, D, seed=int(g["data_seed"]))
    w0 = O.default_codebook(X, Y, D, 1234).astype(F32)
    wmid = g[f"{decay}_wmid"]
    wide = O.decay_is_wide(decay)
    e = engine(X, Y, D)
    e.set_data(data)
    for tag, w in (("init", w0), ("mid", wmid), ("last", wmid)):
        eta, sig = float(g[f"{decay}_{tag}_eta"]), float(g[f"{decay}_{tag}_sig"])
        e.set_weights(w)
        e.epoch_accumulate(sig, eta, wide)
        num, den, bmu = e.epoch_fetch()
        ref_bmu = g[f"{decay}_{tag}_bmu"]
        diff = np.flatnonzero(bmu != ref_bmu)
        if len(diff):                                         # only float32 near-ties may differ
            assert len(diff) <= max(2, n // 500)
            assert near_tie_mask(data[diff], w.reshape(-1, D)).all()
        # accumulate path, teacher-forced on the engine's own BMUs
        _, onum, oden = O.update(data, w, eta if not wide else np.float64(eta), sig if not wide else np.float64(sig),
                                 wide=wide, forced_bmu=bmu)
        oden = oden.reshape(-1).astype(F32)
        ok = oden > 1e-30
        np.testing.assert_allclose(den[ok], oden[ok], rtol=1e-5)
        assert rel_err(num, onum.reshape(-1, D)) < 1e-5
        (LOST if len(diff) else COMPARED).append(("g4", shape, decay, tag))
        mag = abs_numerator(data, w, eta if not wide else np.float64(eta), sig if not wide else np.float64(sig), wide, bmu)
        assert rel_elem(num, onum.reshape(-1, D), mag) < 1e-5
        if not len(diff):                                     # same BMUs: the golden itself
            gden = g[f"{decay}_{tag}_den"].reshape(-1)
            np.testing.assert_allclose(den[ok], gden[ok], rtol=1e-5)
            if f"{decay}_{tag}_num" in g:
                assert rel_err(num, g[f"{decay}_{tag}_num"].reshape(-1, D)) < 1e-5
                assert rel_elem(num, g[f"{decay}_{tag}_num"].reshape(-1, D), mag) < 1e-5
        e.epoch_merge()
        wout = e.get_weights()
        if not len(diff):
            gw = g[f"{decay}_{tag}_wout"].reshape(-1, D)
            np.testing.assert_allclose(wout[ok], gw[ok], rtol=1e-5, atol=1e-5 * np.abs(gw).max())
        # units the reference leaves untouched (den == 0) keep their old weights
        zero = g[f"{decay}_{tag}_den"].reshape(-1) == 0
        if zero.any() and not len(diff):
            keep = den == 0
            assert np.array_equal(wout[keep], w.reshape(-1, D)[keep])


@pytest.mark.parametrize("decay", ["linear", "exponential"])
def test_g7_shard_partials_add_up(decay):
    """sum of per-shard accumulators == unsharded accumulator: the identity the RCCL
    all-reduce relies on (reference: per-block _update outputs are summed, xpysom.py:548-556)."""
    g = load_golden("g4_update_24x24x16")
    X, Y, D, n = (int(v) for v in g["shape"])
    data = O.gaussian_blobs(n, D, seed=int(g["data_seed"]))
    w = g[f"{decay}_wmid"]
    eta, sig = float(g[f"{decay}_mid_eta"]), float(g[f"{decay}_mid_sig"])
    wide = O.decay_is_wide(decay)
    e = engine(X, Y, D)
    e.set_weights(w)
    tot_num, tot_den = np.zeros((X * Y, D), F32), np.zeros(X * Y, F32)
    for part in np.array_split(np.arange(n), 2):
        e.set_data(data[part])
        e.epoch_accumulate(sig, eta, wide)
        num, den, _ = e.epoch_fetch()
        tot_num += num
        tot_den += den
    assert rel_err(tot_num, g[f"{decay}_shard2_num"].reshape(-1, D)) < 2e-5
    np.testing.assert_allclose(tot_den, g[f"{decay}_shard2_den"].reshape(-1), rtol=2e-5, atol=1e-30)


# ----------------------------------------------------------------------------- G6 end to end
@pytest.mark.parametrize("decay", ["linear", "exponential"])
def test_g6_iris_random_init_trajectory(decay):
    """random_weights_init copies data rows into the codebook; iris has exact distance ties, so
    the 100-epoch run is chaotic beyond float32 noise.  Parity is checked the well-posed way:
    every epoch teacher-forced from the reference's own codebook (iter_beg=t, iter_end=t+1)."""
    from xpysom_dask_amd import XPySom
    g = load_golden("g6_iris")
    z = g["iris_z"]
    traj = g[f"{decay}_random_traj"]
    som = XPySom(6, 6, 4, random_seed=10, decay_function=decay)
    som.random_weights_init(z)
    np.testing.assert_array_equal(som._weights, g[f"{decay}_random_w0"])
    prev = g[f"{decay}_random_w0"]
    worst = 0.0
    for t in range(100):
        som._weights = np.array(prev)
        som.train(z, 100, iter_beg=t, iter_end=t + 1)
        worst = max(worst, rel_err(som._weights, traj[t]))
        prev = traj[t]
    assert worst < 1e-5, worst
    ids = np.array([i * 6 + j for i, j in som.winner(z)])
    assert np.array_equal(ids, g[f"{decay}_random_bmu"])


@pytest.mark.parametrize("decay", ["linear", "exponential"])
@pytest.mark.parametrize("init", ["default", "pca"])
def test_g6_iris_end_to_end(decay, init):
    from xpysom_dask_amd import XPySom
    g = load_golden("g6_iris")
    z = g["iris_z"]
    som = XPySom(6, 6, 4, random_seed=10, decay_function=decay)
    if init == "pca":
        som.pca_weights_init(z)
    np.testing.assert_array_equal(som._weights, g[f"{decay}_{init}_w0"])     # host init bit-exact
    assert abs(som.quantization_error(z) - float(g[f"{decay}_{init}_qe0"])) < 1e-5
    som.train(z, 100)
    assert som._weights.dtype == np.float32
    # 100 epochs end to end: the reference against ITSELF under another mini-batch split agrees to 7e-7;
    # measured here 3e-7 .. 1.2e-6 normwise -- the bound is twice the worst
    assert rel_err(som._weights, g[f"{decay}_{init}_w"]) < 2.5e-6
    ids = np.array([i * 6 + j for i, j in som.winner(z)])
    assert np.array_equal(ids, g[f"{decay}_{init}_bmu"])
    assert abs(som.quantization_error(z) - float(g[f"{decay}_{init}_qe"])) < 1e-5


def test_g6_readme_config_quantization_error():
    from xpysom_dask_amd import XPySom
    g = load_golden("g6_iris")
    raw = g["iris_raw"]
    som = XPySom(6, 6, 4, sigma=0.3, learning_rate=0.5, random_seed=10)
    assert abs(som.quantization_error(raw) - float(g["readme_qe0"])) < 1e-5
    som.train(raw, 100)
    # the codebook of this config is chaotic (SURVEY 7 hard part 1); QE is the stable observable
    assert abs(som.quantization_error(raw) - float(g["readme_qe"])) < 0.1 * float(g["readme_qe"])


# ----------------------------------------------------------------------------- G8 cosine + mexican hat
@pytest.mark.parametrize("decay", ["linear", "exponential"])
def test_g8_cosine_mexican_hat(decay):
    g = load_golden("g8_cosine_mexican")
    data, w0 = g["data"], g[f"{decay}_w0"]
    X, Y, D = w0.shape
    wide = O.decay_is_wide(decay)
    e = engine(X, Y, D, distance="cosine", neighborhood="mexican_hat")
    e.set_weights(w0)
    e.set_data(data)
    e.epoch_accumulate(float(g[f"{decay}_sig"]), float(g[f"{decay}_eta"]), wide)
    num, den, bmu = e.epoch_fetch()
    assert np.array_equal(bmu, g[f"{decay}_bmu"])
    assert rel_err(num, g[f"{decay}_num"].reshape(-1, D)) < 1e-5
    assert rel_err(den, g[f"{decay}_den"].reshape(-1)) < 1e-5
    e.epoch_merge()
    gw = g[f"{decay}_wout"].reshape(-1, D)
    big = np.abs(g[f"{decay}_den"].reshape(-1)) > 1e-3 * np.abs(g[f"{decay}_den"]).max()
    np.testing.assert_allclose(e.get_weights()[big], gw[big], rtol=2e-4, atol=1e-5 * np.abs(gw[big]).max())


def test_g8_cosine_gaussian_epoch():
    from xpysom_dask_amd import XPySom
    g = load_golden("g8_cosine_mexican")
    data = g["data"]
    som = XPySom(8, 8, 6, random_seed=3, decay_function="linear", activation_distance="cosine")
    som._weights = np.abs(som._weights).astype(F32)
    ids = np.array([i * 8 + j for i, j in som.winner(data)])
    assert np.array_equal(ids, g["cosgauss_bmu"])
    som.train(data, 10, iter_beg=0, iter_end=1)
    np.testing.assert_allclose(som._weights, g["cosgauss_wout"], rtol=1e-5, atol=1e-6)


# ----------------------------------------------------------------------------- G9 inference
def test_g9_winner_and_quantization_error():
    from xpysom_dask_amd import XPySom
    g = load_golden("g9_inference")
    probe = O.gaussian_blobs(700, 10, seed=int(g["probe_seed"]))
    som = XPySom(16, 12, 10, random_seed=5, decay_function="linear")
    som._weights = g["w"]
    w = som.winner(probe)
    assert isinstance(w, list) and isinstance(w[0], tuple) and isinstance(w[0][0], np.int64)
    ids = np.array([i * 12 + j for i, j in w])
    bad = np.flatnonzero(ids != g["winner"])
    assert len(bad) <= 1 and near_tie_mask(probe[bad], g["w"].reshape(-1, 10)).all()
    one = som.winner(probe[3])
    assert one == (int(g["winner"][3]) // 12, int(g["winner"][3]) % 12) and isinstance(one[0], int)
    assert abs(som.quantization_error(probe) - float(g["qe"])) < 1e-5


def test_reference_unit_test_known_answers():
    """xpysom_dask/tests.py:31-33,77-79,98-121 without MiniSom."""
    from xpysom_dask_amd import XPySom
    som = XPySom(5, 5, 1, std_coeff=1)
    som._weights = np.zeros((5, 5, 1))
    som._weights[2, 3] = 5.0
    som._weights[1, 1] = 2.0
    assert som.quantization_error([[5], [2]]) == 0.0
    assert som.quantization_error([[4], [1]]) == 1.0
    assert som.winner([5.0]) == (2, 3)                      # activate(5.0).argmin() == 13
    q = som.quantization(np.array([[4], [2]]))
    assert q[0] == 5.0 and q[1] == 2.0
    resp = som.activation_response([[5.0], [2.0]])
    assert resp[2, 3] == 1 and resp[1, 1] == 1
    wm = som.win_map([[5.0], [2.0]])
    assert wm[(2, 3)][0] == [5.0] and wm[(1, 1)][0] == [2.0]
    lm = som.labels_map([[5.0], [2.0], [5.1]], ['a', 'b', 'a'])
    assert lm[(2, 3)]['a'] == 2 and lm[(1, 1)]['b'] == 1 and list(lm) == [(2, 3), (1, 1)]
    # determinism + "train lowers QE"
    rs = np.random.RandomState(1234)
    data = rs.rand(100, 2)
    a = XPySom(5, 5, 2, sigma=1.0, learning_rate=0.5, random_seed=1)
    b = XPySom(5, 5, 2, sigma=1.0, learning_rate=0.5, random_seed=1)
    np.testing.assert_array_almost_equal(a._weights, b._weights)
    a.train_random(data, 10)
    b.train_random(data, 10)
    np.testing.assert_array_almost_equal(a._weights, b._weights)
    som = XPySom(5, 5, 2, sigma=1.0, learning_rate=0.5, random_seed=1)
    d2 = np.array([[4, 2], [3, 1]])
    q1 = som.quantization_error(d2)
    som.train(d2, 10)
    assert q1 > som.quantization_error(d2)


# ----------------------------------------------------------------------------- shapes at the edges
@pytest.mark.parametrize("X,Y,D,n", [(1, 1, 1, 1), (1, 7, 3, 5), (9, 1, 2, 130), (13, 11, 33, 257),
                                     (40, 36, 70, 1000), (3, 3, 130, 64), (50, 50, 5, 129)])
@pytest.mark.parametrize("precision", ["f32", "bf16", "exact"])
def test_ragged_shapes_against_oracle(X, Y, D, n, precision):
    data = O.gaussian_blobs(n, D, seed=X * 100 + D)
    w = O.default_codebook(X, Y, D, 42).astype(F32) * 3
    e = engine(X, Y, D, precision=precision)
    e.set_weights(w)
    e.set_data(data)
    sig, eta = max(min(X, Y) / 2, 1.0), 0.5
    e.epoch_accumulate(sig, eta, True)
    num, den, bmu = e.epoch_fetch()
    ref = O.bmu_ids(data, w.reshape(-1, D))
    bad = np.flatnonzero(bmu != ref)
    if precision in ("f32", "exact"):
        assert near_tie_mask(data[bad], w.reshape(-1, D)).all()
    else:
        assert len(bad) <= 0.12 * n + 1
        assert bf16_misses_are_near_best(data, w.reshape(-1, D), bmu, bad)
    _, onum, oden = O.update(data, w, np.float64(eta), np.float64(sig), wide=True, forced_bmu=bmu)
    assert rel_err(num, onum.reshape(-1, D)) < 1e-5
    assert rel_err(den, oden.reshape(-1)) < 1e-5
    e.epoch_merge()
    want = O.merge(w, onum.astype(F32), oden.astype(F32)).reshape(-1, D)
    ok = oden.reshape(-1) > 1e-30
    np.testing.assert_allclose(e.get_weights()[ok], want[ok], rtol=1e-5, atol=1e-5 * np.abs(want).max())
    assert e.bmu(data[:0]).shape == (0,)                      # empty input


# ----------------------------------------------------------------------------- G11 bubble / triangle, rectangular
@pytest.mark.parametrize("shape", [(8, 8, 3, 500), (5, 7, 4, 300)])
@pytest.mark.parametrize("neigh,compact", [("bubble", False), ("triangle", False), ("triangle", True)])
def test_g11_bubble_triangle_against_the_reference(shape, neigh, compact):
    """neighborhoods.py:99-130 through the table-driven transform, against the reference's own _update and
    one-epoch outputs for all three schedules (the triangle is float64 in the reference whatever sigma's type)."""
    g = load_golden("g11_bubble_triangle")
    X, Y, D, n = shape
    data = O.gaussian_blobs(n, D, seed=400 + X)
    w0 = O.default_codebook(X, Y, D, 21).astype(F32)
    e = engine(X, Y, D, neighborhood=neigh, compact_support=compact)
    e.set_data(data)
    for decay in ("linear", "exponential", "asymptotic"):
        key = f"{X}x{Y}x{D}_{neigh}{'_cs' if compact else ''}_{decay}"
        e.set_weights(w0)
        e.epoch_accumulate(float(g[key + "_sig"]), float(g[key + "_eta"]), O.decay_is_wide(decay))
        num, den, bmu = e.epoch_fetch()
        assert np.array_equal(bmu, g[key + "_bmu"]), key
        assert rel_err(num, g[key + "_num"].reshape(-1, D)) < 1e-5, key
        assert rel_err(den, g[key + "_den"].reshape(-1)) < 1e-5, key
        e.epoch_merge()
        gw = g[key + "_wout"].reshape(-1, D)
        ok = g[key + "_den"].reshape(-1) > 1e-30
        np.testing.assert_allclose(e.get_weights()[ok], gw[ok], rtol=1e-5, atol=1e-5 * np.abs(gw).max())
        keep = g[key + "_den"].reshape(-1) == 0                  # outside every box: the old weights stay
        assert np.array_equal(e.get_weights()[keep], w0.reshape(-1, D)[keep])


@pytest.mark.parametrize("decay,tag", [("linear", "init"), ("exponential", "init"), ("exponential", "mid")])
@pytest.mark.parametrize("precision", ["f32", "exact", "bf16"])
def test_g12_configs1_map_against_the_reference(decay, tag, precision):
    """64x64x32 (BASELINE configs[1]'s map), one _update of 4096 rows: BMUs, denominator, strided numerator and
    merged rows of the reference itself."""
    g = load_golden("g12_update_64x64x32")
    X, Y, D, n = (int(v) for v in g["shape"])
    st = int(g["stride"])
    data = O.gaussian_blobs(n, D, seed=int(g["data_seed"]))
    w = O.default_codebook(X, Y, D, 1234).astype(F32) if tag == "init" else g["exponential_wmid"]
    key = f"{decay}_{tag}"
    e = engine(X, Y, D, precision=precision)
    e.set_weights(w)
    e.set_data(data)
    e.epoch_accumulate(float(g[key + "_sig"]), float(g[key + "_eta"]), O.decay_is_wide(decay))
    num, den, bmu = e.epoch_fetch()
    diff = np.flatnonzero(bmu != g[key + "_bmu"])
    if precision in ("f32", "exact"):
        if len(diff):                                         # only float32 near-ties (1-2 ulp gaps) may differ
            assert len(diff) <= max(2, n // 500) and near_tie_mask(data[diff], w.reshape(-1, D)).all()
    else:
        assert len(diff) <= n // 10 and bf16_misses_are_near_best(data, w.reshape(-1, D), bmu, diff)
    # accumulate path, teacher-forced on the engine's own BMUs, at the fixture's strided units
    _, onum, oden = O.update(data, w.reshape(X, Y, D), np.float64(g[key + "_eta"]) if O.decay_is_wide(decay) else float(g[key + "_eta"]),
                             np.float64(g[key + "_sig"]) if O.decay_is_wide(decay) else float(g[key + "_sig"]),
                             wide=O.decay_is_wide(decay), forced_bmu=bmu)
    oden = oden.reshape(-1).astype(F32)
    ok = oden > 1e-30
    np.testing.assert_allclose(den[ok], oden[ok], rtol=1e-5)
    assert rel_err(num, onum.reshape(-1, D)) < 1e-5
    mag = abs_numerator(data, w.reshape(X, Y, D), np.float64(g[key + "_eta"]) if O.decay_is_wide(decay) else float(g[key + "_eta"]),
                        np.float64(g[key + "_sig"]) if O.decay_is_wide(decay) else float(g[key + "_sig"]), O.decay_is_wide(decay), bmu)
    assert rel_elem(num, onum.reshape(-1, D), mag) < 1e-5
    if precision == "f32":
        (LOST if len(diff) else COMPARED).append(("g12", decay, tag))
    if len(diff):
        return
    gden = g[key + "_den"].reshape(-1)                        # same BMUs: the reference's own outputs
    ok = gden > 1e-30
    np.testing.assert_allclose(den[ok], gden[ok], rtol=1e-5)
    assert rel_err(num[::st], g[key + "_num16"]) < 1e-5
    assert rel_elem(num[::st], g[key + "_num16"], mag[::st]) < 1e-5
    e.epoch_merge()
    gw = g[key + "_wout16"]
    np.testing.assert_allclose(e.get_weights()[::st][ok[::st]], gw[ok[::st]], rtol=1e-5, atol=1e-5 * np.abs(gw).max())


@pytest.mark.parametrize("XD", [(5, 3), (12, 6)])
def test_g13_hexagonal_topographic_error(XD):
    """topographic_error on topology='hexagonal' as the reference evaluates it (xpysom.py:739-746) on square maps."""
    from xpysom_dask_amd import XPySom
    g = load_golden("g13_hex_topographic")
    X, D = XD
    key = f"{X}x{X}x{D}"
    seeds = g[key + "_seeds"]
    probe = O.gaussian_blobs(400, D, seed=int(seeds[1]))
    som = XPySom(X, X, D, topology="hexagonal", random_seed=8, decay_function="linear")
    som._weights = g[key + "_w"]
    b1, b2 = som._upload_weights().bmu_top2(probe)
    assert np.array_equal(b1, g[key + "_top2"][:, 0]) and np.array_equal(b2, g[key + "_top2"][:, 1])
    assert som.topographic_error(probe) == float(g[key + "_te"])
    train = O.gaussian_blobs({5: 300, 12: 1500}[X], D, seed=int(seeds[0]))
    assert som.topographic_error(train) == float(g[key + "_te_train"])


@pytest.mark.parametrize("neigh", ["bubble", "triangle"])
def test_other_separable_neighbourhoods(neigh):
    """bubble / triangle (neighborhoods.py:99-130) through the same table-driven transform,
    against a direct restatement of their formulas."""
    X, Y, D, n = 9, 7, 4, 300
    data = O.gaussian_blobs(n, D, seed=9)
    w = O.default_codebook(X, Y, D, 1).astype(F32)
    e = engine(X, Y, D, neighborhood=neigh)
    e.set_weights(w)
    e.set_data(data)
    sig, eta = 3.0, 0.25
    e.epoch_accumulate(sig, eta, False)
    num, den, bmu = e.epoch_fetch()
    ci, cj = bmu // Y, bmu % Y
    ni, nj = np.arange(X)[None, :], np.arange(Y)[None, :]
    if neigh == "bubble":
        ax = ((ni > ci[:, None] - sig) & (ni < ci[:, None] + sig)).astype(np.float64)
        ay = ((nj > cj[:, None] - sig) & (nj < cj[:, None] + sig)).astype(np.float64)
    else:
        ax = np.clip(sig - np.abs(ci[:, None] - ni), 0, None).astype(np.float64)
        ay = np.clip(sig - np.abs(cj[:, None] - nj), 0, None).astype(np.float64)
    g = ax[:, :, None] * ay[:, None, :] * eta
    assert rel_err(den, g.sum(0).reshape(-1)) < 1e-5
    assert rel_err(num, g.reshape(n, -1).T @ data.astype(np.float64)) < 1e-5


# ----------------------------------------------------------------------------- mid-size, both precisions
def test_64x64x32_epoch_f32_and_bf16():
    """BASELINE configs[1] map (64x64x32) on 8192 rows: f32 mode reproduces the oracle's BMUs
    except float32 near-ties and its codebook to 1e-5; bf16 mode may pick a
    different unit only within its operand rounding (every miss is a near-best unit)."""
    X, Y, D, n = 64, 64, 32, 8192
    data = O.gaussian_blobs(n, D, seed=1234)
    w = O.train(data, O.default_codebook(X, Y, D, 1234), 10, sigma0=32.0, decay="linear", iter_end=2)
    sig, eta = 20.0, 0.3
    ref = O.bmu_ids(data, w.reshape(-1, D))
    for precision in ("f32", "bf16"):
        e = engine(X, Y, D, precision=precision)
        e.set_weights(w)
        e.set_data(data)
        e.epoch_accumulate(sig, eta, False)
        num, den, bmu = e.epoch_fetch()
        bad = np.flatnonzero(bmu != ref)
        if precision == "f32":
            assert len(bad) < 20 and near_tie_mask(data[bad], w.reshape(-1, D), tol=1e-5).all()
        else:
            # two epochs at sigma 32 leave neighbouring units nearly identical: many BMUs are
            # decided below bf16 resolution, but every miss must be a near-best unit
            assert len(bad) < 0.5 * n
            assert bf16_misses_are_near_best(data, w.reshape(-1, D), bmu, bad)
        # float32 neighbourhood: the reference itself sums 8192 float32 terms per unit here, so its
        # own rounding error is ~2e-5; the float64 neighbourhood (exponential decay) is the tight check
        _, onum, oden = O.update(data, w, eta, sig, wide=False, forced_bmu=bmu)
        assert rel_err(num, onum.reshape(-1, D)) < 5e-5
        assert rel_err(den, oden.reshape(-1)) < 5e-5
        e.epoch_accumulate_forced(bmu, sig, eta, True)
        num, den, _ = e.epoch_fetch(want_bmu=False)
        _, onum, oden = O.update(data, w, eta, sig, wide=True, forced_bmu=bmu)
        assert rel_err(num, onum.reshape(-1, D)) < 1e-5
        assert rel_err(den, oden.reshape(-1)) < 1e-5


# ----------------------------------------------------------------------------- RCCL plumbing on one GPU
_NCCL_SCRIPT = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["SOM_REPO"])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["SOM_PORT"], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                  SOM_FORCE_ALLREDUCE="1")
import torch, torch.distributed as dist
from oracle import som_oracle as O
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd import distributed as D
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
data = O.gaussian_blobs(500, 6, seed=1)
w = O.default_codebook(9, 8, 6, 2).astype(np.float32)
e = HipEngine(9, 8, 6)
e.set_weights(w); e.set_data(data)
e.epoch_accumulate(3.0, 0.4, True)
num0, den0, _ = e.epoch_fetch()
t = e.accum_tensor()
assert t.is_cuda and t.dtype == torch.float32 and t.numel() == 72 * 8
D.allreduce_accumulator(e)                     # RCCL all-reduce over 1 rank: values unchanged
num1, den1, _ = e.epoch_fetch()
assert np.array_equal(num0, num1) and np.array_equal(den0, den1)
t.mul_(2.0); torch.cuda.synchronize()          # the tensor IS the engine's buffer
num2, den2, _ = e.epoch_fetch()
assert np.array_equal(num2, 2 * num0) and np.array_equal(den2, 2 * den0)
t.mul_(0.5); torch.cuda.synchronize()
D.epoch(e, 3.0, 0.4, True)
_, _, _, want = O.epoch(data, w.reshape(9, 8, 6), 0.4, 3.0, wide=True, n_parallel=500)
assert np.abs(e.get_weights().reshape(9, 8, 6) - want).max() < 1e-5
# the blockwise form (all-reduce of finished 128-row blocks on a second stream, under the next block's transform):
# a 300-row map = three blocks, RCCL on its own stream, same epoch as the monolithic form bit for bit
data = O.gaussian_blobs(3000, 6, seed=3)
w = O.default_codebook(300, 5, 6, 4).astype(np.float32)
outs = []
for overlap in ("0", "1"):
    os.environ["SOM_OVERLAP"] = overlap
    e = HipEngine(300, 5, 6)
    e.set_weights(w); e.set_data(data)
    for sig, eta in ((9.0, 0.5), (4.0, 0.3), (2.0, 0.2)):
        D.epoch(e, sig, eta, True)
    outs.append(e.get_weights())
assert np.array_equal(outs[0], outs[1])
# SOM_COMM=native: the same epochs with the collective inside libsomhip (torch's copy of RCCL, bound with dlopen)
os.environ["SOM_COMM"] = "native"
e = HipEngine(300, 5, 6)
e.set_weights(w); e.set_data(data)
for sig, eta in ((9.0, 0.5), (4.0, 0.3), (2.0, 0.2)):
    D.epoch(e, sig, eta, True)
assert e.has_comm and np.array_equal(e.get_weights(), outs[0])
e.comm_destroy()
dist.destroy_process_group()
print("nccl-path-ok")
"""


_NATIVE_COMM_SCRIPT = r"""
# a caller with nothing but include/somhip.h (no torch in this process): communicator of one rank, epochs through
# som_epoch -- whole-buffer all-reduce on a one-block map, block-by-block on a second stream on a three-block map
import os, sys
import numpy as np
sys.path.insert(0, os.environ["SOM_REPO"])
from oracle import som_oracle as O
from xpysom_dask_amd.engine import HipEngine
assert "torch" not in sys.modules
for (X, Y, D) in ((9, 8, 6), (300, 5, 6)):
    data = O.gaussian_blobs(3000, D, seed=3)
    w = O.default_codebook(X, Y, D, 4).astype(np.float32)
    outs = []
    for native in (False, True):
        e = HipEngine(X, Y, D)
        if native:
            e.comm_init(1, 0, e.comm_unique_id())
        e.set_weights(w); e.set_data(data)
        for sig, eta in ((4.0, 0.5), (2.0, 0.3), (1.0, 0.2)):
            e.epoch(sig, eta, True)
        outs.append(e.get_weights())
        if native:
            e.stream_epoch_accumulate([data[:1000], data[1000:]], 1.0, 0.1, True)
            e.epoch_allreduce(); e.epoch_merge()
            assert np.isfinite(e.get_weights()).all()
            e.comm_destroy()
    assert np.array_equal(outs[0], outs[1])
print("native-comm-ok")
"""


def test_rccl_inside_the_library(tmp_path):
    """som_comm_* (include/somhip.h): the collective behind the C ABI, no torch in the process."""
    import os, subprocess, sys
    from tests.conftest import REPO
    script = tmp_path / "native_comm.py"
    script.write_text(_NATIVE_COMM_SCRIPT)
    env = dict(os.environ, SOM_REPO=REPO, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "native-comm-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_rccl_allreduce_runs_in_place_on_the_engine_buffer(tmp_path):
    """One rank, backend 'nccl' (= RCCL): the fused accumulator is handed to torch.distributed
    zero-copy (one HIP runtime shared by torch and libsomhip) and reduced in place."""
    import os
    import socket
    import subprocess
    import sys
    from tests.conftest import REPO
    script = tmp_path / "nccl_path.py"
    script.write_text(_NCCL_SCRIPT)
    for attempt in range(3):                                  # (a port given back can be taken before the store listens on it)
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ, SOM_REPO=REPO, SOM_PORT=str(port))
        r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
        if "EADDRINUSE" not in r.stderr and "address already in use" not in r.stderr:
            break
    assert r.returncode == 0 and "nccl-path-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bf16_cosine_picks_near_best_similarity():
    """'cosine' in bf16 mode (BASELINE configs[4] semantics at D <= 128): unit-length codebook
    image, argmax of x~.w^~.  Exactly the reference's pick unless two similarities are closer than
    the bf16 operand rounding."""
    X, Y, D, n = 20, 20, 24, 2000
    data = np.abs(O.gaussian_blobs(n, D, seed=21))
    data[7] = 0.0                                             # zero row: all distances 1 -> unit 0
    w = np.abs(O.default_codebook(X, Y, D, 4)).astype(F32) * 2.5
    w[3, 3] = 0.0                                             # zero unit: similarity nan_to_num(0/0) = 0
    ref = O.bmu_ids(data, w.reshape(-1, D), "cosine")
    e = engine(X, Y, D, distance="cosine", precision="bf16")
    e.set_weights(w)
    got = e.bmu(data)
    assert got[7] == ref[7] == 0
    x64, w64 = data.astype(np.float64), w.reshape(-1, D).astype(np.float64)
    with np.errstate(all="ignore"):
        sim = np.nan_to_num((x64 @ w64.T) / np.sqrt((x64 ** 2).sum(1)[:, None] * (w64 ** 2).sum(1)[None, :]))
    chosen = sim[np.arange(n), got]
    assert (chosen >= sim.max(1) - 2.0 ** -7).all()
    assert (got != ref).mean() < 0.2
    # and a full epoch runs through the same update path
    e.set_data(data)
    e.epoch(5.0, 0.5, False)
    assert np.isfinite(e.get_weights()).all()


def test_top2_and_topographic_error():
    from xpysom_dask_amd import XPySom
    g = load_golden("g9_inference")
    probe = O.gaussian_blobs(700, 10, seed=int(g["probe_seed"]))
    som = XPySom(16, 12, 10, random_seed=5, decay_function="linear")
    som._weights = g["w"]
    b1, b2 = som._upload_weights().bmu_top2(probe)
    assert np.array_equal(b1, g["top2"][:, 0]) and np.array_equal(b2, g["top2"][:, 1])
    assert som.topographic_error(probe) == float(g["te"])
    # known answers of the reference's unit test (xpysom_dask/tests.py:81-90)
    som = XPySom(5, 5, 1, std_coeff=1)
    som._weights = np.zeros((5, 5, 1))
    for (i, j), v in {(2, 3): 5.0, (1, 1): 2.0, (2, 4): 6.0, (4, 4): 15.0, (0, 0): 14.0}.items():
        som._weights[i, j] = v
    assert som.topographic_error([[5]]) == 0.0
    assert som.topographic_error([[15]]) == 1.0
    # the generic (input_len > 128) kernel's top-2 against the oracle
    data = O.gaussian_blobs(300, 130, seed=8)
    w = O.default_codebook(9, 7, 130, 2).astype(F32) * 4
    e = engine(9, 7, 130)
    e.set_weights(w)
    a, b = e.bmu_top2(data)
    ref = O.top2_ids(data, w)
    assert np.array_equal(a, ref[:, 0]) and np.array_equal(b, ref[:, 1])


@pytest.mark.parametrize("XY", [(6, 5, 3, 200), (9, 8, 4, 400)])
def test_g10_hexagonal_topology(XY):
    """topology='hexagonal' (gaussian_generic / mexican_hat_generic / bubble) against the reference's
    own _update outputs: the 3-class separable form is exact algebra."""
    g = load_golden("g10_hexagonal")
    X, Y, D, n = XY
    data = O.gaussian_blobs(n, D, seed=300 + X)
    w0 = O.default_codebook(X, Y, D, 77).astype(F32)
    for neigh in ("gaussian", "mexican_hat", "bubble"):
        e = engine(X, Y, D, neighborhood=neigh, topology="hexagonal")
        e.set_weights(w0)
        e.set_data(data)
        for decay in ("linear", "exponential"):
            key = f"{X}x{Y}_{neigh}_{decay}"
            e.epoch_accumulate(float(g[key + "_sig"]), float(g[key + "_eta"]), O.decay_is_wide(decay))
            num, den, bmu = e.epoch_fetch()
            assert np.array_equal(bmu, g[key + "_bmu"])
            assert rel_err(num, g[key + "_num"].reshape(-1, D)) < 1e-5, key
            assert rel_err(den, g[key + "_den"].reshape(-1)) < 1e-5, key
    # compact support on the generic gaussian, against the oracle (pinned bit-exactly by the same fixture)
    e = engine(X, Y, D, neighborhood="gaussian", topology="hexagonal", compact_support=True)
    e.set_weights(w0)
    e.set_data(data)
    e.epoch_accumulate(2.5, 0.3, True)
    num, den, bmu = e.epoch_fetch()
    _, onum, oden = O.update(data, w0, 0.3, 2.5, wide=True, neighbourhood="gaussian_hex", compact=True, forced_bmu=bmu)
    assert rel_err(num, onum.reshape(-1, D)) < 1e-5 and rel_err(den, oden.reshape(-1)) < 1e-5


def test_manhattan_and_norm_p_distances():
    """'manhattan', 'norm_p' (even and odd p), 'norm_p_no_opt': BMUs against the reference's winner()
    (golden) -- bit-exact for p <= 2 (NumPy's exact power fast paths), near-ties excepted above."""
    from xpysom_dask_amd import XPySom
    g = load_golden("g9_inference")
    probe = O.gaussian_blobs(700, 10, seed=int(g["probe_seed"]))
    for name, p in (("manhattan", 1), ("norm_p", 2), ("norm_p_no_opt", 2), ("norm_p", 3), ("norm_p", 4)):
        som = XPySom(16, 12, 10, random_seed=5, activation_distance=name, activation_distance_kwargs={"p": p})
        som._weights = g["w"]
        ids = np.array([i * 12 + j for i, j in som.winner(probe)])
        ref = g["win_%s_p%d" % (name, p)]
        if p <= 2:
            assert np.array_equal(ids, ref), (name, p)
        else:
            assert (ids != ref).sum() <= 2, (name, p)
    # exact-arithmetic binary cases of the reference's distance tests: argmin of every matrix
    gd = load_golden("g2_distances")
    for c in range(int(gd["n_cases"]) - 8):
        x, w = gd[f"c{c:03d}_x"].astype(F32), gd[f"c{c:03d}_w"].astype(F32)
        K, D = w.shape
        for dist, key, p in (("manhattan", "l1", 0), ("norm_p", "p2", 2), ("norm_p", "p3", 3), ("norm_p", "p4", 4)):
            e = engine(K, 1, D, distance=dist, norm_p=p)
            e.set_weights(w)
            assert np.array_equal(e.bmu(x), np.argmin(gd[f"c{c:03d}_{key}"], axis=1)), (c, dist, p)
    # and a training epoch runs through the shared update path
    data = O.gaussian_blobs(400, 6, seed=2)
    som = XPySom(7, 7, 6, random_seed=1, activation_distance="manhattan", decay_function="linear")
    w0 = som._weights.astype(F32)
    som.train(data, 5, iter_beg=0, iter_end=1)
    bmu = O.bmu_ids_pairwise(data, w0.reshape(-1, 6), "manhattan")
    _, onum, oden = O.update(data, w0, O.linear_decay(0.5, 0.01, 0, 5), O.linear_decay(3.5, 1, 0, 5), wide=False,
                             forced_bmu=bmu)
    want = O.merge(w0, onum, oden)
    np.testing.assert_allclose(som._weights, want, rtol=1e-5, atol=1e-6)


def test_activate_distance_from_weights_and_distance_map():
    """The analysis calls that return the (n, K) matrix, with the reference's unit-test answers
    (xpysom_dask/tests.py:66-75,136-143) and the golden distance matrices (bit-exact)."""
    from xpysom_dask_amd import XPySom
    som = XPySom(5, 5, 1, std_coeff=1)
    som._weights = np.zeros((5, 5, 1))
    som._weights[2, 3] = 5.0
    som._weights[1, 1] = 2.0
    assert som.activate(5.0).argmin() == 13
    data = np.arange(-5, 5).reshape(-1, 1)
    d = som.distance_from_weights(data, None)
    wf = som._weights.reshape(-1, 1)
    for i in range(len(data)):
        for j in range(len(wf)):
            assert d[i][j] == np.linalg.norm(data[i] - wf[j])
    som2 = XPySom(2, 2, 2, random_seed=1)
    som2._weights = np.array([[[1., 0.], [0., 1.]], [[1., 0.], [0., 1.]]])
    np.testing.assert_array_equal(som2.distance_map(), np.array([[1., 1.], [1., 1.]]))
    g = load_golden("g2_distances")
    for c in range(int(g["n_cases"])):                           # float64 goldens; float32-exact cases only
        x, w = g[f"c{c:03d}_x"], g[f"c{c:03d}_w"]
        if c >= int(g["n_cases"]) - 8:
            continue
        K, D = w.shape
        for dist, key in (("euclidean", "part"), ("euclidean_no_opt", "sq"), ("cosine", "cos")):
            e = engine(K, 1, D, distance=dist)
            e.set_weights(w.astype(F32))
            got, want = e.distance_matrix(x.astype(F32)), g[f"c{c:03d}_{key}"]
            if key == "cos":                                     # float64 golden: sqrt/divide round differently
                np.testing.assert_allclose(got, want, rtol=0, atol=2e-7)
            else:                                                # small integers: exact in any precision
                np.testing.assert_array_equal(got, want.astype(F32))
        np.testing.assert_allclose(e.distance_matrix(x.astype(F32), quantization=True), g[f"c{c:03d}_l2"], rtol=0, atol=2e-7)
    # float data: bit-identical to NumPy float32 on the host the goldens came from; here to rounding
    rs = np.random.RandomState(3)
    x, w = rs.randn(70, 9).astype(F32), rs.randn(33, 9).astype(F32)
    e = engine(33, 1, 9)
    e.set_weights(w)
    np.testing.assert_allclose(e.distance_matrix(x), O.dist_euclid_part(x, w), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(e.distance_matrix(x, quantization=True), O.dist_euclid(x, w), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("X,Y,D,n", [(12, 11, 200, 700), (6, 7, 784, 300), (20, 13, 130, 1100)])
def test_bf16_large_input_len_tiled_kernel(X, Y, D, n):
    """input_len > 128 in bf16 mode (two-sided tiling, BASELINE configs[4] has 784 features):
    every pick is the float32 BMU or a unit within the bf16 operand rounding of it; the update path
    is the shared exact-f32 one; cosine runs on the unit-length image."""
    data = O.gaussian_blobs(n, D, seed=D)
    w = O.default_codebook(X, Y, D, 9).astype(F32) * 5
    wf = w.reshape(-1, D)
    e = engine(X, Y, D, precision="bf16")
    e.set_weights(w)
    e.set_data(data)
    e.epoch_accumulate(3.0, 0.4, True)
    num, den, bmu = e.epoch_fetch()
    ref = O.bmu_ids(data, wf)
    bad = np.flatnonzero(bmu != ref)
    assert len(bad) <= 0.1 * n
    assert bf16_misses_are_near_best(data, wf, bmu, bad)
    _, onum, oden = O.update(data, w, 0.4, 3.0, wide=True, forced_bmu=bmu)
    assert rel_err(num, onum.reshape(-1, D)) < 1e-5 and rel_err(den, oden.reshape(-1)) < 1e-5
    assert np.array_equal(e.bmu(data[:37]), bmu[:37]) or (e.bmu(data[:37]) != bmu[:37]).sum() <= 2
    # cosine
    pos = np.abs(data)
    wpos = np.abs(w)
    ec = engine(X, Y, D, precision="bf16", distance="cosine")
    ec.set_weights(wpos)
    got = ec.bmu(pos)
    x64, w64 = pos.astype(np.float64), wpos.reshape(-1, D).astype(np.float64)
    sim = (x64 @ w64.T) / np.sqrt((x64 ** 2).sum(1)[:, None] * (w64 ** 2).sum(1)[None, :])
    assert (sim[np.arange(n), got] >= sim.max(1) - 2.0 ** -7).all()


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_streamed_epoch_equals_resident_epoch(precision):
    """Rows handed over in ragged chunks (out-of-core path) give the resident epoch's BMUs, sums and
    codebook: the segment sums are additive over chunks (float32 add order aside)."""
    from xpysom_dask_amd import XPySom
    X, Y, D, n = 14, 9, 20, 5000
    data = O.gaussian_blobs(n, D, seed=4)
    w = O.default_codebook(X, Y, D, 6).astype(F32) * 3
    e = engine(X, Y, D, precision=precision)
    e.set_weights(w)
    e.set_data(data)
    e.epoch_accumulate(3.0, 0.4, True)
    num, den, _ = e.epoch_fetch()
    cuts = [0, 1, 130, 131, 2000, 4999, 5000]
    e2 = engine(X, Y, D, precision=precision)
    e2.set_weights(w)
    e2.stream_epoch_accumulate((data[a:b] for a, b in zip(cuts[:-1], cuts[1:])), 3.0, 0.4, True)
    num2, den2, _ = e2.epoch_fetch(want_bmu=False)
    tol = 2e-6 if precision == "f32" else 5e-3       # bf16: the offset B is per chunk, near-ties may move
    assert rel_err(num2, num) < tol and rel_err(den2, den) < tol
    # the host method, three epochs, against resident training
    a = XPySom(X, Y, D, random_seed=2, decay_function="linear", precision=precision)
    b = XPySom(X, Y, D, random_seed=2, decay_function="linear", precision=precision)
    a.train(data, 3)
    b.train_streaming(lambda: (data[i:i + 777] for i in range(0, n, 777)), 3)
    if precision == "f32":
        np.testing.assert_allclose(b._weights, a._weights, rtol=2e-5, atol=2e-6)
    else:
        assert abs(b.quantization_error(data) - a.quantization_error(data)) < 1e-2 * a.quantization_error(data)


def test_streamed_epoch_from_pinned_double_buffers():
    """Chunks handed over in two alternating PINNED buffers take the asynchronous path (copy on a second
    stream under the previous chunk's kernels) and still give the resident epoch's sums."""
    X, Y, D, n = 16, 16, 24, 9000
    data = O.gaussian_blobs(n, D, seed=14)
    w = O.default_codebook(X, Y, D, 3).astype(F32) * 3
    e = engine(X, Y, D)
    e.set_weights(w)
    e.set_data(data)
    e.epoch_accumulate(4.0, 0.3, True)
    num, den, _ = e.epoch_fetch()
    e2 = engine(X, Y, D)
    e2.set_weights(w)
    bufs = [e2.pinned_empty((1000, D)), e2.pinned_empty((1000, D))]

    def chunks():
        for i, lo in enumerate(range(0, n, 1000)):
            b = bufs[i & 1]
            b[:] = data[lo:lo + 1000]        # reuse is safe: call i+1 has returned before buffer i is rewritten
            yield b
    for _ in range(2):                       # twice: slot reuse across epochs
        e2.stream_epoch_accumulate(chunks(), 4.0, 0.3, True)
        num2, den2, _ = e2.epoch_fetch(want_bmu=False)
        assert rel_err(num2, num) < 2e-6 and rel_err(den2, den) < 2e-6


# ----------------------------------------------------------------------------- retired precisions
def test_the_split_operand_precisions_are_refused():
    """'bf16x3' / 'f16x3' (hi/lo-split operands, ids 2 and 4) are retired: precision='exact' returns float32's own BMUs,
    faster.  The class refuses the names, the library the ids, each with a message that says so."""
    import ctypes as C
    from xpysom_dask_amd import XPySom, _lib
    for name in ("bf16x3", "f16x3"):
        with pytest.raises(ValueError, match="retired"):
            XPySom(8, 8, 4, precision=name)
    lib = _lib.load()
    for pid in (2, 4):
        cfg = _lib.SomConfig(8, 8, 4, 0, 0, 0, pid, 0, 0.5, None, 0, 0, 0.0)
        h = C.c_void_p()
        assert lib.som_create(C.byref(cfg), C.byref(h)) != 0
        assert b"retired" in lib.som_last_error(None)


# ----------------------------------------------------------------------------- opt-in epoch hipGraph
def test_epoch_graph_replay_matches_eager_launches(monkeypatch):
    """SOM_GRAPH=1 replays a captured hipGraph of the whole epoch from the third epoch on; sigma/eta
    travel through device memory.  Same BMUs and accumulators as the eager launches, epoch by epoch
    (teacher-forced from the same codebook, so float-atomic order is the only difference)."""
    X, Y, D, n = 12, 9, 7, 1500
    data = O.gaussian_blobs(n, D, seed=11)
    w0 = O.default_codebook(X, Y, D, 3).astype(F32)
    sched = [(3.0, 0.5), (2.5, 0.45), (2.0, 0.4), (1.5, 0.3), (1.0, 0.2)]
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SOM_GRAPH", mode)
        e = engine(X, Y, D)
        e.set_data(data)
        res = []
        for t, (sig, eta) in enumerate(sched):
            e.set_weights(w0 * (1.0 + 0.1 * t))
            e.epoch_accumulate(sig, eta, t % 2 == 0)
            res.append(e.epoch_fetch())
            e.epoch_merge()
        outs[mode] = res
    for (n0, d0, b0), (n1, d1, b1) in zip(outs["0"], outs["1"]):
        assert np.array_equal(b0, b1)
        np.testing.assert_allclose(n1, n0, rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(d1, d0, rtol=2e-6, atol=1e-6)


# ----------------------------------------------------------------------------- tiled kernel: schedule races
@pytest.mark.parametrize("X,Y,D,n,precision", [(64, 64, 784, 5000, "bf16"), (100, 90, 257, 7000, "bf16"),
                                               (256, 256, 128, 20000, "f16"), (2, 2, 129, 257, "bf16"),
                                               (256, 16, 640, 3333, "bf16"), (64, 64, 900, 3000, "bf16"),
                                               (64, 70, 150, 2500, "f16"), (30, 30, 300, 1000, "bf16"),
                                               (64, 64, 784, 4000, "f16"), (40, 40, 400, 2000, "f16")])
def test_tiled_kernel_is_repeatable_and_near_best(X, Y, D, n, precision):
    """bmu_bf16_tiled_kernel runs two wave groups one barrier apart over a 4-slot LDS-DMA ring with
    counted vmcnt waits; bmu_bf16_wide_kernel (bf16, 128 < input_len <= 800, maps of >= 4096 units: the
    first, second, fifth and seventh case) refills a 3-slot ring behind one barrier per stage.  A mis-ordered read
    or refill would show up as run-to-run differences or as picks outside the operand-rounding bound, so:
    six launches agree bit for bit, the resident-row path (different padding and grid) agrees with the
    query path, and every pick is near-best."""
    rs = np.random.RandomState(n)
    data = O.gaussian_blobs(n, D, seed=n % 97)
    w = (rs.rand(X, Y, D) * 2 - 1).astype(F32) * 2
    e = engine(X, Y, D, precision=precision)
    e.set_weights(w)
    outs = [e.bmu(data) for _ in range(6)]
    for o in outs[1:]:
        assert np.array_equal(outs[0], o)
    e.set_data(data)
    e.epoch_accumulate(2.0, 0.3, True)
    assert np.array_equal(e.epoch_fetch()[2], outs[0])
    idx = rs.choice(n, size=min(n, 1500), replace=False)
    x64, w64 = data[idx].astype(np.float64), w.reshape(-1, D).astype(np.float64)
    dd = np.sqrt(np.maximum((x64 ** 2).sum(1)[:, None] - 2 * x64 @ w64.T + (w64 ** 2).sum(1)[None, :], 0))
    slack = {"bf16": 2.0 ** -8, "f16": 2.0 ** -11}.get(precision, 2.0 ** -15) * (np.linalg.norm(x64, axis=1) + np.linalg.norm(w64, axis=1).max())
    assert (dd[np.arange(len(idx)), outs[0][idx]] <= dd.min(1) + slack).all()


@pytest.mark.parametrize("X,Y,D,n,dist,parts,prec", [
    (64, 64, 129, 1000, "euclidean", None, "bf16"), (64, 70, 200, 777, "cosine", None, "bf16"),
    (70, 70, 257, 2049, "euclidean", "3", "bf16"), (64, 64, 400, 1, "euclidean", None, "bf16"),
    (80, 80, 784, 5001, "cosine", None, "bf16"), (64, 64, 800, 257, "euclidean", "7", "bf16"),
    (72, 64, 540, 256, "euclidean", None, "bf16"),
    (64, 70, 150, 2500, "euclidean", None, "f16"), (64, 64, 266, 700, "cosine", "5", "f16")])
def test_wide_kernel_matches_tiled_kernel(monkeypatch, X, Y, D, n, dist, parts, prec):
    """The two bf16 kernels for input_len > 128 compute the same accumulation chain (same operand images, same
    offset B, features in the same order): bmu_bf16_wide_kernel (samples resident in registers, the default on
    maps of >= 4096 units up to 800 features) must pick what bmu_bf16_tiled_kernel (SOM_BF16_WIDE=0) picks,
    except where two units tie inside the index bits the keys give up (3 of the wide kernel's mantissa
    bits, 4 of the tiled kernel's), and every pick must be near-best.  Row counts off the 256-row block,
    unit counts off the 32-unit stage, forced part counts."""
    data = np.abs(O.gaussian_blobs(n, D, seed=D))
    w = np.abs(O.default_codebook(X, Y, D, 4).astype(F32)) * 3
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SOM_BF16_WIDE", mode)
        if parts and mode == "1":
            monkeypatch.setenv("SOM_BF16_PARTS", parts)
        else:
            monkeypatch.delenv("SOM_BF16_PARTS", raising=False)
        e = engine(X, Y, D, precision=prec, distance=dist)
        e.set_weights(w)
        e.set_data(data)
        e.epoch_accumulate(2.0, 0.3, True)
        got[mode] = (e.epoch_fetch()[2].copy(), e.bmu(data[: min(n, 300)]).copy())
        e.close()
    assert np.array_equal(got["1"][0][: len(got["1"][1])], got["1"][1])
    differ = np.flatnonzero(got["1"][0] != got["0"][0])
    assert len(differ) <= max(1, n // 1000)
    x64, w64 = data.astype(np.float64), w.reshape(-1, D).astype(np.float64)
    if dist == "cosine":
        x64 = x64 / np.linalg.norm(x64, axis=1, keepdims=True)
        w64 = w64 / np.linalg.norm(w64, axis=1, keepdims=True)
    idx = np.arange(n) if n <= 1500 else np.random.RandomState(0).choice(n, 1500, replace=False)
    dd = np.sqrt(np.maximum((x64[idx] ** 2).sum(1)[:, None] - 2 * x64[idx] @ w64.T + (w64 ** 2).sum(1)[None, :], 0))
    slack = (2.0 ** -8 if prec == "bf16" else 2.0 ** -15) * (np.linalg.norm(x64[idx], axis=1) + np.linalg.norm(w64, axis=1).max())
    assert (dd[np.arange(len(idx)), got["1"][0][idx]] <= dd.min(1) + slack).all()


# ----------------------------------------------------------------------------- banded transform
@pytest.mark.parametrize("X,Y,neigh,topo,sigma", [(300, 260, "gaussian", "rectangular", 1.5),
                                                  (300, 260, "mexican_hat", "rectangular", 2.0),
                                                  (257, 300, "gaussian", "hexagonal", 1.2),
                                                  (280, 129, "bubble", "rectangular", 3.0),
                                                  (200, 270, "triangle", "rectangular", 4.0),
                                                  (300, 260, "gaussian", "rectangular", 80.0)])
def test_banded_transform_is_bit_identical(monkeypatch, X, Y, neigh, topo, sigma):
    """Small sigma leaves the neighbourhood tables exact float32 zeros away from the diagonal; the
    transform GEMMs then walk only the nonzero bands (update.hpp).  Skipped chunks would have added
    0 * m, so numerator and denominator must equal the full walk (SOM_NO_BANDS=1) bit for bit."""
    D, n = 5, 4000
    data = O.gaussian_blobs(n, D, seed=X)
    w = O.default_codebook(X, Y, D, 8).astype(F32)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SOM_NO_BANDS", mode)
        e = engine(X, Y, D, neighborhood=neigh, topology=topo)
        e.set_weights(w)
        e.set_data(data)
        e.epoch_accumulate(sigma, 0.4, True)
        num, den, bmu = e.epoch_fetch()
        outs[mode] = (num.copy(), den.copy(), bmu.copy())
        if mode == "0" and neigh == "gaussian" and topo == "rectangular":
            _, onum, oden = O.update(data, w, 0.4, sigma, wide=True, forced_bmu=bmu)
            assert rel_err(num, onum.reshape(-1, D)) < 1e-5 and rel_err(den, oden.reshape(-1)) < 1e-5
    assert np.array_equal(outs["0"][2], outs["1"][2])
    assert np.array_equal(outs["0"][0], outs["1"][0])
    assert np.array_equal(outs["0"][1], outs["1"][1])


@pytest.mark.parametrize("X,Y,n", [(64, 64, 100000), (90, 91, 5000), (8, 8, 2048), (30, 30, 70001)])
def test_counting_sort_by_bmu_equals_the_library_sort(monkeypatch, X, Y, n):
    """Maps of <= 8192 units bucket the rows of an epoch by BMU with the in-tree stable counting sort
    (cs_hist / cs_scan / cs_scatter, update.hpp) instead of rocPRIM's radix sort (SOM_COUNTING_SORT=0).
    Both are stable, the segment sum adds a unit's rows in list order: numerator, denominator and BMUs
    must be bitwise equal -- on clustered rows too, where a handful of units own every row."""
    D = 7
    data = O.gaussian_blobs(n, D, seed=X + n % 97)
    data[: n // 3] = data[0]                                  # a third of the rows on one unit
    w = O.default_codebook(X, Y, D, 4).astype(F32)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SOM_COUNTING_SORT", mode)
        e = engine(X, Y, D)
        e.set_weights(w)
        e.set_data(data)
        e.epoch_accumulate(3.0, 0.4, True)
        num, den, bmu = e.epoch_fetch()
        outs[mode] = (num.copy(), den.copy(), bmu.copy())
    for a, b in zip(outs["0"], outs["1"]):
        assert np.array_equal(a, b)


# ----------------------------------------------------------------------------- rows already in HBM
def test_train_accepts_device_resident_rows():
    """A torch CUDA tensor (what a CuPy array is to the reference, xpysom.py:487-510) is trained on in
    place -- no host round trip -- and gives the codebook the same rows give from the host; analysis
    calls accept it too."""
    import torch
    from xpysom_dask_amd import XPySom
    X, Y, D, n = 12, 10, 6, 3000
    data = O.gaussian_blobs(n, D, seed=21)
    a = XPySom(X, Y, D, sigma=3.0, random_seed=5).train(data, 6)
    t = torch.from_numpy(data).cuda()
    b = XPySom(X, Y, D, sigma=3.0, random_seed=5).train(t, 6)
    np.testing.assert_allclose(b._weights, a._weights, rtol=2e-5, atol=1e-6)
    assert b.winner(t[:50]) == a.winner(data[:50])
    assert abs(b.quantization_error(t) - a.quantization_error(data)) < 1e-5
    t64 = torch.from_numpy(data.astype(np.float64)).cuda()[:, :]       # other dtypes are converted on the device
    c = XPySom(X, Y, D, sigma=3.0, random_seed=5).train(t64, 6)
    np.testing.assert_allclose(c._weights, a._weights, rtol=2e-5, atol=1e-6)
    with pytest.raises(ValueError):
        XPySom(X, Y, D + 1, random_seed=5).train(t, 1)

    # a foreign producer (any __cuda_array_interface__ object, e.g. a CuPy array): rows still being written
    # on ITS stream when train() is called -- the engine waits for that stream -- and analysis calls copy back
    class Foreign:
        def __init__(self, tensor, stream):
            self.keep = tensor
            self.__cuda_array_interface__ = {"shape": tuple(tensor.shape), "typestr": "<f4", "strides": None,
                                             "data": (tensor.data_ptr(), False), "version": 3, "stream": stream}

        def __len__(self):
            return self.keep.shape[0]
    side = torch.cuda.Stream()
    pinned = torch.from_numpy(data).pin_memory()
    with torch.cuda.stream(side):
        big = torch.empty(64 << 20, device="cuda").normal_()   # keeps the side stream busy ahead of the copy
        late = torch.empty((n, D), dtype=torch.float32, device="cuda")
        late.copy_(pinned, non_blocking=True)
    d = XPySom(X, Y, D, sigma=3.0, random_seed=5).train(Foreign(late, side.cuda_stream), 6)
    np.testing.assert_allclose(d._weights, a._weights, rtol=2e-5, atol=1e-6)
    assert abs(d.quantization_error(Foreign(late, None)) - a.quantization_error(data)) < 1e-5
    before = torch.cuda.current_device()
    assert before == 0 and big.numel() > 0


def test_cosine_resolves_short_rows_as_well_as_long_ones():
    """Cosine does not depend on |x|, but the bf16 kernels compare B - x~.w^~ with B = max|x~| max|w~|: a short
    row next to long ones would be resolved at B's absolute precision.  The row images therefore go in at
    unit length (found by tests/fuzz/fuzz_shapes.py: 19x4 map, 2 features, a row of norm 0.1 among norms up to 12)."""
    X, Y, D, n = 19, 4, 2, 3000
    data = np.abs(O.gaussian_blobs(n, D, seed=405))
    data[::7] *= 1e-3                                          # rows four orders of magnitude shorter than the rest
    w = np.abs(O.default_codebook(X, Y, D, 406).astype(F32) * 3)
    x64, w64 = data.astype(np.float64), w.reshape(-1, D).astype(np.float64)
    dd = 1 - (x64 @ w64.T) / np.sqrt((x64 ** 2).sum(1)[:, None] * (w64 ** 2).sum(1)[None, :])
    for precision, tol in (("f16", 2.0 ** -9), ("bf16", 2.0 ** -6)):
        e = engine(X, Y, D, precision=precision, distance="cosine")
        e.set_weights(w)
        got = e.bmu(data)
        gap = dd[np.arange(n), got] - dd.min(1)
        assert gap.max() <= tol, (precision, gap.max())
        assert gap[::7].max() <= tol


def test_support_mask_at_a_sigma_one_ulp_off_an_integer():
    """bubble / compact_support use the reference's mask literally -- n > c - sigma and n < c + sigma in float64
    (neighborhoods.py:29-31, 105-110).  With sigma = 5 / (1 + 2/3) = 3.0000000000000004 (what the asymptotic
    decay yields at t = 1 of 3) c - sigma rounds to an integer and the unit at distance exactly 3 is OUT, whereas
    |n - c| < sigma would let it in (found by tests/fuzz/fuzz_train.py)."""
    sigma = 5.0 / (1.0 + 1.0 / 1.5)
    assert sigma > 3.0 and sigma - 3.0 < 1e-15
    X, Y, D, n = 10, 24, 5, 300
    data = O.gaussian_blobs(n, D, seed=12)
    w = O.default_codebook(X, Y, D, 4).astype(F32)
    for neigh, compact in (("bubble", False), ("gaussian", True), ("triangle", True)):
        e = engine(X, Y, D, neighborhood=neigh, compact_support=compact)
        e.set_weights(w)
        e.set_data(data)
        e.epoch_accumulate(sigma, 0.5, False)
        num, den, bmu = e.epoch_fetch()
        _, onum, oden = O.update(data, w, 0.5, sigma, wide=False, neighbourhood=neigh, compact=compact, forced_bmu=bmu)
        assert rel_err(num, onum.reshape(-1, D)) < 1e-5, neigh
        assert rel_err(den, oden.reshape(-1)) < 1e-5, neigh


# ----------------------------------------------------------------------------- segment sum: fixed order, no atomics
@pytest.mark.parametrize("D", [6, 7, 130])
def test_segment_sum_is_bitwise_reproducible_and_skew_proof(D):
    """The BMU-ordered segment sum forms every unit's sum in ONE order fixed by (N, the BMUs) (csrc/update.hpp):
    two passes over the same state are bitwise equal, whatever the run lengths -- runs inside one 32-row chunk,
    runs cut by one or many chunk boundaries, one unit winning every row, units with no rows."""
    X, Y, n = 9, 8, 20000
    K = X * Y
    data = O.gaussian_blobs(n, D, seed=31)
    w = O.default_codebook(X, Y, D, 3).astype(F32)
    rs = np.random.RandomState(5)
    lengths = [1, 31, 32, 33, 63, 64, 65, 1, 1, 2047, 2048, 2049, 4096 + 17, 3, 700]
    units = rs.permutation(K)[:len(lengths)]
    cut = np.repeat(units, lengths)
    patterns = {
        "one unit wins all": np.full(n, 37),
        "long and short runs": np.concatenate([cut, rs.randint(0, K, size=n - len(cut))]),
        "random": rs.randint(0, K, size=n),
        "two units": np.where(np.arange(n) % 3 == 0, 5, 70),
    }
    e = engine(X, Y, D)
    e.set_weights(w)
    e.set_data(data)
    for name, bmu in patterns.items():
        bmu = rs.permutation(bmu).astype(np.int32) if name == "long and short runs" else bmu.astype(np.int32)
        e.epoch_accumulate_forced(bmu, 2.0, 0.4, True)
        num1, den1, _ = e.epoch_fetch(want_bmu=False)
        e.epoch_accumulate_forced(bmu, 2.0, 0.4, True)
        num2, den2, _ = e.epoch_fetch(want_bmu=False)
        assert np.array_equal(num1, num2) and np.array_equal(den1, den2), name
        _, onum, oden = O.update(data, w, np.float64(0.4), np.float64(2.0), wide=True, forced_bmu=bmu)
        assert rel_err(num1, onum.reshape(-1, D)) < 2e-6, name
        assert rel_err(den1, oden.reshape(-1)) < 2e-6, name
    # the same sums when the rows arrive as streamed chunks (added to the running sums, chunk after chunk)
    bmu = patterns["random"].astype(np.int32)
    e.epoch_accumulate_forced(bmu, 2.0, 0.4, True)
    num1, den1, _ = e.epoch_fetch(want_bmu=False)
    e2 = engine(X, Y, D)
    e2.set_weights(w)
    e2.set_data(data)
    e2.epoch_accumulate(2.0, 0.4, True)
    numr, denr, bmur = e2.epoch_fetch()
    e2.stream_epoch_accumulate([data[:7001], data[7001:7002], data[7002:]], 2.0, 0.4, True)
    nums, dens, _ = e2.epoch_fetch(want_bmu=False)
    assert rel_err(nums, numr) < 2e-6 and rel_err(dens, denr) < 2e-6
    e2.stream_epoch_accumulate([data[:7001], data[7001:7002], data[7002:]], 2.0, 0.4, True)
    nums2, dens2, _ = e2.epoch_fetch(want_bmu=False)
    assert np.array_equal(nums, nums2) and np.array_equal(dens, dens2)


@pytest.mark.parametrize("D", [128, 100, 20])
def test_fused_merge_and_operand_preparation(D, monkeypatch):
    """Round 2's launch-level fusion on the bf16 path -- merge + next epoch's operand preparation in one kernel --
    against the separate launches.  The first epoch
    (operands prepared the same way on both sides) and its merge are bitwise equal; from then on the fused kernel
    sums |w~|^2 in another order than prep_wnorm_kernel, so a few bf16 near-ties may fall the other way."""
    X, Y, n = 20, 24, 5000
    data = O.gaussian_blobs(n, D, seed=77)
    w = O.default_codebook(X, Y, D, 9).astype(F32)
    outs = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("SOM_FUSE_MERGE", fuse)
        e = engine(X, Y, D, precision="bf16")
        e.set_weights(w)
        e.set_data(data)
        trace = []
        for t, (sig, eta) in enumerate([(8.0, 0.5), (4.0, 0.3), (1.5, 0.1), (0.4, 0.05)]):   # 0.4: den == 0 for most units
            e.epoch_accumulate(sig, eta, t % 2 == 0)
            num, den, bmu = e.epoch_fetch()
            e.epoch_merge()
            trace.append((bmu, e.get_weights(), num, den))
        outs.append(trace)
    (b0, w0, n0, d0), (b1, w1, n1, d1) = outs[0][0], outs[1][0]
    assert np.array_equal(b0, b1) and np.array_equal(n0, n1) and np.array_equal(d0, d1) and np.array_equal(w0, w1)
    for (b0, w0, _, _), (b1, w1, _, _) in zip(outs[0][1:], outs[1][1:]):
        assert (b0 != b1).mean() < 0.01
        assert np.abs(w0 - w1).mean() < 2e-3 * np.abs(w0).max()    # (a flipped near-tie moves a small unit's mean)
    # the fused operands alone: same state in, a second engine's BMUs through freshly prepared operands
    monkeypatch.setenv("SOM_FUSE_MERGE", "1")
    e = engine(X, Y, D, precision="bf16")
    e.set_weights(w)
    e.set_data(data)
    e.epoch(8.0, 0.5, True)                                    # merge + fused operand preparation
    e.epoch_accumulate(4.0, 0.3, True)
    _, _, bmu_fused = e.epoch_fetch()
    e2 = engine(X, Y, D, precision="bf16")
    e2.set_weights(outs[1][0][1])                             # the same merged codebook, prepared the ordinary way
    e2.set_data(data)
    e2.epoch_accumulate(4.0, 0.3, True)
    _, _, bmu_plain = e2.epoch_fetch()
    diff = np.flatnonzero(bmu_fused != bmu_plain)
    assert len(diff) <= n // 100 and bf16_misses_are_near_best(data, outs[1][0][1], bmu_fused, diff)


@pytest.mark.parametrize("D,dist", [(784, "cosine"), (200, "euclidean"), (130, "cosine"), (133, "euclidean")])
def test_fused_merge_and_operand_preparation_wide_path(D, dist, monkeypatch):
    """The same fusion on the wide path (bf16, 128 < input_len <= 800, >= 4096 units; merge_prep_wide_kernel), for
    both GEMM-form distances, aligned and unaligned rows: the first epoch and its merge are bitwise equal to the
    separate launches; afterwards the row norms are summed in another order, so a few bf16 near-ties may differ."""
    X, Y, n = 64, 66, 3000
    data = np.abs(O.gaussian_blobs(n, D, seed=3))
    w = np.abs(O.default_codebook(X, Y, D, 9).astype(F32))
    outs = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("SOM_FUSE_MERGE", fuse)
        e = engine(X, Y, D, precision="bf16", distance=dist)
        e.set_weights(w)
        e.set_data(data)
        trace = []
        for t, (sig, eta) in enumerate([(8.0, 0.5), (2.0, 0.3), (0.4, 0.05)]):     # 0.4: den == 0 for most units
            e.epoch_accumulate(sig, eta, t % 2 == 0)
            num, den, bmu = e.epoch_fetch()
            e.epoch_merge()
            trace.append((bmu, e.get_weights(), num, den))
        outs.append(trace)
        e.close()
    (b0, w0, n0, d0), (b1, w1, n1, d1) = outs[0][0], outs[1][0]
    assert np.array_equal(b0, b1) and np.array_equal(n0, n1) and np.array_equal(d0, d1) and np.array_equal(w0, w1)
    assert (outs[1][2][3] == 0).any()                         # the old-weights branch of the fused merge was taken
    for (b0, w0, _, _), (b1, w1, _, _) in zip(outs[0][1:], outs[1][1:]):
        assert (b0 != b1).mean() < 0.01
        assert np.abs(w0 - w1).mean() < 2e-3 * np.abs(w0).max()
    # the fused operands alone against freshly prepared ones on the same merged codebook
    monkeypatch.setenv("SOM_FUSE_MERGE", "1")
    e = engine(X, Y, D, precision="bf16", distance=dist)
    e.set_weights(w)
    e.set_data(data)
    e.epoch(8.0, 0.5, True)
    e.epoch_accumulate(2.0, 0.3, True)
    _, _, bmu_fused = e.epoch_fetch()
    e2 = engine(X, Y, D, precision="bf16", distance=dist)
    e2.set_weights(outs[1][0][1])
    e2.set_data(data)
    e2.epoch_accumulate(2.0, 0.3, True)
    _, _, bmu_plain = e2.epoch_fetch()
    assert (bmu_fused != bmu_plain).sum() <= n // 100


def test_staged_epoch_equals_the_monolithic_one():
    """som_epoch_accumulate_begin + one som_epoch_accumulate_block per 128-row map block (the form the overlapped
    all-reduce uses) leaves the accumulator of som_epoch_accumulate, bit for bit, and reports disjoint slices that
    tile it."""
    for (X, Y, D, neigh, topo) in ((300, 40, 9, "gaussian", "rectangular"), (129, 130, 20, "mexican_hat", "hexagonal"), (20, 24, 128, "gaussian", "rectangular")):
        n = 4000
        data = O.gaussian_blobs(n, D, seed=5)
        w = O.default_codebook(X, Y, D, 6).astype(F32)
        e = engine(X, Y, D, neighborhood=neigh, topology=topo)
        e.set_weights(w)
        e.set_data(data)
        e.epoch_accumulate(3.0, 0.4, True)
        num0, den0, bmu0 = e.epoch_fetch()
        e.epoch_accumulate_begin(3.0, 0.4, True)
        nb = e.epoch_block_count()
        assert nb == -(-X // 128)
        at = 0
        for b in range(nb):
            off, cnt = e.epoch_accumulate_block(b)
            assert off == at and cnt > 0
            at += cnt
        assert at == X * Y * (-(-(D + 1) // 4) * 4)
        num1, den1, bmu1 = e.epoch_fetch()
        assert np.array_equal(bmu0, bmu1) and np.array_equal(num0, num1) and np.array_equal(den0, den1)
        from xpysom_dask_amd.engine import SomHipError
        with pytest.raises(SomHipError):
            e.epoch_accumulate_block(0)                        # no epoch in progress


@pytest.mark.parametrize("precision,shape", [("f32", (4, 5, 6)), ("bf16", (4, 5, 6)), ("bf16", (64, 65, 200)),
                                             ("bf16", (5, 5, 200)), ("exact", (4, 5, 6)), ("exact", (64, 64, 16)), ("f32", (6, 6, 150)),
                                             ("f16", (4, 5, 6)), ("f16", (64, 65, 200))])
def test_nan_semantics_documented_in_design(precision, shape):
    """DESIGN.md 4, known difference: a unit whose distance is NaN never wins (`<` semantics), where numpy.argmin
    returns the FIRST NaN unit; a row whose distances are ALL NaN returns unit 0, as numpy.argmin does.  Only a
    codebook (or a row) that already holds NaN can get there; this pins what the engine does -- in every BMU kernel
    (resident, wide and tiled bf16 forms, the exact mode, both float32 forms)."""
    X, Y, D = shape
    rs = np.random.RandomState(2)
    w = rs.randn(X * Y, D).astype(F32)
    x = rs.randn(50, D).astype(F32)
    e = engine(X, Y, D, precision=precision)
    e.set_weights(w)
    clean = e.bmu(x)
    wn = w.copy()
    wn[3] = np.nan                                              # one NaN unit: it never wins, the others keep their order
    e.set_weights(wn)
    got = e.bmu(x)
    assert (got != 3).all()
    keep = clean != 3
    if precision == "f32":
        assert np.array_equal(got[keep], clean[keep])
    else:                                                       # (bf16: the offset B = max|x~| max|w~| ignores NaN norms)
        assert (got[keep] == clean[keep]).mean() > 0.9
    xn = x.copy()
    xn[7] = np.nan                                              # an all-NaN row: unit 0
    e.set_weights(w)
    got = e.bmu(xn)
    assert got[7] == 0 and np.array_equal(np.delete(got, 7), np.delete(clean, 7))


@pytest.mark.parametrize("neigh", ["gaussian", "mexican_hat"])
def test_g16_hexagonal_compact_support_at_a_lattice_sigma_against_the_reference(neigh):
    """G16: the reference's own `_update` at sigma = 3.0000000000000004 on the hexagonal topology with
    compact_support (the four-parity-class tables, update.hpp)."""
    g = load_golden("g16_hex_compact_lattice_sigma")
    X, Y, D, n = 10, 12, 16, 200
    data = O.gaussian_blobs(n, D, seed=1131)
    w0 = O.default_codebook(X, Y, D, 131).astype(F32)
    e = engine(X, Y, D, neighborhood=neigh, compact_support=True, topology="hexagonal", std_coeff=1.0)
    e.set_weights(w0)
    e.set_data(data)
    e.epoch_accumulate(float(g["sigma"]), float(g[neigh + "_eta"]), O.decay_is_wide("asymptotic"))
    num, den, bmu = e.epoch_fetch()
    assert np.array_equal(bmu, g[neigh + "_bmu"])
    assert rel_err(num, g[neigh + "_num"].reshape(-1, D)) < 1e-5
    assert rel_err(den, g[neigh + "_den"].reshape(-1)) < 1e-5
    # the faithful K x N x D form agrees (it generates g from the same tables)
    e.epoch_accumulate_faithful(float(g["sigma"]), float(g[neigh + "_eta"]), O.decay_is_wide("asymptotic"))
    num2, den2, _ = e.epoch_fetch(want_bmu=False)
    assert rel_err(num2, g[neigh + "_num"].reshape(-1, D)) < 1e-5 and rel_err(den2, g[neigh + "_den"].reshape(-1)) < 1e-5


@pytest.mark.parametrize("neigh", ["mexican_hat", "gaussian"])
def test_hexagonal_compact_support_mask_at_a_sigma_one_ulp_off_the_lattice(neigh):
    """asymptotic_decay(5, ., 1, 3) = 5 / (1 + 2/3) = 3.0000000000000004: the reference's mask `nx < cx + sigma`
    rounds cx + sigma first, so whether the unit exactly 3.0 away in x is inside depends on cx -- on the BMU's absolute
    coordinate, half-unit row offset included (neighborhoods.py:50-54, :91-93).  The hexagonal factor tables therefore
    keep four parity classes under compact_support (update.hpp); a mask on the coordinate difference alone gets the
    boundary units of this epoch wrong (found by tests/fuzz/fuzz_train.py, seed 9 case 131)."""
    from xpysom_dask_amd import XPySom
    X, Y, D, n, T, t_at = 10, 12, 16, 200, 3, 1
    data = O.gaussian_blobs(n, D, seed=1131)
    som = XPySom(X, Y, D, sigma=5.0, learning_rate=0.5, decay_function="asymptotic", neighborhood_function=neigh,
                 topology="hexagonal", random_seed=131, compact_support=True, std_coeff=1.0)
    w0 = som._weights.copy()
    ids = som._upload_weights().bmu(data)
    som.train(data, T, iter_beg=t_at, iter_end=t_at + 1)
    f = O.DECAYS["asymptotic"]
    sig_t, eta_t = f(5.0, 1, t_at, T), f(0.5, 0.01, t_at, T)
    assert sig_t != 3.0 and abs(sig_t - 3.0) < 1e-15
    _, _, den, want = O.epoch(data, w0.astype(F32), eta_t, sig_t, wide=O.decay_is_wide("asymptotic"), n_parallel=n,
                              forced_bmu=ids, compact=True, std_coeff=1.0, neighbourhood=neigh + "_hex")
    live = np.abs(den[..., 0]) > 1e-3 * np.abs(den).max()
    assert live.sum() > X * Y // 4
    err = np.abs(som._weights[live] - want[live]).max() / np.abs(want[live]).max()
    assert err < 5e-5


@pytest.mark.parametrize("topo,shape", [("rectangular", (9, 9, 4, 400)), ("hexagonal", (9, 8, 4, 400)), ("hexagonal", (7, 7, 3, 300))])
def test_g15_mexican_hat_with_compact_support_as_the_reference_computes_it(topo, shape):
    """neighborhoods.py:69-71 / :91-93: px masked twice (on the rectangular topology the second mask compares the
    row index with the BMU's column), py never.  Reproduced: four separable terms on the hexagonal grid, row stage ->
    mask -> column stage on the rectangular one; against the reference's own _update outputs."""
    g = load_golden("g15_mexican_compact")
    X, Y, D, n = shape
    data = O.gaussian_blobs(n, D, seed=600 + X + Y)
    w0 = O.default_codebook(X, Y, D, 41).astype(F32)
    e = engine(X, Y, D, neighborhood="mexican_hat", compact_support=True, topology=topo)
    e.set_weights(w0)
    e.set_data(data)
    for decay in ("linear", "exponential"):
        key = f"{topo}_{X}x{Y}x{D}_{decay}"
        e.epoch_accumulate(float(g[key + "_sig"]), float(g[key + "_eta"]), O.decay_is_wide(decay))
        num, den, bmu = e.epoch_fetch()
        assert np.array_equal(bmu, g[key + "_bmu"])
        assert rel_err(num, g[key + "_num"].reshape(-1, D)) < 1e-5, key
        assert rel_err(den, g[key + "_den"].reshape(-1)) < 1e-5, key
        # the staged form (blocks of map rows) gives the same accumulator
        e.epoch_accumulate_begin(float(g[key + "_sig"]), float(g[key + "_eta"]), O.decay_is_wide(decay))
        for b in range(e.epoch_block_count()):
            e.epoch_accumulate_block(b)
        num2, den2, _ = e.epoch_fetch(want_bmu=False)
        assert np.array_equal(num, num2) and np.array_equal(den, den2)
    if topo == "rectangular":
        from xpysom_dask_amd.engine import SomHipError
        with pytest.raises(SomHipError, match="square map"):
            engine(5, 7, 3, neighborhood="mexican_hat", compact_support=True)
    # a wider square map: more than one 128-row block in both stages
    X = Y = 130
    data = O.gaussian_blobs(500, 5, seed=9)
    w0 = O.default_codebook(X, Y, 5, 3).astype(F32)
    e = engine(X, Y, 5, neighborhood="mexican_hat", compact_support=True, topology=topo)
    e.set_weights(w0)
    e.set_data(data)
    e.epoch_accumulate(3.0, 0.4, True)
    num, den, bmu = e.epoch_fetch()
    _, onum, oden = O.update(data, w0, np.float64(0.4), np.float64(3.0), wide=True, compact=True, forced_bmu=bmu,
                             neighbourhood="mexican_hat" + ("_hex" if topo == "hexagonal" else ""))
    assert rel_err(num, onum.reshape(-1, 5)) < 1e-5 and rel_err(den, oden.reshape(-1)) < 1e-5


@pytest.mark.parametrize("cfg", [
    dict(X=20, Y=30, D=12, n=3000, neighborhood="gaussian"),
    dict(X=9, Y=8, D=4, n=400, neighborhood="mexican_hat", topology="hexagonal"),
    dict(X=13, Y=11, D=130, n=700, neighborhood="triangle", compact_support=True),
    dict(X=150, Y=5, D=7, n=1000, neighborhood="bubble"),
    dict(X=7, Y=7, D=3, n=300, neighborhood="mexican_hat", topology="hexagonal", compact_support=True),
])
def test_faithful_update_form_equals_the_bucketed_one(cfg):
    """The update as the reference states it -- g[n, k] generated per (sample, unit) inside a K x N x D MFMA GEMM,
    num = g^T x, den = sum_n g (xpysom.py:434-441; som_epoch_accumulate_faithful) -- against the bucketed algebra
    the engine trains with (segment sums + separable transform): same accumulator to float32 summation order."""
    cfg = dict(cfg)
    X, Y, D, n = (cfg.pop(k) for k in ("X", "Y", "D", "n"))
    data = O.gaussian_blobs(n, D, seed=X + D)
    w = O.default_codebook(X, Y, D, 2).astype(F32)
    e = engine(X, Y, D, **cfg)
    e.set_weights(w)
    e.set_data(data)
    for sig, eta, wide in ((3.0, 0.4, True), (1.2, 0.1, False)):
        e.epoch_accumulate(sig, eta, wide)
        num0, den0, bmu0 = e.epoch_fetch()
        e.epoch_accumulate_faithful(sig, eta, wide)
        num1, den1, bmu1 = e.epoch_fetch()
        assert np.array_equal(bmu0, bmu1)
        assert rel_err(num1, num0) < 1e-5 and rel_err(den1, den0) < 1e-5
    from xpysom_dask_amd.engine import SomHipError
    if cfg.get("neighborhood") == "gaussian":
        sq = engine(5, 5, 3, neighborhood="mexican_hat", compact_support=True)
        sq.set_weights(O.default_codebook(5, 5, 3, 1).astype(F32))
        sq.set_data(O.gaussian_blobs(50, 3, seed=1))
        with pytest.raises(SomHipError, match="not a sum of row factor"):
            sq.epoch_accumulate_faithful(1.0, 0.1, True)


def test_linear_schedule_ending_at_sigma_zero_with_mexican_hat_raises_like_the_reference():
    """The reference's last epoch evaluates 2/d with d = 0.0 (a Python float) and raises; the engine would happily
    compute infinities -- the host raises the same error before the launch."""
    from xpysom_dask_amd import XPySom
    data = O.gaussian_blobs(50, 3, seed=1)
    som = XPySom(5, 5, 3, sigma=1.0, sigmaN=0, decay_function="linear", neighborhood_function="mexican_hat", random_seed=1)
    with pytest.raises(ZeroDivisionError, match="float division by zero"):
        som.train(data, 2)
    som = XPySom(5, 5, 3, sigma=1.0, sigmaN=0, decay_function="linear", neighborhood_function="bubble", random_seed=1)
    som.train(data, 2)                                             # bubble at sigma 0: nobody inside, weights unchanged
    assert np.isfinite(som._weights).all()



def test_winner_on_float64_rows_uses_float64_arithmetic():
    """xpysom.py:379-396 does not coerce x: float64 rows against trained float32 weights are scored in float64 by NumPy.  The
    reference's own answer (G9 `winner64`) must come back, in the float32 mode and in the exact mode; and on rows built to
    make the two arithmetics disagree, the float64 path agrees with a float64 evaluation where the float32 one does not."""
    from xpysom_dask_amd import XPySom
    g = load_golden("g9_inference")
    probe = O.gaussian_blobs(700, 10, seed=int(g["probe_seed"]))
    for precision in ("f32", "exact"):
        som = XPySom(16, 12, 10, random_seed=5, precision=precision)
        som._weights = g["w"]
        ids = np.array([i * 12 + j for i, j in som.winner(probe.astype(np.float64))])
        assert np.array_equal(ids, g["winner64"]), precision
        assert som.winner(probe[3].astype(np.float64)) == tuple(int(v) for v in divmod(int(g["winner64"][3]), 12))
    # near-ties only float64 resolves: two units at almost the same distance from x
    rs = np.random.RandomState(2)
    w = rs.randn(8, 8, 6).astype(F32)
    x = ((w[2, 3].astype(np.float64) + w[5, 1].astype(np.float64)) / 2)[None, :] + 1e-9 * rs.randn(200, 6)
    som = XPySom(8, 8, 6, random_seed=1)
    som._weights = w
    ids64 = np.array([i * 8 + j for i, j in som.winner(x)])
    wf = w.reshape(-1, 6).astype(np.float64)
    want = np.argmin(-2 * x @ wf.T + (w.reshape(-1, 6) ** 2).sum(1).astype(np.float64)[None, :], axis=1)
    assert np.array_equal(ids64, want)


def test_g19_norm_p_with_a_real_exponent():
    """`norm_p` with p = 0.5, 1.5, 2.5, 3.7 (distances.py:61-75): the reference's winners; a pick may differ from the
    reference's only on a float32 near-tie (the device's pow is float64 pow rounded once, glibc's powf is < 1 ulp)."""
    from xpysom_dask_amd import XPySom
    g = load_golden("g19_norm_p_real")
    g9 = load_golden("g9_inference")
    probe = O.gaussian_blobs(700, 10, seed=int(g["probe_seed"]))
    w = g9["w"].reshape(-1, 10).astype(np.float64)
    for p in (0.5, 1.5, 2.5, 3.7):
        tag = str(p).replace(".", "_")
        for name in ("norm_p", "norm_p_no_opt"):
            som = XPySom(16, 12, 10, random_seed=5, activation_distance=name, activation_distance_kwargs={"p": p})
            som._weights = g9["w"]
            ids = np.array([i * 12 + j for i, j in som.winner(probe)])
            ref = g["win_%s_p%s" % (name, tag)]
            diff = np.flatnonzero(ids != ref)
            assert len(diff) <= 2, (name, p, len(diff))
            if len(diff):
                d = (np.abs(probe[diff].astype(np.float64)[:, None, :] - w[None, :, :]) ** p).sum(2)
                gap = d[np.arange(len(diff)), ids[diff]] - d.min(1)
                assert (gap <= 1e-5 * d.min(1)).all(), (name, p, gap)


# ----------------------------------------------------------------------------- G17 / G18: wide shapes pinned by the reference
@pytest.mark.parametrize("decay", ["linear", "exponential"])
@pytest.mark.parametrize("precision", ["f32", "exact", "bf16"])
def test_g17_configs4_semantics_at_a_wide_kernel_shape(decay, precision):
    """cosine + mexican_hat, 784 features, non-negative unit rows (BASELINE configs[4]) on a 64 x 64 map: the reference's
    own BMUs, denominator, strided numerator and merged rows (distances.py:45-59, neighborhoods.py:57-74).  With 784
    features the reference's sgemm splits K into blocks (one k-ordered chain only up to 448, measured on the generating
    host), so float32 near-ties may fall the other way: counted, bounded, and each one checked to BE a near-tie."""
    g = load_golden("g17_configs4_64x64x784")
    X, Y, D, n = (int(v) for v in g["shape"])
    st = int(g["stride"])
    data = np.abs(O.gaussian_blobs(n, D, seed=int(g["data_seed"])))
    data /= np.linalg.norm(data, axis=1, keepdims=True)
    data = data.astype(F32)
    w = np.abs(O.default_codebook(X, Y, D, 1234)).astype(F32)
    wide = O.decay_is_wide(decay)
    e = engine(X, Y, D, precision=precision, distance="cosine", neighborhood="mexican_hat")
    e.set_weights(w)
    e.set_data(data)
    e.epoch_accumulate(float(g[decay + "_sig"]), float(g[decay + "_eta"]), wide)
    num, den, bmu = e.epoch_fetch()
    ref = g[decay + "_bmu"]
    diff = np.flatnonzero(bmu != ref)
    x64, w64 = data.astype(np.float64), w.reshape(-1, D).astype(np.float64)
    cosd = 1.0 - (x64 @ w64.T) / np.sqrt((x64 ** 2).sum(1)[:, None] * (w64 ** 2).sum(1)[None, :])
    gap = cosd[diff, bmu[diff]] - cosd[diff].min(1)         # how much worse than the best unit the engine's pick is
    tol = {"f32": 2e-6, "exact": 2e-6, "bf16": 2.0 ** -6}[precision]
    assert (gap <= tol).all(), (precision, gap.max())
    assert len(diff) <= {"f32": max(2, n // 500), "exact": max(2, n // 500), "bf16": n // 4}[precision], len(diff)
    if precision == "exact":                                 # (the wide screen served, not the float32 fallback)
        rows, fb, _ = e.exact_stats()
        assert rows == n and fb <= n // 50
    # the update from the engine's own BMUs against the oracle, elementwise
    sig = np.float64(g[decay + "_sig"]) if wide else float(g[decay + "_sig"])
    eta = np.float64(g[decay + "_eta"]) if wide else float(g[decay + "_eta"])
    _, onum, oden = O.update(data, w, eta, sig, wide=wide, forced_bmu=bmu, neighbourhood="mexican_hat", distance="cosine")
    assert rel_err(num, onum.reshape(-1, D)) < 1e-5 and rel_err(den, oden.reshape(-1)) < 1e-5
    if precision == "f32":
        (LOST if len(diff) else COMPARED).append(("g17", decay))
    if len(diff):
        return
    gden = g[decay + "_den"].reshape(-1)                     # same BMUs: the reference's own accumulators
    assert rel_err(den, gden) < 1e-5
    assert rel_err(num[::st], g[decay + "_num32"]) < 1e-5
    e.epoch_merge()
    gw = g[decay + "_wout32"]
    ok = np.abs(gden[::st]) > 1e-3 * np.abs(gden).max()      # (mexican hat: a denominator near zero amplifies everything)
    np.testing.assert_allclose(e.get_weights()[::st][ok], gw[ok], rtol=2e-4, atol=1e-5 * np.abs(gw).max())


def test_g17_euclidean_bmus_at_784_features():
    g = load_golden("g17_configs4_64x64x784")
    X, Y, D, n = (int(v) for v in g["shape"])
    data = np.abs(O.gaussian_blobs(n, D, seed=int(g["data_seed"])))
    data /= np.linalg.norm(data, axis=1, keepdims=True)
    data = data.astype(F32)
    w = np.abs(O.default_codebook(X, Y, D, 1234)).astype(F32)
    for precision in ("f32", "exact"):
        e = engine(X, Y, D, precision=precision)
        e.set_weights(w)
        ids = e.bmu(data)
        diff = np.flatnonzero(ids != g["euclidean_bmu"])
        assert len(diff) <= max(2, n // 500) and near_tie_mask(data[diff], w.reshape(-1, D)).all(), (precision, len(diff))
        e.close()


@pytest.mark.parametrize("state", ["seeded", "sheet"])
def test_g18_bmus_at_the_configs2_shape(state):
    """256 x 256 x 128, 4 096 rows: the reference's `_winner` on the seeded codebook and on a smooth sheet (the state of
    the early schedule: hundreds of near-best units per row).  Up to 448 features the reference's sgemm IS one
    k-ordered fma chain per output, which is what the float32 MFMA kernel computes: ids must be IDENTICAL, in the
    float32 mode and in the exact mode; the throughput modes are held to their bounds."""
    import zlib
    g = load_golden("g18_bmus_256x256x128")
    X, Y, D, n = (int(v) for v in g["shape"])
    data = O.gaussian_blobs(n, D, seed=int(g["data_seed"]))
    if state == "seeded":
        w = O.default_codebook(X, Y, D, int(g["codebook_seed"])).astype(F32)
    else:
        w = O.smooth_sheet_codebook(X, Y, D, int(g["sheet_seed"]), amplitude=float(g["sheet_amplitude"]),
                                    centre=data.astype(np.float64).mean(0))
    assert zlib.crc32(np.ascontiguousarray(w).tobytes()) == int(g[state + "_w_crc"]), "the codebook recipe left the fixture's"
    ref = g[state + "_bmu"]
    x64, w64 = data.astype(np.float64), w.reshape(-1, D).astype(np.float64)
    for precision in ("f32", "exact", "f16", "bf16"):
        e = engine(X, Y, D, precision=precision)
        e.set_weights(w)
        ids = e.bmu(data)
        e.close()
        if precision in ("f32", "exact"):
            assert np.array_equal(ids, ref), (precision, int((ids != ref).sum()))
            continue
        diff = np.flatnonzero(ids != ref)
        got = ((x64[diff] - w64[ids[diff]]) ** 2).sum(1)
        best = ((x64[diff] - w64[ref[diff]]) ** 2).sum(1)
        slack = {"f16": 2.0 ** -9, "bf16": 2.0 ** -6}[precision] * \
            (np.linalg.norm(x64[diff], axis=1) + np.linalg.norm(w64, axis=1).max()) ** 2
        assert (got <= best + slack).all(), precision


# ----------------------------------------------------------------------------- G20: two consecutive epochs on resident rows
@pytest.mark.parametrize("case", ["a", "b"])
@pytest.mark.parametrize("precision", ["f32", "exact"])
def test_g20_two_resident_epochs_against_the_reference(case, precision):
    """The exact mode's SECOND-epoch path -- rows visited in the order of last epoch's BMU patch, a plan that proves blocks
    of the distance GEMM empty and skips them, seeds from last epoch's BMUs (csrc/exact_skip.hpp) -- engages only on rows
    that stay resident; every other golden compares a first epoch.  Here the reference ran two consecutive teacher-forced
    epochs, train(.., iter_beg=t, iter_end=t+1) from W_t and again from W_{t+1} (xpysom.py:458,481-482,515-577): the rows
    stay resident, epoch t runs (ids compared), W_{t+1} is set, epoch t+1 runs WITH the plan engaged (skipped blocks
    asserted) and its BMUs, denominator, strided numerator and merged rows are the reference's.
    a) 64 x 64 x 32: the reference's own trained states; b) 256 x 256 x 128 on seeded sheets (crc-checked recipes)."""
    import zlib
    g = load_golden("g20_two_resident_epochs")
    X, Y, D, n = (int(v) for v in g[case + "_shape"])
    st = int(g[case + "_stride"])
    if case == "a":
        data = O.gaussian_blobs(n, D, seed=int(g["a_data_seed"]))
        ws = [g["a_w0"], g["a_w1"]]
    else:
        s0, s1, s2 = (int(v) for v in g["b_seeds"])
        w0 = O.smooth_sheet_codebook(X, Y, D, s0, amplitude=float(g["b_amplitude"]))
        w1 = O.sheet_step(w0, O.smooth_sheet_codebook(X, Y, D, s1, amplitude=float(g["b_amplitude"])), float(g["b_mix"]))
        gen = O.rows_on_codebook(w0, n + 256, s2, float(g["b_noise"]))
        # (drawn with 256 spare rows; the generator dropped the float32 near-ties -- top-2 gap below 4e-6 -- it lists)
        data = np.ascontiguousarray(gen[np.setdiff1d(np.arange(len(gen)), g["b_dropped"])[:n]])
        ws = [w0, w1]
        for i, w in enumerate(ws):
            assert zlib.crc32(np.ascontiguousarray(w).tobytes()) == int(g["b_w%d_crc" % i]), "the codebook recipe left the fixture's"
        assert zlib.crc32(np.ascontiguousarray(data).tobytes()) == int(g["b_data_crc"]), "the row recipe left the fixture's"
    e = engine(X, Y, D, precision=precision)
    e.set_data(data)                                           # resident across both epochs
    for i, w in enumerate(ws):
        key = "%s_e%d" % (case, i)
        e.set_weights(w)                                       # teacher-forced: the reference's own state
        r0, t0 = e.exact_skip_stats() if precision == "exact" else (0, 0)
        e.epoch_accumulate(float(g[key + "_sig"]), float(g[key + "_eta"]), True)     # (exponential decay: float64 neighbourhood)
        num, den, bmu = e.epoch_fetch()
        if precision == "exact":
            r1, t1 = e.exact_skip_stats()
            assert t1 > t0
            if i == 0:
                assert r1 - r0 == t1 - t0                      # first epoch on fresh rows: nothing to plan from
            else:
                assert r1 - r0 < 0.6 * (t1 - t0), "the second epoch did not skip: %d of %d blocks run" % (r1 - r0, t1 - t0)
        ref = g[key + "_bmu"]
        diff = np.flatnonzero(bmu != ref)
        if case == "b":
            assert len(diff) == 0, (precision, len(diff))      # one k-ordered fma chain per output on both sides: IDENTICAL ids
        elif len(diff):
            assert len(diff) <= max(2, n // 500) and near_tie_mask(data[diff], w.reshape(-1, D)).all()
        if precision == "f32":
            (LOST if len(diff) else COMPARED).append(("g20", case, i))
        if len(diff):
            continue
        gden = g[key + "_den"].reshape(-1)
        ok = gden > 1e-30
        np.testing.assert_allclose(den[ok], gden[ok], rtol=1e-5)
        assert rel_err(num[::st], g[key + "_num"]) < 1e-5
        e.epoch_merge()
        gw = g[key + "_wout"]
        np.testing.assert_allclose(e.get_weights()[::st][ok[::st]], gw[ok[::st]], rtol=1e-5, atol=1e-5 * np.abs(gw).max())
    if precision == "exact":
        rows, fb, _ = e.exact_stats()
        assert rows == 2 * n and fb <= n // 100
    e.close()


def test_zz_few_golden_comparisons_were_lost_to_near_ties():
    """The float32 mode's BMUs equal the reference's except on float32 near-ties (G12's one row; G17's K-split sgemm):
    the comparisons of accumulators / merged rows that such a difference makes meaningless are counted here."""
    assert len(COMPARED) >= 20, (len(COMPARED), LOST)
    assert len(LOST) <= 3, LOST
