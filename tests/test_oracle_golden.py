"""The NumPy oracle against the golden vectors captured from the reference
(oracle/make_golden.py).  CPU only.  This is what 'pins' the oracle."""
import numpy as np
import pytest

from oracle import som_oracle as O
from tests.conftest import load_golden

F32 = np.float32


def test_g1_ties_first_minimum_wins():
    g = load_golden("g1_ties")
    x, w = g["x"], g["w"]
    assert np.array_equal(O.winner_ids(x, w), g["ids"])
    assert np.array_equal(O.winner_ids(x, np.zeros_like(w)), g["ids_zero"])
    assert np.array_equal(O.winner_ids(x, np.ones_like(w)), g["ids_same"])
    assert (g["ids_zero"] == 0).all() and (g["ids_same"] == 0).all()
    # duplicated rows 7, 10, 19 -> 7
    assert g["ids"][0] == 7 and g["ids"][2] == 7 and g["ids"][3] == 7


def test_g2_distances():
    g = load_golden("g2_distances")
    n = int(g["n_cases"])
    assert n > 100
    for c in range(n):
        x, w = g[f"c{c:03d}_x"], g[f"c{c:03d}_w"]
        with np.errstate(all="ignore"):
            np.testing.assert_array_equal(O.dist_euclid_part(x, w), g[f"c{c:03d}_part"])
            np.testing.assert_array_equal(O.dist_euclid_sq(x, w), g[f"c{c:03d}_sq"])
            np.testing.assert_array_equal(O.dist_euclid(x, w), g[f"c{c:03d}_l2"])
            np.testing.assert_array_equal(O.dist_cosine(x, w), g[f"c{c:03d}_cos"])


def test_g2_known_answers_of_the_reference_tests():
    """The closed forms the reference's own test file checks against
    (xpysom_dask/test_distances.py:92-113), to 7 decimals as it does."""
    g = load_golden("g2_distances")
    for c in range(int(g["n_cases"])):
        x, w = g[f"c{c:03d}_x"], g[f"c{c:03d}_w"]
        for i, vx in enumerate(x):
            for j, vy in enumerate(w):
                assert abs(g[f"c{c:03d}_part"][i, j] - (-2 * vx @ vy + vy @ vy)) < 1e-7
                assert abs(g[f"c{c:03d}_l2"][i, j] - np.linalg.norm(vx - vy)) < 1e-7


@pytest.mark.parametrize("XY", [(5, 5), (3, 4)])
def test_g3_neighbourhoods(XY):
    g = load_golden("g3_neighbourhoods")
    X, Y = XY
    ci, cj = np.divmod(np.arange(X * Y), Y)
    for sig in (0.3, 1.0, 2.5):
        for wide in (False, True):
            tag = "f64" if wide else "f32"
            for compact in (False, True):
                key = f"{X}x{Y}_s{sig}_{tag}_{'cs' if compact else 'nc'}"
                for name, sc in (("gauss_", 1.0), ("gauss05_", 0.5)):
                    ref = g[name + key]
                    got = O.neigh_gaussian(X, Y, sc, compact, ci, cj, sig, wide)
                    assert got.dtype == ref.dtype == (np.float64 if wide else np.float32)
                    np.testing.assert_array_equal(got, ref)
            ref = g[f"mex_{X}x{Y}_s{sig}_{tag}_nc"]
            got = O.neigh_mexican_hat(X, Y, 1.0, False, ci, cj, sig, wide)
            assert got.dtype == ref.dtype
            np.testing.assert_array_equal(got, ref)


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
    f = O.DECAYS[decay]
    for tag, w, t in (("init", w0, 0), ("mid", wmid, T // 2), ("last", wmid, T - 1)):
        eta = f(0.5, 0.01, t, T)
        sig = f(min(X, Y) / 2, 1, t, T)
        assert float(eta) == float(g[f"{decay}_{tag}_eta"])
        assert float(sig) == float(g[f"{decay}_{tag}_sig"])
        bmu, num, den, wout = O.epoch(data, w, eta, sig, wide=wide, n_parallel=n)
        assert np.array_equal(bmu, g[f"{decay}_{tag}_bmu"])
        np.testing.assert_allclose(den[:, :, 0], g[f"{decay}_{tag}_den"][:, :, 0], rtol=2e-6, atol=0)
        if f"{decay}_{tag}_num" in g:
            np.testing.assert_allclose(num, g[f"{decay}_{tag}_num"], rtol=1e-5, atol=1e-6)
        ok = den[:, :, 0] > 1e-30
        np.testing.assert_allclose(wout[ok], g[f"{decay}_{tag}_wout"][ok], rtol=1e-5, atol=2e-6)
        if f"{decay}_{tag}_wout77" in g:
            _, _, _, w77 = O.epoch(data, w, eta, sig, wide=wide, n_parallel=77)
            np.testing.assert_allclose(w77[ok], g[f"{decay}_{tag}_wout77"][ok], rtol=1e-5, atol=2e-6)


def test_g5_wmid_from_full_training():
    """Five reference epochs from the default codebook (well-conditioned small cases only)."""
    for shape in ("6x6x4", "8x8x3"):
        g = load_golden("g4_update_" + shape)
        X, Y, D, n = (int(v) for v in g["shape"])
        data = O.gaussian_blobs(n, D, seed=int(g["data_seed"]))
        for decay in ("linear", "exponential"):
            w = O.train(data, O.default_codebook(X, Y, D, 1234), 10, sigma0=min(X, Y) / 2,
                        decay=decay, n_parallel=n, iter_end=5)
            np.testing.assert_allclose(w, g[f"{decay}_wmid"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("decay", ["linear", "exponential"])
def test_g7_shard_identity(decay):
    g = load_golden("g4_update_24x24x16")
    X, Y, D, n = (int(v) for v in g["shape"])
    data = O.gaussian_blobs(n, D, seed=int(g["data_seed"]))
    w = g[f"{decay}_wmid"]
    eta, sig = g[f"{decay}_mid_eta"], g[f"{decay}_mid_sig"]
    eta = np.float64(eta) if decay == "exponential" else float(eta)
    sig = np.float64(sig) if decay == "exponential" else float(sig)
    wide = O.decay_is_wide(decay)
    num = np.zeros((X, Y, D), F32)
    den = np.zeros((X, Y, 1), F32)
    for part in np.array_split(np.arange(n), 2):
        _, a, b = O.update(data[part], w, eta, sig, wide=wide)
        num += a.astype(F32)
        den += b.astype(F32)
    np.testing.assert_allclose(num, g[f"{decay}_shard2_num"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(den, g[f"{decay}_shard2_den"], rtol=1e-5, atol=1e-30)
    # and the shard sum equals the unsharded update (the property RCCL all-reduce relies on)
    np.testing.assert_allclose(num, g[f"{decay}_mid_num"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("decay", ["linear", "exponential"])
@pytest.mark.parametrize("init", ["default", "random", "pca"])
def test_g6_iris_end_to_end(decay, init):
    g = load_golden("g6_iris")
    z = g["iris_z"]
    w0 = g[f"{decay}_{init}_w0"]
    assert abs(O.quantization_error(z, w0) - float(g[f"{decay}_{init}_qe0"])) < 1e-6
    w = O.train(z, w0, 100, sigma0=3.0, decay=decay, n_parallel=4000)
    np.testing.assert_allclose(w, g[f"{decay}_{init}_w"], rtol=1e-4, atol=1e-5)
    assert np.array_equal(O.winner_ids(z, w), g[f"{decay}_{init}_bmu"])
    assert abs(O.quantization_error(z, w) - float(g[f"{decay}_{init}_qe"])) < 1e-5


def test_g6_readme_config_qe():
    g = load_golden("g6_iris")
    raw = g["iris_raw"]
    w0 = O.default_codebook(6, 6, 4, 10)
    assert abs(O.quantization_error(raw, w0) - float(g["readme_qe0"])) < 1e-5
    assert abs(float(g["readme_qe"]) - 0.296985) < 1e-5        # SURVEY 8(c) anchor


@pytest.mark.parametrize("decay", ["linear", "exponential"])
def test_g8_cosine_mexican_hat(decay):
    g = load_golden("g8_cosine_mexican")
    data, w0 = g["data"], g[f"{decay}_w0"]
    wide = O.decay_is_wide(decay)
    f = O.DECAYS[decay]
    eta, sig = f(0.5, 0.01, 0, 10), f(4.0, 1, 0, 10)
    assert float(eta) == float(g[f"{decay}_eta"]) and float(sig) == float(g[f"{decay}_sig"])
    bmu, num, den, wout = O.epoch(data, w0, eta, sig, wide=wide, n_parallel=len(data),
                                  distance="cosine", neighbourhood="mexican_hat")
    assert np.array_equal(bmu, g[f"{decay}_bmu"])
    np.testing.assert_allclose(num, g[f"{decay}_num"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(den, g[f"{decay}_den"], rtol=1e-4, atol=1e-5)


def test_g8_cosine_gaussian():
    g = load_golden("g8_cosine_mexican")
    data = g["data"]
    w0 = np.abs(O.default_codebook(8, 8, 6, 3)).astype(F32)
    eta, sig = O.linear_decay(0.5, 0.01, 0, 10), O.linear_decay(4.0, 1, 0, 10)
    bmu, _, den, wout = O.epoch(data, w0, eta, sig, wide=False, n_parallel=len(data), distance="cosine")
    assert np.array_equal(bmu, g["cosgauss_bmu"])
    np.testing.assert_allclose(wout, g["cosgauss_wout"], rtol=1e-5, atol=1e-6)


def test_g9_inference():
    g = load_golden("g9_inference")
    probe = O.gaussian_blobs(700, 10, seed=int(g["probe_seed"]))
    w = g["w"]
    assert np.array_equal(O.winner_ids(probe, w), g["winner"])
    assert np.array_equal(O.winner_ids(probe.astype(np.float64), w), g["winner64"])
    assert abs(O.quantization_error(probe, w) - float(g["qe"])) < 1e-6


def test_reference_unit_test_constants():
    """Known answers from the reference's unittest file that need no MiniSom
    (xpysom_dask/tests.py:31-33,66-67,77-79)."""
    w = np.zeros((5, 5, 1))
    w[2, 3] = 5.0
    w[1, 1] = 2.0
    assert O.quantization_error([[5], [2]], w) == 0.0
    assert O.quantization_error([[4], [1]], w) == 1.0
    assert O.dist_euclid_part(np.array([[5.0]]), w.reshape(-1, 1)).argmin() == 13
    for i in range(5):
        for j in range(5):
            assert abs(np.linalg.norm(O.default_codebook(5, 5, 1, None)[i, j]) - 1.0) < 1e-7


def test_decays_match_closed_forms():
    assert O.asymptotic_decay(2.0, 9, 5, 10) == 1.0
    assert O.linear_decay(3.0, 1.0, 9, 10) == 1.0 and O.linear_decay(3.0, 1.0, 0, 1) == 3.0
    assert abs(O.exponential_decay(8.0, 1.0, 10, 10) - 1.0) < 1e-12
    assert abs(O.exponential_decay(1.0, 0, 10, 10) - 0.1) < 1e-12
    assert isinstance(O.exponential_decay(8.0, 1.0, 3, 10), np.float64)
    assert isinstance(O.linear_decay(8.0, 1.0, 3, 10), float)


def test_g9_topographic_error():
    g = load_golden("g9_inference")
    probe = O.gaussian_blobs(700, 10, seed=int(g["probe_seed"]))
    assert np.array_equal(O.top2_ids(probe, g["w"]), g["top2"])
    assert O.topographic_error(probe, g["w"]) == float(g["te"])
    # known answers of the reference's unit test (xpysom_dask/tests.py:81-90)
    w = np.zeros((5, 5, 1))
    w[2, 3], w[1, 1], w[2, 4], w[4, 4], w[0, 0] = 5.0, 2.0, 6.0, 15.0, 14.0
    assert O.topographic_error([[5]], w) == 0.0
    assert O.topographic_error([[15]], w) == 1.0


@pytest.mark.parametrize("XY", [(6, 5, 3, 200), (9, 8, 4, 400)])
def test_g10_hexagonal_topology(XY):
    g = load_golden("g10_hexagonal")
    X, Y, D, n = XY
    ci, cj = np.divmod(np.arange(X * Y), Y)
    for sig in (0.8, 2.5):
        for wide in (False, True):
            tag = f"{X}x{Y}_s{sig}_{'f64' if wide else 'f32'}"
            np.testing.assert_array_equal(O.neigh_gaussian_hex(X, Y, 0.5, False, ci, cj, sig, wide), g["gauss_" + tag])
            np.testing.assert_array_equal(O.neigh_gaussian_hex(X, Y, 0.5, True, ci, cj, sig, wide), g["gausscs_" + tag])
            np.testing.assert_array_equal(O.neigh_mexican_hat_hex(X, Y, 0.5, False, ci, cj, sig, wide), g["mex_" + tag])
    data = O.gaussian_blobs(n, D, seed=300 + X)
    w0 = O.default_codebook(X, Y, D, 77).astype(F32)
    for neigh in ("gaussian", "mexican_hat", "bubble"):
        for decay in ("linear", "exponential"):
            key = f"{X}x{Y}_{neigh}_{decay}"
            f = O.DECAYS[decay]
            eta, sig = f(0.5, 0.01, 2, 6), f(min(X, Y) / 2, 1, 2, 6)
            assert float(eta) == float(g[key + "_eta"]) and float(sig) == float(g[key + "_sig"])
            bmu, num, den = O.update(data, w0, eta, sig, wide=O.decay_is_wide(decay), neighbourhood=neigh + "_hex")
            assert np.array_equal(bmu, g[key + "_bmu"])
            np.testing.assert_allclose(num, g[key + "_num"], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(den, g[key + "_den"], rtol=1e-5, atol=1e-6)


def test_g2_manhattan_and_norm_p():
    """The remaining registry entries (distances.py:165-169) against the reference's matrices and
    the closed forms of its own tests (test_distances.py:115-134)."""
    g = load_golden("g2_distances")
    for c in range(int(g["n_cases"])):
        x, w = g[f"c{c:03d}_x"], g[f"c{c:03d}_w"]
        np.testing.assert_array_equal(O.dist_manhattan(x, w), g[f"c{c:03d}_l1"])
        np.testing.assert_array_equal(O.dist_norm_p(x, w, 2), g[f"c{c:03d}_p2"])
        np.testing.assert_array_equal(O.dist_norm_p(x, w, 3), g[f"c{c:03d}_p3"])
        np.testing.assert_array_equal(O.dist_norm_p(x, w, 4), g[f"c{c:03d}_p4"])
        for i, vx in enumerate(x):
            for j, vy in enumerate(w):
                assert abs(g[f"c{c:03d}_l1"][i, j] - np.abs(vx - vy).sum()) < 1e-7
                assert abs(g[f"c{c:03d}_p3"][i, j] - (np.abs(vx - vy) ** 3).sum()) < 1e-7
                assert abs(g[f"c{c:03d}_p4"][i, j] - ((vx - vy) ** 4).sum()) < 1e-7


def test_g9_winner_under_the_other_distances():
    g = load_golden("g9_inference")
    probe = O.gaussian_blobs(700, 10, seed=int(g["probe_seed"]))
    w = g["w"].reshape(-1, 10)
    for name, p in (("manhattan", 1), ("norm_p", 2), ("norm_p", 3), ("norm_p", 4), ("norm_p_no_opt", 2)):
        assert np.array_equal(O.bmu_ids_pairwise(probe, w, name, p), g["win_%s_p%d" % (name, p)]), (name, p)


# ----------------------------------------------------------------------------- G11 bubble / triangle (rectangular)
G11_SIGMAS = (0.3, 1.0, 2.0, 2.5, 3.0000000000000004)


@pytest.mark.parametrize("XY", [(5, 5), (3, 4)])
def test_g11_bubble_triangle_tensors(XY):
    """neighborhoods.py:99-130, every centre, sigma on and next to the open box's edge; the triangle is float64
    for both sigma types (int64 - ... + sigma), the bubble float32."""
    g = load_golden("g11_bubble_triangle")
    X, Y = XY
    ci, cj = np.divmod(np.arange(X * Y), Y)
    for sig in G11_SIGMAS:
        for wide in (False, True):
            key = f"{X}x{Y}_s{sig!r}_{'f64' if wide else 'f32'}"
            ref = g["bubble_" + key]
            got = O.neigh_bubble(X, Y, 0.5, False, ci, cj, sig, wide)
            assert got.dtype == ref.dtype == np.float32
            np.testing.assert_array_equal(got, ref)
            for compact in (False, True):
                ref = g["tri_" + key + ("_cs" if compact else "_nc")]
                got = O.neigh_triangle(X, Y, 0.5, compact, ci, cj, sig, wide)
                assert got.dtype == ref.dtype == np.float64
                np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("shape", [(8, 8, 3, 500), (5, 7, 4, 300)])
@pytest.mark.parametrize("neigh,compact", [("bubble", False), ("triangle", False), ("triangle", True)])
def test_g11_update_and_epoch(shape, neigh, compact):
    g = load_golden("g11_bubble_triangle")
    X, Y, D, n = shape
    data = O.gaussian_blobs(n, D, seed=400 + X)
    w0 = O.default_codebook(X, Y, D, 21).astype(F32)
    for decay in ("linear", "exponential", "asymptotic"):
        key = f"{X}x{Y}x{D}_{neigh}{'_cs' if compact else ''}_{decay}"
        f, wide = O.DECAYS[decay], O.decay_is_wide(decay)
        eta, sig = f(0.5, 0.01, 1, 6), f(min(X, Y) / 2, 1, 1, 6)
        assert float(eta) == float(g[key + "_eta"]) and float(sig) == float(g[key + "_sig"])
        bmu, num, den = O.update(data, w0, eta, sig, wide=wide, neighbourhood=neigh, compact=compact)
        assert np.array_equal(bmu, g[key + "_bmu"])
        assert str(num.dtype) == str(g[key + "_numdtype"])
        np.testing.assert_allclose(num, g[key + "_num"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(den, g[key + "_den"], rtol=1e-6, atol=0)
        _, _, _, wout = O.epoch(data, w0, eta, sig, wide=wide, n_parallel=n, neighbourhood=neigh, compact=compact)
        np.testing.assert_allclose(wout, g[key + "_wout"], rtol=1e-5, atol=2e-6)


# ----------------------------------------------------------------------------- G12 the configs[1] map
@pytest.mark.parametrize("decay,tag", [("linear", "init"), ("exponential", "init"), ("exponential", "mid")])
def test_g12_update_64x64x32(decay, tag):
    g = load_golden("g12_update_64x64x32")
    X, Y, D, n = (int(v) for v in g["shape"])
    T, st = int(g["T"]), int(g["stride"])
    data = O.gaussian_blobs(n, D, seed=int(g["data_seed"]))
    w = O.default_codebook(X, Y, D, 1234).astype(F32) if tag == "init" else g["exponential_wmid"]
    t = 0 if tag == "init" else T // 2
    f, wide = O.DECAYS[decay], O.decay_is_wide(decay)
    eta, sig = f(0.5, 0.01, t, T), f(min(X, Y) / 2, 1, t, T)
    key = f"{decay}_{tag}"
    assert float(eta) == float(g[key + "_eta"]) and float(sig) == float(g[key + "_sig"])
    bmu, num, den, wout = O.epoch(data, w, eta, sig, wide=wide, n_parallel=n)
    assert np.array_equal(bmu, g[key + "_bmu"])
    np.testing.assert_allclose(den[:, :, 0], g[key + "_den"][:, :, 0], rtol=2e-6, atol=0)
    np.testing.assert_allclose(num.reshape(-1, D)[::st], g[key + "_num16"], rtol=1e-5, atol=1e-6)
    ok = den.reshape(-1)[::st] > 1e-30
    np.testing.assert_allclose(wout.reshape(-1, D)[::st][ok], g[key + "_wout16"][ok], rtol=1e-5, atol=2e-6)


# ----------------------------------------------------------------------------- G13 hexagonal topographic error
@pytest.mark.parametrize("XD", [(5, 3), (12, 6)])
def test_g13_hex_topographic_error(XD):
    g = load_golden("g13_hex_topographic")
    X, D = XD
    key = f"{X}x{X}x{D}"
    seeds = g[key + "_seeds"]
    probe = O.gaussian_blobs(400, D, seed=int(seeds[1]))
    w = g[key + "_w"]
    assert np.array_equal(O.top2_ids(probe, w), g[key + "_top2"])
    assert O.topographic_error(probe, w, topology="hexagonal") == float(g[key + "_te"])
    n = {5: 300, 12: 1500}[X]
    assert O.topographic_error(O.gaussian_blobs(n, D, seed=int(seeds[0])), w, topology="hexagonal") == float(g[key + "_te_train"])


# ----------------------------------------------------------------------------- G15 mexican_hat + compact_support
@pytest.mark.parametrize("topo,XY", [("rect", (5, 5)), ("hex", (6, 5)), ("hex", (5, 5))])
def test_g15_mexican_hat_compact_tensors(topo, XY):
    """The reference's double mask on px (neighborhoods.py:69-71, :91-93), restated literally: bit-exact tensors."""
    g = load_golden("g15_mexican_compact")
    X, Y = XY
    ci, cj = np.divmod(np.arange(X * Y), Y)
    f = O.neigh_mexican_hat if topo == "rect" else O.neigh_mexican_hat_hex
    for sig in (0.8, 1.7, 2.5):
        for wide in (False, True):
            ref = g[f"mexcs_{topo}_{X}x{Y}_s{sig}_{'f64' if wide else 'f32'}"]
            got = f(X, Y, 0.5, True, ci, cj, sig, wide)
            assert got.dtype == ref.dtype
            np.testing.assert_array_equal(got, ref)
    if topo == "rect":
        with pytest.raises(ValueError):                        # non-square: the second mask does not broadcast
            O.neigh_mexican_hat(5, 7, 0.5, True, np.zeros(3, int), np.zeros(3, int), 1.0, True)


@pytest.mark.parametrize("topo,shape", [("rectangular", (9, 9, 4, 400)), ("hexagonal", (9, 8, 4, 400)), ("hexagonal", (7, 7, 3, 300))])
def test_g15_update(topo, shape):
    g = load_golden("g15_mexican_compact")
    X, Y, D, n = shape
    data = O.gaussian_blobs(n, D, seed=600 + X + Y)
    w0 = O.default_codebook(X, Y, D, 41).astype(F32)
    for decay in ("linear", "exponential"):
        key = f"{topo}_{X}x{Y}x{D}_{decay}"
        f, wide = O.DECAYS[decay], O.decay_is_wide(decay)
        eta, sig = f(0.5, 0.01, 2, 6), f(min(X, Y) / 2, 1, 2, 6)
        assert float(eta) == float(g[key + "_eta"]) and float(sig) == float(g[key + "_sig"])
        bmu, num, den = O.update(data, w0, eta, sig, wide=wide, compact=True,
                                 neighbourhood="mexican_hat" + ("_hex" if topo == "hexagonal" else ""))
        assert np.array_equal(bmu, g[key + "_bmu"])
        np.testing.assert_allclose(num, g[key + "_num"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(den, g[key + "_den"], rtol=1e-5, atol=1e-5)


# ----------------------------------------------------------------------------- G16 hexagonal compact_support, lattice sigma
@pytest.mark.parametrize("XY", [(10, 12), (9, 7)])
def test_g16_hexagonal_compact_support_tensors_at_a_sigma_one_ulp_off_the_lattice(XY):
    """sigma = 5 / (1 + 2/3) = 3.0000000000000004: the generic masks round cx -/+ sigma before comparing
    (neighborhoods.py:50-54, :91-93), so the boundary units depend on the BMU's absolute coordinate.  Bit-exact."""
    g = load_golden("g16_hex_compact_lattice_sigma")
    X, Y = XY
    sig = float(g["sigma"])
    assert sig != 3.0 and abs(sig - 3.0) < 1e-15
    ci, cj = np.divmod(np.arange(X * Y), Y)
    for wide in (False, True):
        tag = f"{X}x{Y}_{'f64' if wide else 'f32'}"
        for name, f in (("gauss_", O.neigh_gaussian_hex), ("mex_", O.neigh_mexican_hat_hex)):
            got = f(X, Y, 1.0, True, ci, cj, sig, wide)
            assert got.dtype == g[name + tag].dtype
            np.testing.assert_array_equal(got, g[name + tag])
    # the mask IS position dependent at this sigma: the same (dx = 3.0) pair is inside for one BMU and outside for another
    m = g[f"gauss_{X}x{Y}_f64"] != 0
    inside = []
    for b in range(X * Y):
        i0, j0 = divmod(b, Y)
        if i0 + 3 < X:
            inside.append(bool(m[b, i0 + 3, j0]))            # same row: dx = 3.0 exactly
    assert any(inside) and not all(inside)


@pytest.mark.parametrize("neigh", ["gaussian", "mexican_hat"])
def test_g16_update(neigh):
    g = load_golden("g16_hex_compact_lattice_sigma")
    X, Y, D, n = 10, 12, 16, 200
    data = O.gaussian_blobs(n, D, seed=1131)
    w0 = O.default_codebook(X, Y, D, 131).astype(F32)
    eta, sig = O.asymptotic_decay(0.5, 0.01, 1, 3), O.asymptotic_decay(5.0, 1, 1, 3)
    assert float(eta) == float(g[neigh + "_eta"]) and float(sig) == float(g["sigma"])
    bmu, num, den = O.update(data, w0, eta, sig, wide=O.decay_is_wide("asymptotic"), compact=True, std_coeff=1.0,
                             neighbourhood=neigh + "_hex")
    assert np.array_equal(bmu, g[neigh + "_bmu"])
    np.testing.assert_allclose(num, g[neigh + "_num"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(den, g[neigh + "_den"], rtol=1e-5, atol=1e-5)



@pytest.mark.parametrize("decay", ["linear", "exponential"])
def test_g17_configs4_semantics_64x64x784(decay):
    """cosine + mexican_hat at 784 features (BASELINE configs[4]'s semantics): BMUs, denominator, strided numerator.
    The reference's sgemm splits K = 784 into blocks; NumPy here does the same on the same host, another host's BLAS
    may order the sum differently: a few float32 near-ties are allowed to move."""
    g = load_golden("g17_configs4_64x64x784")
    X, Y, D, n = (int(v) for v in g["shape"])
    st = int(g["stride"])
    data = np.abs(O.gaussian_blobs(n, D, seed=int(g["data_seed"])))
    data /= np.linalg.norm(data, axis=1, keepdims=True)
    data = data.astype(F32)
    w = np.abs(O.default_codebook(X, Y, D, 1234)).astype(F32)
    wide = O.decay_is_wide(decay)
    sig = np.float64(g[decay + "_sig"]) if wide else float(g[decay + "_sig"])
    eta = np.float64(g[decay + "_eta"]) if wide else float(g[decay + "_eta"])
    bmu, num, den = O.update(data, w, eta, sig, wide=wide, neighbourhood="mexican_hat", distance="cosine")
    assert (bmu != g[decay + "_bmu"]).sum() <= max(2, n // 500)
    if np.array_equal(bmu, g[decay + "_bmu"]):
        np.testing.assert_allclose(den.reshape(-1).astype(F32), g[decay + "_den"].reshape(-1), rtol=1e-5, atol=1e-5 * np.abs(g[decay + "_den"]).max())
        ref = g[decay + "_num32"]
        assert np.abs(num.reshape(-1, D)[::st] - ref).max() <= 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize("state", ["seeded", "sheet"])
def test_g18_bmus_256x256x128(state):
    """The oracle's winner ids at the configs[2] shape against the reference's, and the host-independent codebook recipe."""
    import zlib
    g = load_golden("g18_bmus_256x256x128")
    X, Y, D, n = (int(v) for v in g["shape"])
    data = O.gaussian_blobs(n, D, seed=int(g["data_seed"]))
    if state == "seeded":
        w = O.default_codebook(X, Y, D, int(g["codebook_seed"])).astype(F32)
    else:
        w = O.smooth_sheet_codebook(X, Y, D, int(g["sheet_seed"]), amplitude=float(g["sheet_amplitude"]),
                                    centre=data.astype(np.float64).mean(0))
    assert zlib.crc32(np.ascontiguousarray(w).tobytes()) == int(g[state + "_w_crc"])
    ids = O.winner_ids(data[:1024], w, n_parallel=1024)          # (a quarter of the rows: 8.6 GFLOP on the CPU)
    assert (ids != g[state + "_bmu"][:1024]).sum() <= 2


@pytest.mark.parametrize("i", [0, 1])
def test_g20_two_resident_epochs_case_a(i):
    """The oracle's epoch against the reference's two consecutive teacher-forced epochs (64 x 64 x 32, its own states)."""
    g = load_golden("g20_two_resident_epochs")
    X, Y, D, n = (int(v) for v in g["a_shape"])
    st = int(g["a_stride"])
    data = O.gaussian_blobs(n, D, seed=int(g["a_data_seed"]))
    w = g["a_w%d" % i]
    key = "a_e%d" % i
    bmu, num, den, wout = O.epoch(data, w, np.float64(g[key + "_eta"]), np.float64(g[key + "_sig"]), wide=True, n_parallel=2048)
    assert (bmu != g[key + "_bmu"]).sum() <= 2
    if np.array_equal(bmu, g[key + "_bmu"]):
        gden = g[key + "_den"]
        np.testing.assert_allclose(den.reshape(-1).astype(F32), gden, rtol=1e-5, atol=1e-5 * np.abs(gden).max())
        ref = g[key + "_num"]
        assert np.abs(num.reshape(-1, D)[::st] - ref).max() <= 1e-5 * np.abs(ref).max()
        ok = gden[::st] > 1e-30
        np.testing.assert_allclose(wout.reshape(-1, D)[::st][ok], g[key + "_wout"][ok], rtol=1e-5, atol=1e-5 * np.abs(g[key + "_wout"]).max())
    if i == 0:
        # the reference's own next state is what the second epoch starts from
        assert np.array_equal(g["a_w1"].reshape(-1, D)[::st], g["a_e0_wout"])


def test_g20_case_b_recipes_and_a_slice_of_its_winners():
    """256 x 256 x 128: the host-independent recipes of both codebooks and of the rows (crc), and the oracle's winners on a
    slice of the rows against the reference's, both epochs."""
    import zlib
    g = load_golden("g20_two_resident_epochs")
    X, Y, D, n = (int(v) for v in g["b_shape"])
    s0, s1, s2 = (int(v) for v in g["b_seeds"])
    w0 = O.smooth_sheet_codebook(X, Y, D, s0, amplitude=float(g["b_amplitude"]))
    w1 = O.sheet_step(w0, O.smooth_sheet_codebook(X, Y, D, s1, amplitude=float(g["b_amplitude"])), float(g["b_mix"]))
    gen = O.rows_on_codebook(w0, n + 256, s2, float(g["b_noise"]))
    # (drawn with 256 spare rows; the generator dropped the float32 near-ties -- top-2 gap below 4e-6 -- it lists)
    data = np.ascontiguousarray(gen[np.setdiff1d(np.arange(len(gen)), g["b_dropped"])[:n]])
    assert zlib.crc32(np.ascontiguousarray(w0).tobytes()) == int(g["b_w0_crc"])
    assert zlib.crc32(np.ascontiguousarray(w1).tobytes()) == int(g["b_w1_crc"])
    assert zlib.crc32(np.ascontiguousarray(data).tobytes()) == int(g["b_data_crc"])
    for i, w in enumerate((w0, w1)):
        ids = O.winner_ids(data[:512], w, n_parallel=512)
        assert (ids != g["b_e%d_bmu" % i][:512]).sum() <= 1


def test_g19_norm_p_with_a_real_exponent():
    """distances.py:61-75 takes any real p: the oracle's generic form against the reference's winners and a distance block."""
    g = load_golden("g19_norm_p_real")
    g9 = load_golden("g9_inference")
    D = 10
    probe = O.gaussian_blobs(700, D, seed=int(g["probe_seed"]))
    w = g9["w"].reshape(-1, D).astype(F32)
    for p in (0.5, 1.5, 2.5, 3.7):
        tag = str(p).replace(".", "_")
        d = O.dist_norm_p_generic(probe[:40].astype(F32), w[:60], p)
        assert np.array_equal(d.astype(F32), g["dist_p" + tag])
        ids = O.bmu_ids_pairwise(probe, w, "norm_p", p)
        assert np.array_equal(ids, g["win_norm_p_p" + tag])
        assert np.array_equal(ids, g["win_norm_p_no_opt_p" + tag])
