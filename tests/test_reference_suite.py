"""The reference's own unit tests (xpysom_dask/tests.py, TestCupySom / TestCupySomHex), test by test and under the
same names, run against the drop-in class.  Where the reference compares with MiniSom (an un-vendored dependency,
absent here) the expected values come from the golden tensors captured from the reference itself (tests/golden/,
oracle/make_golden.py) or from the pinned oracle.  Tests that reach the engine need the GPU; the rest run on CPU."""
import os
import pickle

import numpy as np
import pytest

from oracle import som_oracle as O
from tests.conftest import load_golden

gpu = pytest.mark.gpu
F32 = np.float32


@pytest.fixture
def som():
    """tests.py:21-38 setUp: 5x5x1 map, std_coeff 1, fake weights with two marked units."""
    from xpysom_dask_amd import XPySom
    s = XPySom(5, 5, 1, std_coeff=1)
    for i in range(5):
        for j in range(5):
            np.testing.assert_almost_equal(1.0, np.linalg.norm(s._weights[i, j]))   # weights normalisation
    s._weights = np.zeros((5, 5, 1))
    s._weights[2, 3] = 5.0
    s._weights[1, 1] = 2.0
    np.random.seed(1234)
    return s


def test_unavailable_neigh_function():
    from xpysom_dask_amd import XPySom
    with pytest.raises(ValueError):
        XPySom(5, 5, 1, neighborhood_function='boooom')


def test_unavailable_distance_function():
    from xpysom_dask_amd import XPySom
    with pytest.raises(ValueError):
        XPySom(5, 5, 1, activation_distance='ridethewave')


@gpu
def test_win_map(som):
    winners = som.win_map([[5.0], [2.0]])
    assert winners[(2, 3)][0] == [5.0]
    assert winners[(1, 1)][0] == [2.0]


@gpu
def test_labels_map(som):
    labels_map = som.labels_map([[5.0], [2.0]], ['a', 'b'])
    assert labels_map[(2, 3)]['a'] == 1
    assert labels_map[(1, 1)]['b'] == 1
    with pytest.raises(ValueError):
        som.labels_map([[5.0]], ['a', 'b'])


@gpu
def test_activation_reponse(som):
    response = som.activation_response([[5.0], [2.0]])
    assert response[2, 3] == 1
    assert response[1, 1] == 1


@gpu
def test_activate(som):
    assert som.activate(5.0).argmin() == 13.0  # unravel(13) = (2,3)


@gpu
def test_distance_from_weights(som):
    data = np.arange(-5, 5).reshape(-1, 1)
    weights = som._weights.reshape(-1, som._weights.shape[2])
    distances = som.distance_from_weights(data, weights)
    for i in range(len(data)):
        for j in range(len(weights)):
            assert distances[i][j] == np.linalg.norm(data[i] - weights[j])


@gpu
def test_quantization_error(som):
    assert som.quantization_error([[5], [2]]) == 0.0
    assert som.quantization_error([[4], [1]]) == 1.0


@gpu
def test_topographic_error(som):
    # 5 will have bmu_1 in (2,3) and bmu_2 in (2, 4): same neighbourhood; 15: (4, 4) and (0, 0): not
    som._weights[2, 4] = 6.0
    som._weights[4, 4] = 15.0
    som._weights[0, 0] = 14.
    assert som.topographic_error([[5]]) == 0.0
    assert som.topographic_error([[15]]) == 1.0


@gpu
def test_quantization(som):
    q = som.quantization(np.array([[4], [2]]))
    assert q[0] == 5.0
    assert q[1] == 2.0


@gpu
def test_random_seed():
    from xpysom_dask_amd import XPySom
    som1 = XPySom(5, 5, 2, sigma=1.0, learning_rate=0.5, random_seed=1)
    som2 = XPySom(5, 5, 2, sigma=1.0, learning_rate=0.5, random_seed=1)
    np.testing.assert_array_almost_equal(som1._weights, som2._weights)          # same initialization
    np.random.seed(1234)
    data = np.random.rand(100, 2)
    som1 = XPySom(5, 5, 2, sigma=1.0, learning_rate=0.5, random_seed=1)
    som1.train_random(data, 10)
    som2 = XPySom(5, 5, 2, sigma=1.0, learning_rate=0.5, random_seed=1)
    som2.train_random(data, 10)
    np.testing.assert_array_equal(som1._weights, som2._weights)                 # same state after training, bit for bit


@gpu
def test_train(capsys):
    from xpysom_dask_amd import XPySom
    som = XPySom(5, 5, 2, sigma=1.0, learning_rate=0.5, random_seed=1)
    data = np.array([[4, 2], [3, 1]])
    q1 = som.quantization_error(data)
    som.train(data, 10)
    assert q1 > som.quantization_error(data)
    data = np.array([[1, 5], [6, 7]])
    q1 = som.quantization_error(data)
    som.train(data, 10, verbose=True)
    assert q1 > som.quantization_error(data)
    assert "quantization error" in capsys.readouterr().out


def test_random_weights_init():
    from xpysom_dask_amd import XPySom
    som = XPySom(2, 2, 2, random_seed=1)
    som.random_weights_init(np.array([[1.0, .0]]))
    for w in som._weights:
        np.testing.assert_array_equal(w[0], np.array([1.0, .0]))


def test_pca_weights_init():
    from xpysom_dask_amd import XPySom
    som = XPySom(2, 2, 2)
    som.pca_weights_init(np.array([[1., 0.], [0., 1.], [1., 0.], [0., 1.]]))
    expected = np.array([[[0., -1.41421356], [-1.41421356, 0.]],
                         [[1.41421356, 0.], [0., 1.41421356]]])
    np.testing.assert_array_almost_equal(som._weights, expected)


def test_distance_map():
    from xpysom_dask_amd import XPySom
    som = XPySom(2, 2, 2, random_seed=1)
    som._weights = np.array([[[1., 0.], [0., 1.]], [[1., 0.], [0., 1.]]])
    np.testing.assert_array_equal(som.distance_map(), np.array([[1., 1.], [1., 1.]]))
    som = XPySom(2, 2, 2, topology='hexagonal', random_seed=1)      # (the reference checks MiniSom here)
    som._weights = np.array([[[1., 0.], [0., 1.]], [[1., 0.], [0., 1.]]])
    np.testing.assert_array_equal(som.distance_map(), np.array([[.5, 1.], [1., .5]]))


def test_pickling(som, tmp_path):
    with open(tmp_path / 'som.p', 'wb') as outfile:
        pickle.dump(som, outfile)
    with open(tmp_path / 'som.p', 'rb') as infile:
        back = pickle.load(infile)
    np.testing.assert_array_equal(back._weights, som._weights)
    os.remove(tmp_path / 'som.p')


# --------------------------------------------------------------------------- distances (tests.py:152-186)
def _random_case():
    np.random.seed(1234)
    return np.random.rand(100, 20), np.random.rand(10, 10, 20)


@gpu
def test_euclidean_distance():
    from xpysom_dask_amd import XPySom
    x, w = _random_case()
    som = XPySom(10, 10, 20, activation_distance='euclidean_no_opt')
    som._weights = w
    cs_dist = som.activate(x).reshape((100, 10, 10))
    for i, sample in enumerate(x):
        ms_dist = np.linalg.norm(sample - w, axis=-1) ** 2                       # MiniSom._euclidean_distance ** 2
        np.testing.assert_array_almost_equal(ms_dist, cs_dist[i], decimal=5)


@gpu
def test_cosine_distance():
    from xpysom_dask_amd import XPySom
    x, w = _random_case()
    som = XPySom(10, 10, 20, activation_distance='cosine')
    som._weights = w
    cs_dist = som.activate(x).reshape((100, 10, 10))
    for i, sample in enumerate(x):
        ms_dist = 1 - (w * sample).sum(axis=2) / (np.linalg.norm(w, axis=2) * np.linalg.norm(sample))   # MiniSom._cosine_distance
        np.testing.assert_array_almost_equal(ms_dist, cs_dist[i], decimal=6)


@gpu
def test_manhattan_distance():
    """The engine fuses this distance with its argmin (no (n, K) matrix): the winners against the definition."""
    from xpysom_dask_amd import XPySom
    x, w = _random_case()
    som = XPySom(10, 10, 20, activation_distance='manhattan')
    som._weights = w
    ms = np.abs(x[:, None, None, :] - w[None]).sum(-1).reshape(100, -1)          # MiniSom._manhattan_distance
    want = [tuple(int(v) for v in np.unravel_index(k, (10, 10))) for k in ms.argmin(1)]
    assert [tuple(int(v) for v in t) for t in som.winner(x)] == want


# --------------------------------------------------------------------------- neighbourhoods (tests.py:188-305)
def _neighbourhood_on_device(X, Y, name, sigma, wide, topology="rectangular", compact=False, std_coeff=1.0):
    """h(c -> .) for every centre c as the engine evaluates it: one sample of value 1, its BMU forced to c, eta = 1 --
    the denominator of that update IS the neighbourhood slice (xpysom.py:436)."""
    from xpysom_dask_amd.engine import HipEngine
    e = HipEngine(X, Y, 1, neighborhood=name, topology=topology, compact_support=compact, std_coeff=std_coeff)
    e.set_weights(np.zeros((X * Y, 1), F32))
    e.set_data(np.ones((1, 1), F32))
    out = np.empty((X * Y, X, Y), F32)
    for c in range(X * Y):
        e.epoch_accumulate_forced(np.array([c], np.int32), sigma, 1.0, wide)
        _, den, _ = e.epoch_fetch(want_bmu=False)
        out[c] = den.reshape(X, Y)
    return out


@gpu
@pytest.mark.parametrize("wide", [False, True])
def test_gaussian(wide):
    g = load_golden("g3_neighbourhoods")                         # gaussian_rect, std_coeff 1, every centre of the 5x5 map
    got = _neighbourhood_on_device(5, 5, "gaussian", 1.0, wide)
    np.testing.assert_allclose(got, g["gauss_5x5_s1.0_%s_nc" % ("f64" if wide else "f32")], rtol=2e-6, atol=1e-7)


@gpu
@pytest.mark.parametrize("wide", [False, True])
def test_mexican_hat(wide):
    g = load_golden("g3_neighbourhoods")
    got = _neighbourhood_on_device(5, 5, "mexican_hat", 1.0, wide)
    np.testing.assert_allclose(got, g["mex_5x5_s1.0_%s_nc" % ("f64" if wide else "f32")], rtol=2e-6, atol=2e-7)


@gpu
def test_bubble():
    g = load_golden("g11_bubble_triangle")
    got = _neighbourhood_on_device(5, 5, "bubble", 1.0, False)
    np.testing.assert_array_equal(got, g["bubble_5x5_s1.0_f32"])


@gpu
def test_triangle():
    g = load_golden("g11_bubble_triangle")
    got = _neighbourhood_on_device(5, 5, "triangle", 1.0, False)
    np.testing.assert_allclose(got, g["tri_5x5_s1.0_f32_nc"], rtol=1e-6, atol=1e-7)


@gpu
@pytest.mark.parametrize("XY", [(6, 5), (9, 8)])
def test_gaussian_hex(XY):
    """TestCupySomHex.test_gaussian / test_mexican_hat / test_bubble: the generic neighbourhoods on the hexagonal grid."""
    g = load_golden("g10_hexagonal")
    X, Y = XY
    for sig in (0.8, 2.5):
        for wide in (False, True):
            tag = f"{X}x{Y}_s{sig}_{'f64' if wide else 'f32'}"
            got = _neighbourhood_on_device(X, Y, "gaussian", sig, wide, topology="hexagonal", std_coeff=0.5)
            np.testing.assert_allclose(got, g["gauss_" + tag], rtol=3e-6, atol=1e-7)
            got = _neighbourhood_on_device(X, Y, "gaussian", sig, wide, topology="hexagonal", std_coeff=0.5, compact=True)
            np.testing.assert_allclose(got, g["gausscs_" + tag], rtol=3e-6, atol=1e-7)
            got = _neighbourhood_on_device(X, Y, "mexican_hat", sig, wide, topology="hexagonal", std_coeff=0.5)
            np.testing.assert_allclose(got, g["mex_" + tag], rtol=3e-6, atol=3e-7)


@gpu
def test_mexican_hat_compact_support_tensors():
    """(not in the reference's suite: its compact-support branch of mexican_hat, pinned tensor by tensor by G15)"""
    g = load_golden("g15_mexican_compact")
    for topo, (X, Y) in (("rect", (5, 5)), ("hex", (6, 5)), ("hex", (5, 5))):
        for sig in (0.8, 1.7, 2.5):
            for wide in (False, True):
                got = _neighbourhood_on_device(X, Y, "mexican_hat", sig, wide, compact=True, std_coeff=0.5,
                                               topology="hexagonal" if topo == "hex" else "rectangular")
                ref = g[f"mexcs_{topo}_{X}x{Y}_s{sig}_{'f64' if wide else 'f32'}"]
                np.testing.assert_allclose(got, ref, rtol=3e-6, atol=3e-7)
