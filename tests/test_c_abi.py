"""The boundary is a C ABI: examples/c_caller.c (plain C, include/somhip.h, no Python) is compiled with gcc against
libsomhip.so.  Without a GPU it must stop at som_create with the library's message; on a GPU box its trained codebook
and BMUs must be the Python host's, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from oracle import som_oracle as O
from tests.conftest import REPO, _gpu_present


def _build(tmp_path):
    exe = str(tmp_path / "c_caller")
    libdir = os.path.join(REPO, "xpysom_dask_amd")
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "examples", "c_caller.c"),
           "-o", exe, "-L", libdir, "-lsomhip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def _inputs(tmp_path, X, Y, D, n, T):
    data = O.gaussian_blobs(n, D, seed=8)
    w = O.default_codebook(X, Y, D, 4).astype(np.float32)
    data.tofile(tmp_path / "rows.f32")
    w.tofile(tmp_path / "w.f32")
    # the reference's default schedule (exponential decay of sigma = min(X, Y) / 2 -> 1 and of 0.5 -> 0.01), computed
    # where the reference computes it: on the host
    sched = np.array([[O.exponential_decay(min(X, Y) / 2, 1, t, T), O.exponential_decay(0.5, 0.01, t, T)] for t in range(T)])
    sched.astype(np.float64).tofile(tmp_path / "sched.f64")
    return data, w


@pytest.mark.skipif(_gpu_present(), reason="checks the no-GPU failure mode")
def test_c_caller_compiles_and_fails_loudly_without_a_gpu(tmp_path):
    exe = _build(tmp_path)
    _inputs(tmp_path, 6, 5, 4, 50, 2)
    r = subprocess.run([exe, str(tmp_path / "rows.f32"), str(tmp_path / "w.f32"), "6", "5", "4", "50", str(tmp_path / "sched.f64"),
                        str(tmp_path / "ow"), str(tmp_path / "ob")], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_c_caller_equals_the_python_host(tmp_path):
    from xpysom_dask_amd import XPySom
    X, Y, D, n, T = 14, 11, 9, 3000, 5
    exe = _build(tmp_path)
    data, w = _inputs(tmp_path, X, Y, D, n, T)
    r = subprocess.run([exe, str(tmp_path / "rows.f32"), str(tmp_path / "w.f32"), str(X), str(Y), str(D), str(n), str(tmp_path / "sched.f64"),
                        str(tmp_path / "ow"), str(tmp_path / "ob")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    som = XPySom(X, Y, D, random_seed=1)                       # exponential decay, sigma = min(X, Y) / 2: the defaults
    som._weights = w.copy()
    som.train(data, T)
    got_w = np.fromfile(tmp_path / "ow", dtype=np.float32).reshape(X, Y, D)
    got_b = np.fromfile(tmp_path / "ob", dtype=np.int32)
    assert np.array_equal(got_w, som._weights)
    assert np.array_equal(got_b, np.array([i * Y + j for i, j in som.winner(data)], dtype=np.int32))
    assert "quantization error %.6f" % som.quantization_error(data) in r.stdout
