"""The canary (som_set_verify / SOM_VERIFY=n): n strided rows of every BMU launch are scored again by an independent
float32 route (csrc/bmu_exact.hpp, verify_best_kernel: vector-ALU fmaf chains straight from the float32 codebook) and
the launch's own picks must be the float32 picks, or within the precision mode's bound of them.  It must stay silent on
healthy launches of every kernel family and speak up when the operand images the kernels read are damaged behind the
library's back (som_debug_corrupt_operands: what a lost staging copy would look like).  GPU only (`-m gpu`)."""
import numpy as np
import pytest

from oracle import som_oracle as O

pytestmark = pytest.mark.gpu
F32 = np.float32

# every BMU kernel family: float32 resident / tiled, exact (screen + re-score, resident and wide), 16x16x32 resident, wide, tiled
CASES = [("f32", 20, 24, 32, "euclidean"), ("f32", 12, 12, 200, "euclidean"), ("f32", 16, 16, 40, "cosine"),
         ("exact", 64, 64, 32, "euclidean"), ("exact", 30, 30, 128, "euclidean"), ("bf16", 64, 64, 96, "euclidean"),
         ("f16", 24, 20, 17, "euclidean"), ("exact", 64, 64, 150, "euclidean"),
         ("bf16", 64, 66, 200, "cosine"), ("bf16", 12, 12, 300, "euclidean")]


def make(prec, X, Y, D, dist):
    from xpysom_dask_amd.engine import HipEngine
    n = 3000
    data = O.gaussian_blobs(n, D, seed=D)
    w = O.default_codebook(X, Y, D, 4).astype(F32) * 2
    if dist == "cosine":
        data, w = np.abs(data), np.abs(w)
    e = HipEngine(X, Y, D, precision=prec, distance=dist)
    e.set_weights(w)
    e.set_data(data)
    return e, data


@pytest.mark.parametrize("prec,X,Y,D,dist", CASES)
def test_canary_is_silent_on_healthy_launches(prec, X, Y, D, dist):
    e, data = make(prec, X, Y, D, dist)
    e.set_verify(96)
    for sig in (6.0, 2.0, 1.0):
        e.epoch(sig, 0.4, True)                            # a few epochs: the map changes under the canary
    e.bmu(data[:500])
    e.stream_epoch_accumulate([data[:1000], data[1000:]], 1.0, 0.2, True)
    launches, rows = e.verify_stats()
    assert launches == 6 and rows == 96 * 6
    e.close()


@pytest.mark.parametrize("prec,X,Y,D,dist", CASES)
def test_canary_catches_damaged_operand_images(prec, X, Y, D, dist):
    from xpysom_dask_amd.engine import SomHipError
    e, data = make(prec, X, Y, D, dist)
    e.epoch_accumulate(3.0, 0.4, True)                     # healthy first
    good = e.epoch_fetch()[2]
    e.set_verify(128)
    e.epoch_accumulate(3.0, 0.4, True)
    # exact mode: the float32 image alone.  With the screen's image zeroed too every group ties, every row overflows to the
    # float32 fallback, and in patch order the fallback rebuilds its image from the codebook first: that damage heals
    # itself instead of showing (nothing wrong comes out); zeroed re-score operands under a healthy screen do show.
    e.debug_corrupt_operands(2 if prec == "exact" else 3)
    with pytest.raises(SomHipError, match="SOM_VERIFY"):
        e.epoch_accumulate(3.0, 0.4, True)
    # the engine is usable afterwards: a codebook upload rebuilds the images
    e.set_weights(e.get_weights())
    e.epoch_accumulate(3.0, 0.4, True)
    assert np.array_equal(e.epoch_fetch()[2], good)
    e.close()


def test_canary_from_the_environment(monkeypatch):
    monkeypatch.setenv("SOM_VERIFY", "32")
    from xpysom_dask_amd import XPySom
    data = O.gaussian_blobs(800, 6, seed=1)
    som = XPySom(9, 9, 6, random_seed=1, precision="exact").train(data, 3)
    assert som._engine().verify_stats() == (3, 96)


def test_wide_kernel_hand_placed_lds_reads_equal_the_plain_reads(tmp_path):
    """The wide BMU kernel (more than 128 features on big maps: bf16 / f16 modes and the exact mode's screen) reads its LDS
    fragments with inline-assembly `ds_read_b128` and hand-counted waits (csrc/bmu_bf16_wide.hpp).  That is only as good as
    the register allocation around it, so a second library is built here with plain C++ reads (-DSOM_WIDE_PLAIN_READS:
    what build.py gives any compiler but the validated one) and both must return the same ids, bit for bit, on the same
    rows -- the bf16 kernel, the f16 kernel and the exact mode's screen + re-score."""
    import json, os, subprocess, sys
    from xpysom_dask_amd import build as B
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    plain = str(tmp_path / "libsomhip_plain.so")
    B.build(force=True, verbose=False, extra=("-DSOM_WIDE_PLAIN_READS",), out=plain)
    prog = r"""
import sys, json, zlib, numpy as np
sys.path.insert(0, %r)
from oracle import som_oracle as O
from xpysom_dask_amd.engine import HipEngine
out = {}
for D, dist in ((200, "euclidean"), (784, "cosine")):
    data = np.abs(O.gaussian_blobs(6000, D, seed=D))
    w = np.abs(O.default_codebook(64, 64, D, 3)).astype(np.float32)
    for prec in ("bf16", "f16", "exact"):
        e = HipEngine(64, 64, D, precision=prec, distance=dist)
        e.set_weights(w); e.set_data(data)
        e.epoch_accumulate(6.0, 0.3, True)
        out["%%d_%%s_%%s" %% (D, dist, prec)] = int(zlib.crc32(e.epoch_fetch()[2].tobytes()))
        e.close()
print(json.dumps(out))
""" % REPO
    res = {}
    for tag, lib in (("asm", None), ("plain", plain)):
        env = dict(os.environ)
        env.pop("SOM_LIB_PATH", None)
        if lib:
            env["SOM_LIB_PATH"] = lib
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["asm"] == res["plain"], (res["asm"], res["plain"])
