"""How stable is the patch-sorted order of the resident rows from one epoch to the next?

The exact mode visits rows in the order of their last BMU's 8 x 8 patch (csrc/exact_skip.hpp).  This probe trains the
benchmark's schedule and reports, per epoch: the wall time, the share of the distance GEMM's blocks the screen ran, and
how many rows changed their BMU / their BMU's patch against the epoch before -- the measurement behind keeping the
sorted pass resident across epochs.
    RP_T=25 RP_ROWS=1048576 python tools/resid_probe.py
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpysom_dask_amd.decays import exponential_decay
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.synthetic import gaussian_blobs

X = Y = int(os.environ.get("RP_SIDE", "256"))
D = int(os.environ.get("RP_D", "128"))
N = int(os.environ.get("RP_ROWS", str(1 << 20)))
T = int(os.environ.get("RP_T", "25"))
UNSTRUCTURED = os.environ.get("RP_UNSTRUCTURED", "0") == "1"

rs = np.random.RandomState(1234)
w = rs.rand(X, Y, D) * 2 - 1
w /= np.linalg.norm(w, axis=-1, keepdims=True)
data = (np.random.default_rng(1234).standard_normal((N, D)).astype(np.float32) if UNSTRUCTURED
        else gaussian_blobs(N, D, seed=1234, centre_seed=1234))
e = HipEngine(X, Y, D, precision="exact")
e.set_weights(w.astype(np.float32))
e.set_data(data)


def patch(ids):
    return (ids // Y // 8) * (Y // 8) + (ids % Y) // 8


prev = None
out = []
for t in range(T):
    sig, eta = exponential_decay(min(X, Y) / 2, 1, t, T), exponential_decay(0.5, 0.01, t, T)
    s0 = e.exact_skip_stats()
    e.sync()
    t0 = time.perf_counter()
    e.epoch(sig, eta, True)
    e.sync()
    ms = 1e3 * (time.perf_counter() - t0)
    s1 = e.exact_skip_stats()
    _, _, ids = e.epoch_fetch(True)
    row = {"epoch": t, "sigma": float(sig), "ms": ms, "share": (s1[0] - s0[0]) / max(1, s1[1] - s0[1]),
           "groups_per_row": float(e.exact_last_counts(min(N, 1 << 18)).mean())}
    if prev is not None:
        row["unit_changed"] = float(np.mean(ids != prev))
        row["patch_changed"] = float(np.mean(patch(ids) != patch(prev)))
        # ... and by how far: Chebyshev distance between the old and the new patch on the map
        pa, pb = patch(ids), patch(prev)
        dd = np.maximum(np.abs(pa // (Y // 8) - pb // (Y // 8)), np.abs(pa % (Y // 8) - pb % (Y // 8)))
        row["patch_moved_more_than_1"] = float(np.mean(dd > 1))
    prev = ids
    out.append(row)
    print(json.dumps(row), flush=True)
