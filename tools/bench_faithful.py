#!/usr/bin/env python3
"""The update in its two forms at the configs[2] shape: bucketed (what trains) against faithful (g^T x as one K x N x D
float32 MFMA GEMM with g generated in flight, som_epoch_accumulate_faithful).  Prints ms per accumulate for both."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload_rows  # noqa: E402
from xpysom_dask_amd.engine import HipEngine  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
e = HipEngine(256, 256, 128, precision="bf16")
rs = np.random.RandomState(1234)
w = rs.rand(256, 256, 128) * 2 - 1
w /= np.linalg.norm(w, axis=-1, keepdims=True)
e.set_weights(w.astype(np.float32))
e.set_data(workload_rows("c3", rows, 1234))
for name, f, reps in (("bucketed", e.epoch_accumulate, 10), ("faithful", e.epoch_accumulate_faithful, 2)):
    f(20.0, 0.3, True); e.sync()
    e.profile_reset(); e.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        f(20.0, 0.3, True)
    e.sync()
    dt = (time.perf_counter() - t0) / reps
    e.profile_enable(False)
    upd = (e.profile_get("segsum")[0] + e.profile_get("kron")[0]) / reps
    flops = 2.0 * rows * 65536 * 129 if name == "faithful" else 2.0 * 65536 * 512 * 129
    print("%s: %d rows, accumulate %.3f ms of which update %.3f ms (%.1f TFLOP/s on %.3g flop)" % (name, rows, dt * 1e3, upd, flops / (upd * 1e-3) / 1e12, flops))
