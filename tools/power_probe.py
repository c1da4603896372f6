"""Does the hot kernel's time depend on the VALUES it multiplies?  Same launch (256 x 256 x 128, 1 Mi rows -- PP_SIDE / PP_D /
PP_ROWS / PP_DIST for other shapes --, bf16 and the exact
mode's screen), operands random / zero / constant: a kernel at the chip's power limit runs faster on operands that toggle fewer
bits (MI355X_MICROARCH.md, DVFS give-back), a kernel limited by its schedule does not care."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.synthetic import gaussian_blobs

X = Y = int(os.environ.get("PP_SIDE", "256"))
D, N = int(os.environ.get("PP_D", "128")), int(os.environ.get("PP_ROWS", str(1 << 20)))
DIST = os.environ.get("PP_DIST", "euclidean")
rs = np.random.RandomState(1)
cases = {
    "random": (gaussian_blobs(N, D), (rs.rand(X, Y, D) * 2 - 1).astype(np.float32)),
    "zeros": (np.zeros((N, D), np.float32), np.zeros((X, Y, D), np.float32)),
    "ones": (np.ones((N, D), np.float32), np.ones((X, Y, D), np.float32)),
}
out = {}
for prec in ("bf16", "exact"):
    for name, (data, w) in cases.items():
        e = HipEngine(X, Y, D, precision=prec, distance=DIST)
        e.set_data(data); e.set_weights(w)
        for _ in range(3):
            e.epoch_accumulate(64.0, 0.5, True)
        e.sync(); e.profile_reset(); e.profile_enable(True)
        for _ in range(8):
            e.epoch_accumulate(64.0, 0.5, True)
        e.sync(); e.profile_enable(False)
        fam = "screen" if prec == "exact" else "bmu"
        ms = e.profile_get(fam)[0] / max(1, e.profile_get(fam)[1])
        tf = 2.0 * N * X * Y * D / (ms * 1e-3) / 1e12
        out[(prec, name)] = ms
        print("%-5s %-6s: %-6s kernel %.3f ms = %.0f TFLOP/s = %.3f of 2.5 PFLOP/s" % (prec, name, fam, ms, tf, tf / 2500), flush=True)
        e.close()
