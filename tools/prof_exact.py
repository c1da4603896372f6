"""One precision mode, a fixed codebook state, a few epochs: the workload of a rocprofv3 kernel trace.
  EX_MODE=exact EX_STATE=0|1|... (epochs of float32 training before the timed epochs) python tools/prof_exact.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.decays import exponential_decay
from xpysom_dask_amd.synthetic import gaussian_blobs

X = Y = int(os.environ.get("EX_SIDE", "256"))
D = int(os.environ.get("EX_D", "128"))
N = int(os.environ.get("EX_ROWS", "65536"))
T = 10
mode = os.environ.get("EX_MODE", "exact")
state = int(os.environ.get("EX_STATE", "0"))
DIST = os.environ.get("EX_DIST", "euclidean")
NEIGH = os.environ.get("EX_NEIGH", "gaussian")
rs = np.random.RandomState(1234)
w = rs.rand(X, Y, D) * 2 - 1
w /= np.linalg.norm(w, axis=-1, keepdims=True)
data = gaussian_blobs(N, D)
if state > 0:
    tr = HipEngine(X, Y, D, precision="exact", distance=DIST, neighborhood=NEIGH)
    tr.set_data(data); tr.set_weights(w.astype(np.float32))
    for t in range(state):
        tr.epoch(exponential_decay(min(X, Y) / 2, 1, t, T), exponential_decay(0.5, 0.01, t, T), True)
    w = tr.get_weights(); tr.close()
e = HipEngine(X, Y, D, precision=mode, distance=DIST, neighborhood=NEIGH)
e.set_data(data); e.set_weights(w.astype(np.float32))
sig, eta = exponential_decay(min(X, Y) / 2, 1, min(state, T - 1), T), exponential_decay(0.5, 0.01, min(state, T - 1), T)
for _ in range(int(os.environ.get("EX_REPS", "8"))):
    e.set_weights(w.astype(np.float32))
    e.epoch_accumulate(sig, eta, True)
e.sync()
