#!/usr/bin/env python3
"""BASELINE configs[0]: the Iris walk-through of the reference (examples/Iris.ipynb) on the MI355X engine.

    python examples/iris.py            # needs libsomhip.so and a GPU

6x6 map, 4 features, 150 samples z-scored, 100 epochs -- the same calls a user of xpysom_dask.XPySom makes; only the
import changes.  The data come from the fixture the parity tests use (tests/golden/g6_iris.npz holds the reference's
iris.csv columns and, for comparison, the reference's own quantization errors for this run).
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from xpysom_dask_amd import XPySom  # noqa: E402   (reference: from xpysom_dask import XPySom)

g = np.load(os.path.join(REPO, "tests", "golden", "g6_iris.npz"))
raw = g["iris_raw"]
data = (raw - raw.mean(axis=0)) / raw.std(axis=0)

for decay in ("linear", "exponential"):
    som = XPySom(6, 6, 4, random_seed=10, decay_function=decay)
    print("%-12s quantization error before training: %.5f" % (decay, som.quantization_error(data)))
    som.train(data, 100)
    qe = som.quantization_error(data)
    print("%-12s after 100 epochs: %.5f   (reference: %.5f)   topographic error %.4f"
          % (decay, qe, float(g[decay + "_default_qe"]), som.topographic_error(data)))
    wins = som.winner(data)
    print("%-12s first five winners: %s" % (decay, wins[:5]))
    hits = som.activation_response(data)
    print("%-12s busiest unit holds %d of 150 samples; %d of 36 units win at least one" % (decay, hits.max(), (hits > 0).sum()))
