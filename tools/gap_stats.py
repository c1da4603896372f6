#!/usr/bin/env python3
"""How many units sit within the bf16 operand-rounding band of a row's best unit?  (VERDICT r1 item 5:
sizes the 'bf16 screening + float32 re-score' idea before building it.)

Trains the configs[2] map for a few epochs (bf16, on the GPU), then for a sample of rows takes the exact float32
(n, K) distance matrix (som_distance_matrix) and counts, per row, the units whose squared distance lies within
tau of the minimum, for several definitions of tau:
  cs    rigorous Cauchy-Schwarz bound of the bf16 operand rounding: 2 * (|dx| |w|max + |x| |dw|max) * 2
  stat8 eight standard deviations of the rounding noise of x~.w~ (independent uniform roundings)
  r1    the bound the round-1 tests use: 2^-8 (|x| + |w|max), squared form
and the share of rows whose bf16 pick differs from the float32 pick.
"""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload_rows  # noqa: E402
from xpysom_dask_amd.decays import exponential_decay  # noqa: E402
from xpysom_dask_amd.engine import HipEngine  # noqa: E402

X = Y = int(os.environ.get("GAP_SIDE", "256"))
D, N, EPOCHS, SAMPLE = 128, 1 << 18, int(os.environ.get("GAP_EPOCHS", "10")), 4096


def bf16_round(a):
    u = a.astype(np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def main():
    data = workload_rows("c3", N, 1234)
    rs = np.random.RandomState(1234)
    w = rs.rand(X, Y, D) * 2 - 1
    w /= np.linalg.norm(w, axis=-1, keepdims=True)
    w = w.astype(np.float32)
    bf = HipEngine(X, Y, D, precision="bf16")
    f32 = HipEngine(X, Y, D, precision="f32")
    bf.set_data(data)
    for t in range(EPOCHS + 1):
        if t in (0, 1, EPOCHS // 2, EPOCHS):
            wt = w if t == 0 else bf.get_weights()
            bf.set_weights(wt)
            f32.set_weights(wt)
            x = data[:SAMPLE]
            dm = np.concatenate([f32.distance_matrix(x[s:s + 2048]) for s in range(0, SAMPLE, 2048)])   # -2xw + |w|^2
            best = dm.min(1)
            pick_f32 = dm.argmin(1)
            pick_bf = bf.bmu(x)
            wf = wt.reshape(-1, D)
            xn, wn = np.linalg.norm(x, axis=1), np.linalg.norm(wf, axis=1)
            dx = np.linalg.norm(x - bf16_round(x), axis=1)
            dw = np.linalg.norm(wf - bf16_round(wf), axis=1)
            tau = {
                "cs": 4.0 * (dx * wn.max() + xn * dw.max()),
                "stat8": 2.0 * 8.0 * 2.0 ** -9 / np.sqrt(3.0) * np.sqrt(2.0) * xn * wn.max() / np.sqrt(D) * 2.0,
                "r1": (2.0 ** -8 * (xn + wn.max())) ** 2 + 2 * 2.0 ** -8 * (xn + wn.max()) * np.sqrt(np.maximum(best + xn ** 2, 0)),
            }
            line = "epoch %2d: bf16 != f32 on %.2f%% of rows; " % (t, 100.0 * (pick_bf != pick_f32).mean())
            for name, tv in tau.items():
                within = (dm <= (best + tv)[:, None]).sum(1)
                line += "%s: tau/dmin %.3g, units within: median %d, p90 %d, max %d, rows with >1: %.1f%%; " % (
                    name, float(np.median(tv / np.maximum(best + xn ** 2, 1e-9))), int(np.median(within)),
                    int(np.percentile(within, 90)), int(within.max()), 100.0 * (within > 1).mean())
            # rank of the float32 winner in the bf16 ordering proxy: how deep a candidate list must be
            print(line, flush=True)
        if t < EPOCHS:
            sig = exponential_decay(min(X, Y) / 2, 1, t, EPOCHS)
            eta = exponential_decay(0.5, 0.01, t, EPOCHS)
            bf.epoch(sig, eta, True)


if __name__ == "__main__":
    main()
