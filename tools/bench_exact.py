"""precision='exact' against 'f32', 'f16' and 'bf16' at the north-star batch (256 x 256 x 128, 65 536 rows): epoch
time, BMU time, agreement with the float32 BMUs and the re-score's load (candidate groups per row, fallback rows) on
the seeded codebook and on the codebooks of a float32-trained schedule (smooth early maps are the hard case)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.decays import exponential_decay
from xpysom_dask_amd.synthetic import gaussian_blobs

X = Y = int(os.environ.get("EX_SIDE", "256"))
D = int(os.environ.get("EX_D", "128"))
N = int(os.environ.get("EX_ROWS", "65536"))
T = int(os.environ.get("EX_EPOCHS", "10"))
MODES = os.environ.get("EX_MODES", "f32,exact,f16,bf16").split(",")
DIST = os.environ.get("EX_DIST", "euclidean")
NEIGH = os.environ.get("EX_NEIGH", "gaussian")


def main():
    rs = np.random.RandomState(1234)
    w = rs.rand(X, Y, D) * 2 - 1
    w /= np.linalg.norm(w, axis=-1, keepdims=True)
    w = w.astype(np.float32)
    data = gaussian_blobs(N, D)
    eng = {m: HipEngine(X, Y, D, precision=m, distance=DIST, neighborhood=NEIGH) for m in MODES}
    for e in eng.values():
        e.set_data(data)
    trainer = eng["f32"]
    trainer.set_weights(w)
    for t in range(T + 1):
        if t in (0, 1, 2, T // 2, T):
            wt = trainer.get_weights()
            sig, eta = exponential_decay(min(X, Y) / 2, 1, min(t, T - 1), T), exponential_decay(0.5, 0.01, min(t, T - 1), T)
            ref = None
            for m, e in eng.items():
                if m != "f32" or True:
                    e.set_weights(wt)
                e.epoch_accumulate(sig, eta, True); e.sync()
                bmu = e.epoch_fetch()[2]
                if ref is None:
                    ref = bmu
                e.profile_reset(); e.profile_enable("bmu")
                t0 = time.perf_counter()
                for _ in range(5):
                    e.epoch_accumulate(sig, eta, True)
                e.sync(); dt = (time.perf_counter() - t0) / 5
                e.profile_enable(False)
                b = e.profile_get("bmu")[0] / 5
                extra = ""
                if m == "exact":
                    c = e.exact_last_counts(min(N, 65536))
                    rows, fb, passes = e.exact_stats()
                    extra = "  cand groups/row mean %.2f p50 %d p99 %d max %d; fallback rows %d of %d" % (
                        c.mean(), np.percentile(c, 50), np.percentile(c, 99), c.max(), fb, rows)
                print("state after %2d epochs  %-7s: epoch %7.3f ms  bmu %7.3f ms  = f32 BMUs on %.4f%%%s" % (
                    t, m, dt * 1e3, b, 100.0 * np.mean(bmu == ref), extra), flush=True)
            trainer.set_weights(wt)
        if t < T:
            sig, eta = exponential_decay(min(X, Y) / 2, 1, t, T), exponential_decay(0.5, 0.01, t, T)
            trainer.epoch(sig, eta, True)


if __name__ == "__main__":
    main()
