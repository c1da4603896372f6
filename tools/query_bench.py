"""The query path (winner / quantization_error on rows that live in HBM) on trained maps: som_bmu_device /
som_quantization_error_device over QB_ROWS device-resident rows of configs[2]'s map in the states after the listed epochs of
the 25-epoch schedule -- under the scout's plan (default) and with SOM_EXACT_SKIP=0 (every block), ids compared with each
other and with precision 'f32'.
    QB_ROWS=1048576 QB_EPOCHS=2,5,12,24 python tools/query_bench.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpysom_dask_amd.decays import exponential_decay
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.synthetic import gaussian_blobs

X = Y = 256; D = 128
N = int(os.environ.get("QB_ROWS", str(1 << 20))); T = 25
EPOCHS = [int(v) for v in os.environ.get("QB_EPOCHS", "2,5,12,24").split(",")]
CHECK_F32 = os.environ.get("QB_F32", "1") != "0"
data = gaussian_blobs(N, D, seed=1234, centre_seed=1234)
probe = gaussian_blobs(N, D, seed=99, centre_seed=1234)            # other rows of the same mixture
rs = np.random.RandomState(1234)
w = rs.rand(X, Y, D) * 2 - 1; w /= np.linalg.norm(w, axis=-1, keepdims=True)
tr = HipEngine(X, Y, D, precision="exact"); tr.set_data(data); tr.set_weights(w.astype(np.float32))
dev = torch.from_numpy(probe).cuda(); torch.cuda.synchronize()
os.environ["SOM_EXACT_SKIP"] = "0"
full = HipEngine(X, Y, D, precision="exact")
del os.environ["SOM_EXACT_SKIP"]
plan = HipEngine(X, Y, D, precision="exact")
f32 = HipEngine(X, Y, D, precision="f32") if CHECK_F32 else None


def timed(e, fn, reps=5):
    fn(); e.sync()
    e.profile_reset(); e.profile_enable("bmu")
    s0 = e.exact_skip_stats(); t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    e.sync(); dt = (time.perf_counter() - t0) / reps
    e.profile_enable(False); s1 = e.exact_skip_stats()
    return out, 1e3 * dt, e.profile_get("bmu")[0] / reps, (s1[0] - s0[0]) / max(1, s1[1] - s0[1])


for t in range(T):
    sig, eta = exponential_decay(128, 1, t, T), exponential_decay(0.5, 0.01, t, T)
    tr.epoch(sig, eta, True)
    if t in EPOCHS:
        wt = tr.get_weights()
        for e in (full, plan) + ((f32,) if f32 else ()):
            e.set_weights(wt)
        a, wall_a, k_a, _ = timed(full, lambda: full.bmu_device(dev.data_ptr(), N))
        b, wall_b, k_b, sh = timed(plan, lambda: plan.bmu_device(dev.data_ptr(), N))
        qa, qwall_a, qk_a, _ = timed(full, lambda: full.quantization_error_device(dev.data_ptr(), N))
        qb, qwall_b, qk_b, _ = timed(plan, lambda: plan.quantization_error_device(dev.data_ptr(), N))
        same = bool(np.array_equal(a, b))
        same32 = bool(np.array_equal(f32.bmu_device(dev.data_ptr(), N), b)) if f32 else None
        print("map after epoch %2d: winner ids of %d device rows: full scan %.3f ms (BMU search on the stream %.3f) | planned %.3f ms (%.3f), executed share %.4f | "
              "ids equal %s, equal to f32 %s | quantization_error %.3f -> %.3f ms (qe %.6f vs %.6f)" % (
                  t, N, wall_a, k_a, wall_b, k_b, sh, same, same32, qwall_a, qwall_b, qa, qb), flush=True)
        assert same and same32 is not False and abs(qa - qb) <= 1e-6 * qa
print("scout stats (scouted launches, transient launches under a plan):", plan.exact_scout_stats())
