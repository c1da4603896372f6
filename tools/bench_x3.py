"""precision='bf16x3' / 'f16x3' / 'f16' against 'f32' and 'bf16' on the configs[2] shard shape (256 x 256 x 128) and on
configs[1] (64 x 64 x 32): epoch time, BMU kernel time and agreement with the float32 BMUs."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.synthetic import gaussian_blobs

def run(X, Y, D, N, epochs=3):
    rs = np.random.RandomState(1234); w = rs.rand(X, Y, D) * 2 - 1; w /= np.linalg.norm(w, axis=-1, keepdims=True)
    data = gaussian_blobs(N, D)
    ref = None
    for prec in ("f32", "f16x3", "bf16x3", "f16", "bf16"):
        e = HipEngine(X, Y, D, precision=prec)
        e.set_weights(w.astype(np.float32)); e.set_data(data)
        e.epoch_accumulate(min(X, Y) / 2, 0.5, True); e.sync()
        bmu = e.epoch_fetch()[2]
        if ref is None: ref = bmu
        e.profile_reset(); e.profile_enable(True)
        t0 = time.perf_counter()
        for i in range(epochs): e.epoch_accumulate(min(X, Y) / 2, 0.5, True)
        e.sync(); dt = (time.perf_counter() - t0) / epochs
        e.profile_enable(False)
        b = e.profile_get("bmu")[0] / epochs
        fl = 2.0 * N * X * Y * D
        print(f"{X}x{Y}x{D} N={N} {prec:7s}: {dt*1e3:8.3f} ms/epoch  bmu {b:8.3f} ms = {fl/(b*1e-3)/1e12:7.1f} TF/s (algorithmic)  "
              f"agree with f32 BMUs {np.mean(bmu == ref)*100:.4f}%", flush=True)

run(64, 64, 32, 100000)
run(256, 256, 128, 262144)
run(64, 64, 784, 60000)
