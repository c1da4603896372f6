"""Streamed (out-of-core) epoch throughput at the C3 shape: pageable vs pinned double-buffered chunks."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.synthetic import gaussian_blobs
X, Y, D, CH, NCH = 256, 256, 128, 1 << 19, 8
e = HipEngine(X, Y, D, precision="bf16")
rs = np.random.RandomState(1234); w = rs.rand(X, Y, D) * 2 - 1
e.set_weights(w.astype(np.float32))
src = gaussian_blobs(CH, D)
pinned = [e.pinned_empty((CH, D)), e.pinned_empty((CH, D))]
for b in pinned: b[:] = src
def run(chunks, label):
    e.stream_epoch_accumulate(chunks(), 64.0, 0.4, True); e.epoch_merge(); e.sync()
    t0 = time.perf_counter()
    e.stream_epoch_accumulate(chunks(), 64.0, 0.4, True); e.epoch_merge(); e.sync()
    dt = time.perf_counter() - t0
    print(f"{label}: {NCH*CH/dt/1e6:.1f} M samples/s  ({dt*1e3:.1f} ms for {NCH} chunks of {CH} rows, {NCH*CH*D*4/dt/1e9:.1f} GB/s host->HBM)")
run(lambda: (src for _ in range(NCH)), "pageable chunks")
run(lambda: (pinned[i & 1] for i in range(NCH)), "pinned double-buffered chunks")
e.set_data(src); e.epoch(64.0, 0.4, True); e.sync()
t0 = time.perf_counter(); e.epoch(64.0, 0.4, True); e.sync(); dt = time.perf_counter() - t0
print(f"resident: {CH/dt/1e6:.1f} M samples/s")
