"""A short run of the tiled bf16 kernel on a configs[4]-shaped shard, for rocprofv3 --pmc passes (tools/pmc_c5.sh)."""
import sys, numpy as np
sys.path.insert(0, '.')
from xpysom_dask_amd.engine import HipEngine
from xpysom_dask_amd.synthetic import gaussian_blobs
X, Y, D, N = 512, 512, 784, 65536
if len(sys.argv) > 4: X, Y, D, N = [int(a) for a in sys.argv[1:5]]
e = HipEngine(X, Y, D, precision="bf16", distance="cosine", neighborhood="mexican_hat")
rs = np.random.RandomState(1234); w = np.abs(rs.rand(X, Y, D)).astype(np.float32)
e.set_weights(w); e.set_data(np.abs(gaussian_blobs(N, D)))
for i in range(2): e.epoch(min(X, Y) / 2, 0.4, True)
e.sync()
print("done")
