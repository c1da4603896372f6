"""An engine with the HipEngine interface backed by the NumPy oracle -- TEST DOUBLE ONLY.
It lets the CPU suite drive the product's host logic (XPySom.train, shard split,
all-reduce, merge ordering) under gloo without a GPU.  The product never constructs it."""
import numpy as np

from oracle import som_oracle as O


class OracleEngine:
    def __init__(self, x, y, input_len, *, distance="euclidean", neighborhood="gaussian",
                 std_coeff=0.5, compact_support=False, precision="f32", topology="rectangular", norm_p=0, device=0):
        self.x, self.y, self.D = x, y, input_len
        self.K = x * y
        if topology == "hexagonal":
            neighborhood += "_hex"
        self.kw = dict(distance=distance, neighbourhood=neighborhood, std_coeff=std_coeff, compact=compact_support)
        self.W = np.zeros((self.K, self.D), np.float32)
        self.acc = np.zeros((self.K, self.D + 1), np.float32)
        self.data = np.zeros((0, self.D), np.float32)
        self.n_rows = 0

    def set_weights(self, w):
        self.W = np.array(w, dtype=np.float32).reshape(self.K, self.D)

    def get_weights(self):
        return self.W.copy()

    def set_data(self, data):
        self.data = np.ascontiguousarray(data, dtype=np.float32)
        self.n_rows = len(self.data)

    def epoch_accumulate(self, sigma, eta, neigh_f64):
        self.acc[:] = 0
        if self.n_rows:
            W3 = self.W.reshape(self.x, self.y, self.D)
            _, num, den = O.update(self.data, W3, eta, sigma, wide=bool(neigh_f64), **self.kw)
            self.acc[:, :self.D] = num.reshape(self.K, self.D)
            self.acc[:, self.D] = den.reshape(self.K)

    def stream_epoch_accumulate(self, chunks, sigma, eta, neigh_f64):
        self.set_data(np.concatenate([np.asarray(c, np.float32) for c in chunks]))
        self.epoch_accumulate(sigma, eta, neigh_f64)

    # the staged form of the epoch (include/somhip.h: som_epoch_accumulate_begin / _block): blocks of `BLOCK_ROWS`
    # map rows (128 in the engine; small here so that a test map has several)
    BLOCK_ROWS = 3

    def epoch_accumulate_begin(self, sigma, eta, neigh_f64):
        self.epoch_accumulate(sigma, eta, neigh_f64)
        self._staged = self.acc.copy()
        self.acc[:] = np.nan                           # a block is final only after epoch_accumulate_block
        self._next = 0

    def epoch_block_count(self):
        return -(-self.x // self.BLOCK_ROWS)

    def epoch_accumulate_block(self, b):
        assert b == self._next
        self._next += 1
        lo = b * self.BLOCK_ROWS * self.y
        hi = min(self.K, (b + 1) * self.BLOCK_ROWS * self.y)
        self.acc[lo:hi] = self._staged[lo:hi]
        return lo * (self.D + 1), (hi - lo) * (self.D + 1)

    def accum_tensor(self):
        import torch
        return torch.from_numpy(self.acc.reshape(-1))

    def epoch_merge(self):
        den = self.acc[:, self.D:self.D + 1]
        with np.errstate(all="ignore"):
            self.W = np.where(den != 0, self.acc[:, :self.D] / den, self.W).astype(np.float32)

    def sync(self):
        pass

    def bmu(self, x, quantization=False):
        W3 = self.W.reshape(self.x, self.y, self.D)
        if quantization:
            return O.quantization_ids(np.asarray(x, np.float32), W3).astype(np.int32)
        return O.winner_ids(np.asarray(x, np.float32), W3, self.kw["distance"]).astype(np.int32)

    def bmu_f64(self, x):
        """float64 rows against the float32 codebook: NumPy's own float64 arithmetic (xpysom.py:379-396)."""
        return O.winner_ids(np.asarray(x, np.float64), self.W.reshape(self.x, self.y, self.D), self.kw["distance"]).astype(np.int32)

    def quantization_error(self, x):
        return O.quantization_error(x, self.W.reshape(self.x, self.y, self.D))

    def bmu_top2(self, x):
        b = O.top2_ids(np.asarray(x, np.float32), self.W.reshape(self.x, self.y, self.D))
        return b[:, 0].astype(np.int32), b[:, 1].astype(np.int32)

    def distance_matrix(self, x, quantization=False):
        """activate (the configured GEMM-form distance) / distance_from_weights (the full Euclidean distance)."""
        x = np.asarray(x, np.float32)
        if quantization:
            return O.dist_euclid(x, self.W)
        return O.DISTANCES[self.kw["distance"]](x, self.W)

