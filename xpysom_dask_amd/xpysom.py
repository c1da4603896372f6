"""XPySom -- the reference's class surface over the MI355X HIP engine.

Drop-in for ``xpysom_dask.XPySom`` (reference xpysom_dask/xpysom.py:72-892) on its hot path:
``__init__``, ``train``, ``winner``, ``quantization_error`` (+ the thin consumers of those).
The host keeps what the reference keeps on the host -- argument validation and error
strings, the seeded default codebook, the sigma/eta schedules, the epoch loop, result
formatting, pickling -- and hands everything below ``_update``/``_winner``/``_merge_updates``
to libsomhip through ctypes.  There is no ``xp=`` dispatch and no CPU path: without the HIP
library (or a GPU) the compute methods raise.
"""
from collections import Counter, defaultdict
from warnings import warn

import numpy as np

from . import distributed as _dist
from .decays import DECAY_FUNCTIONS

TOPOLOGIES = ("hexagonal", "rectangular")
# name -> implemented in the HIP engine?   (registry order follows xpysom.py:260-283)
NEIGHBORHOODS = {"gaussian": True, "mexican_hat": True, "bubble": True, "triangle": True}
# distances.py:162-170 registry; False = valid reference name the engine does not cover yet
DISTANCES = {"euclidean": True, "euclidean_no_opt": True, "manhattan": True, "manhattan_no_opt": True,
             "cosine": True, "norm_p": True, "norm_p_no_opt": True}
DEFAULT_BATCH_ROWS = 65536


def _device_rows(data):
    """Rows that already live in HBM (the role CuPy arrays play in the reference, xpysom.py:487-510):
    a torch CUDA tensor, or any object with ``__cuda_array_interface__``, as a contiguous float32
    ``(n, input_len)`` block.  Returns ``(pointer, n_rows, n_cols, device_index, owner, stream)`` or ``None``
    for host data; ``stream`` is the producer's stream in ``__cuda_array_interface__`` terms (``None``: the
    producer named none -- the engine then waits for the whole device; ``"done"``: already waited for)."""
    torch = None
    if type(data).__module__.split(".")[0] == "torch":   # (torch is imported only when a tensor is actually passed)
        import torch
    if torch is not None and isinstance(data, torch.Tensor):
        if not data.is_cuda:
            return None
        t = data.detach()
        if t.dim() != 2:
            raise ValueError('data must be 2-dimensional (n_samples, input_len)')
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(torch.float32).contiguous()
        torch.cuda.current_stream(t.device).synchronize()      # the engine reads it on its own stream
        return t.data_ptr(), t.shape[0], t.shape[1], t.device.index, t, "done"
    cai = getattr(data, "__cuda_array_interface__", None)
    if cai is not None:
        shape = tuple(cai["shape"])
        if len(shape) != 2 or cai["typestr"] != "<f4" or cai.get("strides") is not None:
            raise ValueError('device data must be a C-contiguous float32 (n_samples, input_len) array')
        return int(cai["data"][0]), shape[0], shape[1], None, data, cai.get("stream")
    return None


def _host_rows(x, engine=None):
    """float32 host view of array-like input; device rows are copied back through ``engine()`` (analysis
    calls only; the engine is not created for host input)."""
    if hasattr(x, "is_cuda") and getattr(x, "is_cuda"):
        x = x.detach().cpu().numpy()
    elif getattr(x, "__cuda_array_interface__", None) is not None:
        ptr, n, d, _, _, stream = _device_rows(x)
        if engine is None:
            raise TypeError("device rows need an engine to be copied back to the host")
        eng = engine()
        eng.sync_producer(stream)
        return eng.copy_to_host(ptr, (n, d))
    return np.asarray(x, dtype=np.float32)


class XPySom:
    def __init__(self, x, y, input_len,
                 sigma=0, sigmaN=1,
                 learning_rate=0.5, learning_rateN=0.01, decay_function='exponential',
                 neighborhood_function='gaussian', std_coeff=0.5,
                 topology='rectangular',
                 activation_distance='euclidean',
                 activation_distance_kwargs={},
                 random_seed=None, n_parallel=0, compact_support=False,
                 xp=None,
                 use_dask=False, dask_chunks='auto',
                 *, precision='exact', device=None, sharded_input=False, shard='contiguous'):
        """Same positional/keyword surface as the reference constructor (xpysom.py:73-82).

        ``xp``, ``use_dask`` and ``dask_chunks`` are accepted for source compatibility and
        ignored: there is one backend (HIP) and multi-GPU runs use torch.distributed, not Dask.
        Extra keyword-only arguments:
          precision      'exact' (the DEFAULT: the BMUs of 'f32' bit for bit -- hence its accumulators and its trained
                         codebook --, found by an IEEE-half MFMA screen and a float32 re-score of the units its error
                         bound cannot rule out; where no screen applies -- small maps, the VALU distances -- the float32
                         kernels themselves serve it), 'f32' (the exact-float32 MFMA kernels over every unit: the parity
                         mode the default is checked against), 'bf16' (bf16 MFMA distance GEMM: BMUs within the operand
                         rounding of float32's), or the same path on IEEE half operands, 'f16' (three more mantissa bits
                         at the same MFMA rate; rows and units must fit float16: norms <= 65504)
          device         HIP device ordinal (default: LOCAL_RANK or 0)
          sharded_input  under an initialised process group: ``train(data)`` receives only this
                         rank's rows (default: every rank passes the full array and takes its slice)
          shard          which slice that is: 'contiguous' (rows [lo, hi): the reference's Dask blocks, xpysom.py:490,546)
                         or 'strided' (rows rank, rank + world, ...: every rank a sample of the whole file, so that an
                         order in the input cannot become rank skew under block skipping); the same map either way, to
                         float32 summation order
        """
        if sigma >= x or sigma >= y:
            warn('Warning: sigma is too high for the dimension of the map.')

        self._random_generator = np.random.RandomState(random_seed)
        self.use_dask = False
        self.dask_chunks = dask_chunks

        self._learning_rate = learning_rate
        self._learning_rateN = learning_rateN
        self._sigma = min(x, y) / 2 if sigma == 0 else sigma
        self._std_coeff = std_coeff
        self._sigmaN = sigmaN
        self._input_len = input_len

        # seeded default codebook, float64 on the host exactly as xpysom.py:189-190
        self._weights = self._random_generator.rand(x, y, input_len) * 2 - 1
        self._weights /= np.linalg.norm(self._weights, axis=-1, keepdims=True)

        self._neigx = np.arange(x)
        self._neigy = np.arange(y)

        if topology not in TOPOLOGIES:
            msg = '%s not supported only hexagonal and rectangular available'
            raise ValueError(msg % topology)
        self.topology = topology

        if decay_function not in DECAY_FUNCTIONS:
            msg = '%s not supported. Functions available: %s'
            raise ValueError(msg % (decay_function, ', '.join(DECAY_FUNCTIONS.keys())))
        self._decay_function = DECAY_FUNCTIONS[decay_function]
        self._decay_function_name = decay_function

        self.compact_support = compact_support
        available = [n for n in NEIGHBORHOODS if not (topology == 'hexagonal' and n == 'triangle')]
        if neighborhood_function not in available:
            msg = '%s not supported. Functions available: %s'
            raise ValueError(msg % (neighborhood_function, ', '.join(available)))
        self.neighborhood_func_name = neighborhood_function

        if activation_distance not in DISTANCES:
            msg = '%s not supported. Distances available: %s'
            raise ValueError(msg % (activation_distance, ', '.join(DISTANCES.keys())))
        self._activation_distance_name = activation_distance
        self._activation_distance_kwargs = activation_distance_kwargs

        if not DISTANCES[activation_distance]:
            raise NotImplementedError("activation_distance '%s' is not in the HIP engine yet "
                                      "(SURVEY 8(f) rank 3)" % activation_distance)
        if precision in ('bf16x3', 'f16x3'):
            raise ValueError("precision '%s' is retired: 'exact' returns float32's own BMUs, faster" % precision)
        if precision not in ('f32', 'exact', 'bf16', 'f16'):
            raise ValueError("precision must be 'exact', 'f32', 'bf16' or 'f16'")
        # what som_create would refuse is refused here, at construction, as the reference raises at
        # construction (the engine itself is created lazily, on the first train() / winner())
        if neighborhood_function == 'mexican_hat' and compact_support and topology == 'rectangular' and x != y:
            # the reference masks px twice and py never (neighborhoods.py:69-71); its second mask is an (n, y) array
            # against px's (n, x), so NumPy refuses to broadcast it at the first _update.  Reproduced on square maps.
            raise ValueError('mexican_hat with compact_support needs a square map on the rectangular topology: '
                             'operands could not be broadcast together with shapes (n,%d) (n,%d)' % (x, y))
        if precision not in ('f32', 'exact') and activation_distance not in ('euclidean', 'cosine'):
            raise ValueError("precision '%s' implements the GEMM-form distances 'euclidean' and 'cosine'; "
                             "'%s' needs precision='f32'" % (precision, activation_distance))
        if activation_distance.startswith('norm_p'):
            p = activation_distance_kwargs.get('p', 2)
            if isinstance(p, (int, np.integer)) or (isinstance(p, (float, np.floating)) and float(p).is_integer()):
                if not 1 <= int(p) <= 16:
                    raise NotImplementedError("norm_p: an integer exponent p must be in 1..16, got %r" % (p,))
            elif isinstance(p, (float, np.floating)):
                # a real exponent (distances.py:61-75 takes any p): the generic |x - w|^p form
                if not (np.isfinite(p) and 0.0 < float(p) <= 64.0):
                    raise NotImplementedError("norm_p: a real exponent p must be in (0, 64], got %r" % (p,))
            else:
                raise NotImplementedError("norm_p: the exponent p must be a number, got %r" % (p,))

        # n_parallel bounded the (n,K) temporaries of the reference (xpysom.py:242-251); nothing of
        # that size exists here, it only sizes host->device staging of winner()/quantization_error().
        if n_parallel == 0:
            n_parallel = DEFAULT_BATCH_ROWS
        self._n_parallel = n_parallel

        self._precision = precision
        self._device = device
        self._sharded_input = sharded_input
        if shard not in _dist.SHARDS:
            raise ValueError("shard must be one of %s" % ", ".join(_dist.SHARDS))
        self._shard = shard
        self._engine_obj = None

    # ------------------------------------------------------------------ engine plumbing
    def _engine(self):
        if self._engine_obj is None:
            x, y, _ = self._weights.shape
            kw = dict(distance=self._activation_distance_name, neighborhood=self.neighborhood_func_name,
                      std_coeff=self._std_coeff, compact_support=self.compact_support,
                      precision=self._precision, topology=self.topology)
            if self._activation_distance_name.startswith('norm_p'):
                p = self._activation_distance_kwargs.get('p', 2)
                if isinstance(p, (float, np.floating)) and not float(p).is_integer():
                    kw['norm_p'], kw['norm_p_real'] = 2, float(p)
                else:
                    kw['norm_p'] = int(p)
            from . import engine
            dev = self._device
            if dev is None:
                import os
                dev = int(os.environ.get('LOCAL_RANK', '0'))
            self._engine_obj = engine.HipEngine(x, y, self._input_len, device=dev, **kw)
        return self._engine_obj

    def _upload_weights(self):
        eng = self._engine()
        eng.set_weights(np.asarray(self._weights, dtype=np.float32))
        return eng

    def get_weights(self):
        """Returns the weights of the neural network."""
        return self._weights

    def get_euclidean_coordinates(self):
        """Meshgrids (xx, yy) of the units' positions on the euclidean plane of the chosen topology:
        unit (i, j) sits at (xx[i, j], yy[i, j]) (xpysom.py:291-305)."""
        xx, yy = np.meshgrid(self._neigx, self._neigy)
        xx, yy = xx.astype(float), yy.astype(float)
        if self.topology == 'hexagonal':
            xx[::-2] -= 0.5
        return xx.T, yy.T

    def convert_map_to_euclidean(self, xy):
        """Map coordinates -> euclidean coordinates of the chosen topology (xpysom.py:308-320)."""
        xx, yy = self.get_euclidean_coordinates()
        return xx[xy], yy[xy]

    def activate(self, x):
        """Activation map of x: its distance to every unit under the configured distance, shape (n, K)
        (xpysom.py:323-354).  An analysis call: training never materialises this matrix."""
        x = _host_rows(x, self._engine)
        if x.ndim == 0:
            x = x.reshape(1, 1)
        elif x.ndim == 1:
            x = x[None, :]
        if self._activation_distance_name not in ('euclidean', 'euclidean_no_opt', 'cosine'):
            raise NotImplementedError("activate() returns the (n, K) matrix of the GEMM-form distances only; '%s' is "
                                      "a fused distance+argmin kernel here (use winner())" % self._activation_distance_name)
        return self._upload_weights().distance_matrix(x)

    def distance_from_weights(self, data, weights_gpu=None):
        """d[i, j] = euclidean distance between data[i] and the j-th unit (xpysom.py:647-671)."""
        data = _host_rows(data, self._engine)
        return self._upload_weights().distance_matrix(data, quantization=True)

    def distance_map(self):
        """U-matrix: each unit's summed distance to its map neighbours, scaled to a maximum of 1
        (xpysom.py:788-817: 8 neighbours on the rectangular grid; 6 on the hexagonal one, whose offsets
        depend on the parity of the unit's column index j).  One shifted difference per neighbour offset over
        the whole codebook -- host-side, the codebook is tiny next to the data."""
        w = np.asarray(self._weights, dtype=np.float64)
        X, Y = w.shape[:2]
        if self.topology == 'hexagonal':
            offsets = {1: ((0, 1), (1, 0), (0, -1), (-1, -1), (-1, 0), (-1, 1)),      # even j
                       0: ((1, 1), (1, 0), (1, -1), (0, -1), (-1, 0), (0, 1))}        # odd j
        else:
            ring = tuple((di, dj) for di in (-1, 0, 1) for dj in (-1, 0, 1) if (di, dj) != (0, 0))
            offsets = {0: ring, 1: ring}
        um = np.zeros((X, Y))
        even = (np.arange(Y) % 2 == 0)[None, :]
        for parity, offs in offsets.items():
            for di, dj in offs:
                # units (x, y) with an in-range neighbour (x + di, y + dj)
                xs = slice(max(0, -di), X - max(0, di))
                ys = slice(max(0, -dj), Y - max(0, dj))
                xn = slice(max(0, di), X - max(0, -di))
                yn = slice(max(0, dj), Y - max(0, -dj))
                d = np.linalg.norm(w[xs, ys] - w[xn, yn], axis=-1)
                um[xs, ys] += d * (even[:, ys] == bool(parity))
        return um / um.max()

    def _check_input_len(self, data):
        """Checks that the data in input is of the correct shape (xpysom.py:360-366)."""
        data_len = len(data[0])
        if self._input_len != data_len:
            msg = 'Received %d features, expected %d.' % (data_len, self._input_len)
            raise ValueError(msg)

    # ------------------------------------------------------------------ training
    def train(self, data, num_epochs, iter_beg=0, iter_end=None, verbose=False):
        """Trains the SOM (batch algorithm); same contract as xpysom.py:458-594.

        ``iter_beg``/``iter_end`` resume the schedule mid-run (decays are pure functions of
        (iteration, num_epochs)).  Under an initialised torch.distributed group the rows are
        sharded contiguously over the ranks and the per-epoch numerator/denominator are
        all-reduced before the merge.  Returns ``self``; ``_weights`` is float32 afterwards."""
        if iter_end is None:
            iter_end = num_epochs

        rank, world = _dist.dist_info()
        if _device_rows(data) is not None:               # rows already in HBM: no host round trip
            if world > 1 and not self._sharded_input:
                data = _dist.shard_rows(data, rank, world, getattr(self, '_shard', 'contiguous'))
            ptr, n, d, dev_index, owner, stream = _device_rows(data)
            eng = self._upload_weights()
            if dev_index is not None and dev_index != eng.device:
                raise ValueError('device data lives on cuda:%d, the engine on cuda:%d' % (dev_index, eng.device))
            if d != self._input_len:
                raise ValueError('Received %d features, expected %d.' % (d, self._input_len))
            if stream != "done":                         # rows of a foreign producer: wait for its stream (or the device)
                eng.sync_producer(stream)
            eng.set_data_device(ptr, n, keepalive=owner)
        else:
            data = np.asarray(data, dtype=np.float32)
            if data.ndim != 2:
                raise ValueError('data must be 2-dimensional (n_samples, input_len)')
            if world > 1 and not self._sharded_input:
                data = _dist.shard_rows(data, rank, world, getattr(self, '_shard', 'contiguous'))
            eng = self._upload_weights()
            eng.set_data(data)

        try:
            for iteration in range(iter_beg, iter_end):
                eta = self._decay_function(self._learning_rate, self._learning_rateN, iteration, num_epochs)
                # sigma and learning rate decrease with the same rule
                sig = self._decay_function(self._sigma, self._sigmaN, iteration, num_epochs)
                # NumPy >= 2: a numpy scalar sigma makes the reference's neighbourhood float64
                neigh_f64 = isinstance(sig, np.generic)
                self._check_sigma(sig)
                _dist.epoch(eng, sig, eta, neigh_f64)
                if verbose:
                    print('\r [ %d / %d ] %3.0f%%' % (iteration + 1, num_epochs, 100 * (iteration + 1) / num_epochs),
                          end='')
        finally:
            # also on the way out of an exception (mexican_hat's ZeroDivisionError at sigma 0): the object holds
            # the codebook of the epochs that completed, as the reference's does
            self._weights = eng.get_weights().reshape(self._weights.shape)

        if verbose:
            print('\n quantization error:', self.quantization_error(data))
        return self

    def _check_sigma(self, sig):
        """mexican_hat evaluates ``1 - 2/d*p`` with ``d = 2*std_coeff**2*sigma**2`` (neighborhoods.py:72, :94): a
        Python-float sigma of exactly 0 -- a linear schedule ending at sigmaN=0 -- raises there, in the reference, and
        so here (a numpy.float64 sigma divides to inf with a warning; the gaussian's 0/0 gives NaN in both)."""
        if self.neighborhood_func_name == 'mexican_hat' and not isinstance(sig, np.generic) \
                and 2 * self._std_coeff ** 2 * sig ** 2 == 0:    # d, associated as neighborhoods.py:59 writes it
            raise ZeroDivisionError('float division by zero')

    def train_streaming(self, chunks, num_epochs, iter_beg=0, iter_end=None):
        """``train`` for data that does not stay resident in HBM (or in host memory as one array).

        ``chunks`` is a callable returning a fresh iterable of ``(n_i, input_len)`` row blocks for each
        epoch (e.g. slices of a ``numpy.memmap``) -- the role Dask blocks play in the reference
        (xpysom.py:545-556).  Every block goes host -> HBM, through the BMU kernel and into the same
        segment sums; the epoch ends with the usual transform, all-reduce and merge.  Under a process
        group each rank streams its own blocks."""
        if iter_end is None:
            iter_end = num_epochs
        eng = self._upload_weights()
        try:
            for iteration in range(iter_beg, iter_end):
                eta = self._decay_function(self._learning_rate, self._learning_rateN, iteration, num_epochs)
                sig = self._decay_function(self._sigma, self._sigmaN, iteration, num_epochs)
                self._check_sigma(sig)
                _dist.epoch(eng, sig, eta, isinstance(sig, np.generic), chunks=chunks())
        finally:
            self._weights = eng.get_weights().reshape(self._weights.shape)
        return self

    def train_batch(self, data, num_iteration, verbose=False):
        """Compatibility with MiniSom, alias for train"""
        return self.train(data, num_iteration, verbose=verbose)

    def train_random(self, data, num_iteration, verbose=False):
        """Compatibility with MiniSom, alias for train"""
        print("WARNING: due to batch SOM algorithm, random order is not supported. Falling back to train_batch.")
        return self.train(data, num_iteration, verbose=verbose)

    # ------------------------------------------------------------------ inference
    def _device_query(self, x):
        """(engine, pointer, n_rows) for 2-D float32 rows that already live in HBM, else None: such rows are searched where
        they are (som_bmu_device), whole -- no host round trip, no n_parallel chunks (xpysom.py:379-396 on a CuPy array)."""
        shape = getattr(x, 'shape', None)
        if shape is None or len(shape) != 2:
            return None
        dev = _device_rows(x)
        if dev is None:
            return None
        ptr, n, d, dev_index, owner, stream = dev
        if d != self._input_len:
            raise ValueError('Received %d features, expected %d.' % (d, self._input_len))
        eng = self._upload_weights()
        if dev_index is not None and dev_index != eng.device:
            raise ValueError('device data lives on cuda:%d, the engine on cuda:%d' % (dev_index, eng.device))
        if stream != "done":
            eng.sync_producer(stream)
        return eng, ptr, n, owner

    def _ids_of(self, x, quantization=False):
        """Raveled BMU ids of host or device rows."""
        q = self._device_query(x)
        if q is not None:
            eng, ptr, n, _owner = q
            return eng.bmu_device(ptr, n, quantization=quantization)
        return self._winner_ids(_host_rows(x, self._engine), quantization=quantization)

    def _winner_ids(self, x2d, quantization=False):
        eng = self._upload_weights()
        out = [eng.bmu(x2d[s:s + self._n_parallel], quantization=quantization)
               for s in range(0, len(x2d), self._n_parallel)]
        return np.concatenate(out) if out else np.zeros(0, dtype=np.int32)

    def winner(self, x):
        """Coordinates of the winning neuron(s): ``(i, j)`` for one sample, a list of
        ``(i, j)`` tuples (numpy.int64) for a matrix -- xpysom.py:370-408."""
        # the reference does not coerce x (xpysom.py:379-396): float64 rows against trained (float32) weights are scored in
        # float64 by NumPy.  Served for the 'euclidean' distance in the float32-exact modes; elsewhere x is taken as float32
        x64 = None
        if (isinstance(x, np.ndarray) and x.dtype == np.float64 and np.asarray(self._weights).dtype == np.float32
                and self._activation_distance_name == 'euclidean' and self._precision in ('f32', 'exact')):
            x64 = x
        if self._device_query(x) is not None:               # rows in HBM: searched there
            ids = self._ids_of(x).astype(np.int64)
            wi, wj = np.divmod(ids, self._weights.shape[1])
            return list(map(tuple, np.vstack([wi, wj]).T))
        x = _host_rows(x, self._engine)
        one = x.ndim == 1
        if one:
            x = x[None, :]
        if x64 is not None:
            x64 = x64[None, :] if one else x64
            eng = self._upload_weights()
            ids = np.concatenate([eng.bmu_f64(x64[s:s + self._n_parallel])
                                  for s in range(0, len(x64), self._n_parallel)]).astype(np.int64) if len(x64) else np.zeros(0, np.int64)
        else:
            ids = self._winner_ids(x).astype(np.int64)
        wi, wj = np.divmod(ids, self._weights.shape[1])
        if one:
            return (wi[0].item(), wj[0].item())
        return list(map(tuple, np.vstack([wi, wj]).T))

    def quantization(self, data):
        """Assigns a code book (weights vector of the winning neuron) to each sample in data."""
        if self._device_query(data) is None:
            data = _host_rows(data, self._engine)
            self._check_input_len(data)
        ids = self._ids_of(data, quantization=True)
        w = np.asarray(self._weights)
        return w.reshape(-1, w.shape[2])[ids]

    def quantization_error(self, data):
        """Average distance between each input sample and its best matching unit
        (always Euclidean, xpysom.py:673-707).  Returns a Python float."""
        q = self._device_query(data)
        if q is not None:
            eng, ptr, n, _owner = q
            return eng.quantization_error_device(ptr, n) if n else float('nan')
        data = _host_rows(data, self._engine)
        self._check_input_len(data)
        eng = self._upload_weights()
        total, n = 0.0, 0
        for s in range(0, len(data), self._n_parallel):
            chunk = data[s:s + self._n_parallel]
            total += eng.quantization_error(chunk) * len(chunk)
            n += len(chunk)
        return total / n if n else float('nan')

    def topographic_error(self, data):
        """Share of samples whose best and second-best matching units are not adjacent on the map
        (xpysom.py:709-746, rectangular branch: |di| > 1 or |dj| > 1).  The reference sorts the whole
        (n, K) distance matrix; here a fused top-2 variant of the BMU kernel returns the pair."""
        self._check_input_len(data)
        if np.prod(self._weights.shape) == 1:
            warn('The topographic error is not defined for a 1-by-1 map.')
            return np.nan
        data = _host_rows(data, self._engine)
        eng = self._upload_weights()
        Y = self._weights.shape[1]
        bad, n = 0, 0
        for s in range(0, len(data), self._n_parallel):
            b1, b2 = eng.bmu_top2(data[s:s + self._n_parallel])
            i1, j1, i2, j2 = b1 // Y, b1 % Y, b2 // Y, b2 % Y
            if self.topology == 'hexagonal':
                # not adjacent = farther apart than 1.5 in the hexagonal coordinates (xpysom.py:739-746).  The
                # reference reads them as _xx[i, j], _yy[i, j] from its UNtransposed (Y, X) meshgrids, i.e.
                # x = j - s(i)/2, y = i with s marking every second row from the last (xpysom.py:201-206) --
                # reproduced on square maps (pinned by tests/golden/g13); on a non-square map that indexing
                # runs out of bounds in the reference, and the unit's own coordinates x = i - s(j)/2, y = j
                # are used instead (deviations).
                X = self._weights.shape[0]
                if X == Y:
                    x1, y1 = j1 - 0.5 * ((Y - 1 - i1) % 2 == 0), i1
                    x2, y2 = j2 - 0.5 * ((Y - 1 - i2) % 2 == 0), i2
                else:
                    x1, y1 = i1 - 0.5 * ((Y - 1 - j1) % 2 == 0), j1
                    x2, y2 = i2 - 0.5 * ((Y - 1 - j2) % 2 == 0), j2
                bad += int((np.hypot(x1 - x2, (y1 - y2).astype(float)) > 1.5).sum())
            else:
                bad += int(((np.abs(i1 - i2) > 1) | (np.abs(j1 - j2) > 1)).sum())
            n += len(b1)
        return bad / n if n else float('nan')

    def predict(self, data):
        """Raveled BMU index of every sample (xpysom.py:608-617), batched."""
        return self._ids_of(data).astype(np.int64)

    def _rows_by_unit(self, data):
        """BMU ids of `data` grouped: yields ((i, j), row indices in data order), units in order of first win --
        one batched BMU call and one stable sort instead of a winner() call per sample."""
        ids = self._ids_of(data).astype(np.int64)
        if not len(ids):
            return
        order = np.argsort(ids, kind='stable')
        cuts = np.flatnonzero(np.diff(ids[order])) + 1
        groups = np.split(order, cuts)
        Y = self._weights.shape[1]
        for rows in sorted(groups, key=lambda r: r[0]):
            i, j = np.divmod(ids[rows[0]], Y)
            yield (i, j), rows

    def activation_response(self, data):
        """Matrix where element i,j is the number of times neuron i,j won (xpysom.py:819-829)."""
        if self._device_query(data) is None:
            data = _host_rows(data, self._engine)
            self._check_input_len(data)
        X, Y = self._weights.shape[:2]
        return np.bincount(self._ids_of(data), minlength=X * Y).astype(float).reshape(X, Y)

    def win_map(self, data):
        """Dictionary wm where wm[(i,j)] lists the patterns mapped to i,j (xpysom.py:831-840)."""
        self._check_input_len(data)
        winmap = defaultdict(list)
        host = data if hasattr(data, '__getitem__') and _device_rows(data) is None else _host_rows(data, self._engine)
        for unit, rows in self._rows_by_unit(host):
            winmap[unit] = [host[n] for n in rows]
        return winmap

    def labels_map(self, data, labels):
        """Dictionary wm where wm[(i,j)] counts the labels mapped to i,j (xpysom.py:842-865)."""
        self._check_input_len(data)
        if not len(data) == len(labels):
            raise ValueError('data and labels must have the same length.')
        winmap = defaultdict(list)
        for unit, rows in self._rows_by_unit(data):
            winmap[unit] = Counter(labels[n] for n in rows)
        return winmap

    # ------------------------------------------------------------------ host-side initialisers
    def random_weights_init(self, data):
        """Initializes the weights picking random samples from data (xpysom.py:749-759)."""
        self._check_input_len(data)
        x, y, _ = self._weights.shape
        for i in range(x):
            for j in range(y):
                self._weights[i, j] = data[self._random_generator.randint(len(data))]

    def pca_weights_init(self, data):
        """Initializes the weights to span the first two principal components (xpysom.py:762-785)."""
        if self._input_len == 1:
            raise ValueError('The data needs at least 2 features for pca initialization')
        self._check_input_len(data)
        if len(self._neigx) == 1 or len(self._neigy) == 1:
            warn('PCA initialization inappropriate:One of the dimensions of the map is 1.')
        pc_length, pc = np.linalg.eig(np.cov(np.transpose(data)))
        order = np.argsort(-pc_length)
        for i, c1 in enumerate(np.linspace(-1, 1, len(self._neigx))):
            for j, c2 in enumerate(np.linspace(-1, 1, len(self._neigy))):
                self._weights[i, j] = c1 * pc[order[0]] + c2 * pc[order[1]]

    # ------------------------------------------------------------------ pickling (xpysom.py:868-892)
    def __getstate__(self):
        state = self.__dict__.copy()
        state['_engine_obj'] = None          # device handle: rebuilt lazily from (config, weights)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
