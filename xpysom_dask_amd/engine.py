"""Thin object wrapper over the C handle of libsomhip (include/somhip.h).

`HipEngine` is the only thing the host class talks to.  Its method set is the engine
interface the distributed driver (distributed.py) relies on:
    set_weights / get_weights / set_data / epoch_accumulate / accum_tensor / epoch_merge
"""
import ctypes as C

import numpy as np

from . import _lib


class SomHipError(RuntimeError):
    pass


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class HipEngine:
    def __init__(self, x, y, input_len, *, distance="euclidean", neighborhood="gaussian",
                 std_coeff=0.5, compact_support=False, precision="exact", device=0, stream=None,
                 topology="rectangular", norm_p=0, norm_p_real=0.0):
        self._lib = _lib.load()
        self._h = None
        self.K, self.D = int(x) * int(y), int(input_len)
        self.x, self.y = int(x), int(y)
        self.n_rows = 0
        self._keepalive = None
        cfg = _lib.SomConfig(int(x), int(y), int(input_len), _lib.SOM_DIST[distance],
                             _lib.SOM_NEIGH[neighborhood], int(bool(compact_support)),
                             _lib.SOM_PREC[precision], int(device), float(std_coeff),
                             C.c_void_p(stream) if stream else None, _lib.SOM_TOPO[topology], int(norm_p), float(norm_p_real))
        h = C.c_void_p()
        if self._lib.som_create(C.byref(cfg), C.byref(h)) != 0:
            raise SomHipError(self._lib.som_last_error(None).decode())
        self._h = h
        self.has_comm = False
        self.device = int(device)
        self.precision = precision

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise SomHipError(self._lib.som_last_error(self._h).decode())

    def close(self):
        if self._h is not None:
            self._lib.som_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _fp(a):
        return a.ctypes.data_as(C.POINTER(C.c_float))

    @staticmethod
    def _ip(a):
        return a.ctypes.data_as(C.POINTER(C.c_int32))

    # -- codebook / data --------------------------------------------------------------------
    def set_weights(self, w):
        w = _f32(w).reshape(self.K, self.D)
        self._check(self._lib.som_set_weights(self._h, self._fp(w)))

    def get_weights(self):
        w = np.empty((self.K, self.D), dtype=np.float32)
        self._check(self._lib.som_get_weights(self._h, self._fp(w)))
        return w

    def set_data(self, data):
        data = _f32(data)
        if data.ndim != 2 or data.shape[1] != self.D:
            raise ValueError("data must be (n, %d), got %r" % (self.D, data.shape))
        self._check(self._lib.som_set_data(self._h, self._fp(data), data.shape[0]))
        self.n_rows = data.shape[0]
        self._keepalive = None

    def set_data_device(self, dev_ptr, n_rows, keepalive=None):
        """Rows already resident in HBM (float32 [n][D]); `keepalive` is whatever owns them."""
        self._check(self._lib.som_set_data_device(self._h, C.c_void_p(dev_ptr), int(n_rows)))
        self.n_rows = int(n_rows)
        self._keepalive = keepalive

    def sync_producer(self, stream=None):
        """Wait for the producer of device rows: its `__cuda_array_interface__` stream, or (None) the device."""
        self._check(self._lib.som_sync_producer(self._h, C.c_uint64(int(stream or 0)), int(stream is not None)))

    def copy_to_host(self, dev_ptr, shape):
        """float32 device rows of the caller -> a fresh host array."""
        out = np.empty(shape, dtype=np.float32)
        self._check(self._lib.som_copy_to_host(self._h, C.c_void_p(dev_ptr), C.c_uint64(out.nbytes),
                                               out.ctypes.data_as(C.c_void_p)))
        return out

    # -- one epoch --------------------------------------------------------------------------
    def epoch_accumulate(self, sigma, eta, neigh_f64):
        self._check(self._lib.som_epoch_accumulate(self._h, float(sigma), float(eta), int(bool(neigh_f64))))

    def epoch_accumulate_faithful(self, sigma, eta, neigh_f64):
        """epoch_accumulate through the reference's own formulation (g^T x as one big MFMA GEMM): cross-checks."""
        self._check(self._lib.som_epoch_accumulate_faithful(self._h, float(sigma), float(eta), int(bool(neigh_f64))))

    def epoch_accumulate_forced(self, bmu, sigma, eta, neigh_f64):
        bmu = np.ascontiguousarray(bmu, dtype=np.int32)
        if bmu.shape != (self.n_rows,):
            raise ValueError("bmu must have one id per resident row")
        self._check(self._lib.som_epoch_accumulate_forced(self._h, self._ip(bmu), float(sigma), float(eta),
                                                          int(bool(neigh_f64))))

    def epoch_accumulate_begin(self, sigma, eta, neigh_f64):
        """epoch_accumulate up to stage 1 of the transform; `epoch_accumulate_block` finishes the accumulator
        one map-row block at a time (see include/somhip.h)."""
        self._check(self._lib.som_epoch_accumulate_begin(self._h, float(sigma), float(eta), int(bool(neigh_f64))))

    def epoch_block_count(self):
        n = C.c_int32()
        self._check(self._lib.som_epoch_block_count(self._h, C.byref(n)))
        return n.value

    def epoch_accumulate_block(self, block):
        """Stage 2 of map-row block `block`; returns (offset, n_floats) of the now final slice of the accumulator."""
        off, n = C.c_int64(), C.c_int64()
        self._check(self._lib.som_epoch_accumulate_block(self._h, int(block), C.byref(off), C.byref(n)))
        return off.value, n.value

    # -- the collective inside the library (include/somhip.h, som_comm_*) ---------------------------------
    def comm_unique_id(self):
        """128 opaque bytes from RCCL: rank 0 creates them, the host hands them to every rank."""
        import sys
        if "torch" in sys.modules:                    # share torch's copy of RCCL (and so its HIP runtime)
            path = _lib.torch_lib_file("librccl.so")
            if path:
                self._lib.som_comm_load(path.encode())
        buf = C.create_string_buffer(128)
        if self._lib.som_comm_unique_id(buf) != 0:
            raise SomHipError(self._lib.som_last_error(None).decode())
        return buf.raw

    def comm_init(self, world, rank, unique_id):
        import sys
        if "torch" in sys.modules:
            path = _lib.torch_lib_file("librccl.so")
            if path:
                self._lib.som_comm_load(path.encode())
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self._lib.som_comm_init(self._h, int(world), int(rank), buf))
        self.has_comm = True

    def comm_destroy(self):
        self._check(self._lib.som_comm_destroy(self._h))
        self.has_comm = False

    def epoch_allreduce(self):
        self._check(self._lib.som_epoch_allreduce(self._h))

    def epoch_merge(self):
        self._check(self._lib.som_epoch_merge(self._h))

    def epoch(self, sigma, eta, neigh_f64):
        self._check(self._lib.som_epoch(self._h, float(sigma), float(eta), int(bool(neigh_f64))))

    def pinned_empty(self, shape):
        """A float32 array in pinned host memory: chunks streamed from such arrays are copied
        asynchronously and overlap the previous chunk's kernels (use two and alternate)."""
        import weakref
        n = int(np.prod(shape))
        p = C.c_void_p()
        if self._lib.som_pinned_alloc(C.c_uint64(max(1, n) * 4), C.byref(p)) != 0:
            raise SomHipError("som_pinned_alloc failed")
        buf = (C.c_float * max(1, n)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=np.float32, count=n).reshape(shape)
        weakref.finalize(buf, self._lib.som_pinned_free, p)
        return arr

    def stream_epoch_accumulate(self, chunks, sigma, eta, neigh_f64):
        """epoch_accumulate for rows handed over chunk by chunk (an iterable of (n_i, D) arrays)."""
        self._check(self._lib.som_stream_begin(self._h))
        for chunk in chunks:
            chunk = _f32(chunk)
            if chunk.ndim != 2 or chunk.shape[1] != self.D:
                raise ValueError("chunk must be (n, %d), got %r" % (self.D, chunk.shape))
            self._check(self._lib.som_stream_rows(self._h, self._fp(chunk), chunk.shape[0]))
        self._check(self._lib.som_stream_end(self._h, float(sigma), float(eta), int(bool(neigh_f64))))
        self.sync()                       # pinned chunks are copied asynchronously: their buffers are free now

    def epoch_fetch(self, want_bmu=True):
        """(num (K,D), den (K,), bmu (n,) or None) of the last accumulate."""
        num = np.empty((self.K, self.D), dtype=np.float32)
        den = np.empty((self.K,), dtype=np.float32)
        bmu = np.empty((self.n_rows,), dtype=np.int32) if want_bmu else None
        self._check(self._lib.som_epoch_fetch(self._h, self._fp(num), self._fp(den),
                                              self._ip(bmu) if want_bmu else None))
        return num, den, bmu

    def accum_device_ptr(self):
        p, n = C.c_void_p(), C.c_int64()
        self._check(self._lib.som_accum_device_ptr(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def stream_ptr(self):
        """The HIP stream the engine launches on (an integer handle, for torch.cuda.ExternalStream)."""
        p = C.c_void_p()
        self._check(self._lib.som_get_stream(self._h, C.byref(p)))
        return int(p.value or 0)

    def accum_tensor(self):
        """The fused [num|den] accumulator as a torch CUDA tensor aliasing the engine's
        HBM buffer (zero copy) -- what the RCCL all-reduce runs on in place."""
        import torch
        ptr, n = self.accum_device_ptr()

        class _Alias:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False),
                                        "version": 2, "strides": None}
        return torch.as_tensor(_Alias(), device=torch.device("cuda", self.device))

    # -- inference --------------------------------------------------------------------------
    def bmu(self, x, quantization=False):
        x = _f32(x)
        if x.ndim != 2 or x.shape[1] != self.D:
            raise ValueError("x must be (n, %d), got %r" % (self.D, x.shape))
        ids = np.empty((x.shape[0],), dtype=np.int32)
        mode = _lib.SOM_BMU_QUANTIZATION if quantization else _lib.SOM_BMU_ACTIVATION
        self._check(self._lib.som_bmu(self._h, self._fp(x), x.shape[0], mode, self._ip(ids)))
        return ids

    def bmu_device(self, dev_ptr, n_rows, quantization=False):
        """BMU ids of float32 rows that already live in HBM (`dev_ptr`: [n_rows][D], borrowed for the call)."""
        ids = np.empty((int(n_rows),), dtype=np.int32)
        mode = _lib.SOM_BMU_QUANTIZATION if quantization else _lib.SOM_BMU_ACTIVATION
        self._check(self._lib.som_bmu_device(self._h, C.c_void_p(dev_ptr), int(n_rows), mode, self._ip(ids)))
        return ids

    def quantization_error_device(self, dev_ptr, n_rows):
        out = C.c_double()
        self._check(self._lib.som_quantization_error_device(self._h, C.c_void_p(dev_ptr), int(n_rows), C.byref(out)))
        return out.value

    def bmu_f64(self, x):
        """BMUs of float64 rows under the float64 arithmetic NumPy applies to them (som_bmu_f64: euclidean only)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.ndim != 2 or x.shape[1] != self.D:
            raise ValueError("x must be (n, %d), got %r" % (self.D, x.shape))
        ids = np.empty((x.shape[0],), dtype=np.int32)
        self._check(self._lib.som_bmu_f64(self._h, x.ctypes.data_as(C.POINTER(C.c_double)), x.shape[0], self._ip(ids)))
        return ids

    def bmu_top2(self, x):
        """(best, second-best) raveled ids under the full Euclidean distance."""
        x = _f32(x)
        if x.ndim != 2 or x.shape[1] != self.D:
            raise ValueError("x must be (n, %d), got %r" % (self.D, x.shape))
        a = np.empty((x.shape[0],), dtype=np.int32)
        b = np.empty((x.shape[0],), dtype=np.int32)
        self._check(self._lib.som_bmu_top2(self._h, self._fp(x), x.shape[0], self._ip(a), self._ip(b)))
        return a, b

    def distance_matrix(self, x, quantization=False):
        """The (n, K) distance matrix (analysis only)."""
        x = _f32(x)
        if x.ndim != 2 or x.shape[1] != self.D:
            raise ValueError("x must be (n, %d), got %r" % (self.D, x.shape))
        out = np.empty((x.shape[0], self.K), dtype=np.float32)
        mode = _lib.SOM_BMU_QUANTIZATION if quantization else _lib.SOM_BMU_ACTIVATION
        self._check(self._lib.som_distance_matrix(self._h, self._fp(x), x.shape[0], mode, self._fp(out)))
        return out

    def quantization_error(self, x):
        x = _f32(x)
        out = C.c_double()
        self._check(self._lib.som_quantization_error(self._h, self._fp(x), x.shape[0], C.byref(out)))
        return out.value

    def set_verify(self, n_rows):
        """The canary: re-score `n_rows` strided rows of every BMU launch with the float32 kernel (0 = off)."""
        self._check(self._lib.som_set_verify(self._h, int(n_rows)))

    def verify_stats(self):
        a, b = C.c_int64(), C.c_int64()
        self._check(self._lib.som_verify_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def debug_corrupt_operands(self, which=3):
        """TEST HOOK of the canary: zero the operand images behind the library's back (1: 16-bit, 2: float32)."""
        self._check(self._lib.som_debug_corrupt_operands(self._h, int(which)))

    def debug_mfma16(self, a, b, c, f16=True):
        """Measurement hook: d = a (16x32) . b (32x16) + c (16x16) by ONE 16x16x32 MFMA; a, b float16 (or bfloat16 bit
        patterns as uint16 when f16=False), c float32."""
        a = np.ascontiguousarray(a).view(np.uint16).reshape(16, 32)
        b = np.ascontiguousarray(b).view(np.uint16).reshape(32, 16)
        c = np.ascontiguousarray(c, dtype=np.float32).reshape(16, 16)
        d = np.empty((16, 16), dtype=np.float32)
        self._check(self._lib.som_debug_mfma16(self._h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                               self._fp(c), self._fp(d), int(bool(f16))))
        return d

    def exact_stats(self):
        """precision 'exact': (rows screened, rows sent to the float32 fallback kernel, screen passes) so far."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.som_exact_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def exact_skip_stats(self):
        """precision 'exact': (blocks the screens ran, blocks of full scans) so far -- block skipping's executed share."""
        a, b = C.c_int64(), C.c_int64()
        self._check(self._lib.som_exact_skip_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def exact_resident_stats(self):
        """precision 'exact': (epochs run under a plan, of which (re-)sorted the resident rows by their last BMU's patch)."""
        a, b = C.c_int64(), C.c_int64()
        self._check(self._lib.som_exact_resident_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def exact_scout_stats(self):
        """precision 'exact': (BMU launches that ran the scout, launches over transient row sets that ran under a plan)."""
        a, b = C.c_int64(), C.c_int64()
        self._check(self._lib.som_exact_scout_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def exact_refine_stats(self):
        """precision 'exact': (candidate pairs handed to the refinement pass, pairs it left for the float32 re-score)."""
        a, b = C.c_int64(), C.c_int64()
        self._check(self._lib.som_exact_refine_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def exact_last_counts(self, n):
        """precision 'exact': candidate groups per row in the last screen pass (first n rows)."""
        out = np.empty((int(n),), dtype=np.int32)
        self._check(self._lib.som_exact_last_counts(self._h, self._ip(out), int(n)))
        return out

    # -- timing -----------------------------------------------------------------------------
    def sync(self):
        self._check(self._lib.som_sync(self._h))

    def profile_enable(self, on=True):
        self._check(self._lib.som_profile_enable(self._h, 2 if on == "bmu" else int(bool(on))))

    def profile_reset(self):
        self._check(self._lib.som_profile_reset(self._h))

    def profile_get(self, kernel):
        ms, n = C.c_double(), C.c_int64()
        self._check(self._lib.som_profile_get(self._h, _lib.SOM_KERNELS[kernel], C.byref(ms), C.byref(n)))
        return ms.value, n.value
