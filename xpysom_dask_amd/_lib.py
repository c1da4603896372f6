"""ctypes binding of libsomhip.so (include/somhip.h).  There is no fallback:
if the HIP library is missing or does not load, importing the engine fails."""
import ctypes as C
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SOM_LIB_PATH") or os.path.join(HERE, "libsomhip.so")   # override: kernel A/B builds

SOM_DIST = {"euclidean": 0, "euclidean_no_opt": 1, "cosine": 2, "manhattan": 3, "manhattan_no_opt": 3,
            "norm_p": 4, "norm_p_no_opt": 5}
SOM_NEIGH = {"gaussian": 0, "mexican_hat": 1, "bubble": 2, "triangle": 3}
SOM_PREC = {"f32": 0, "bf16": 1, "f16": 3, "exact": 5}      # (ids 2 and 4: the retired split-operand modes)
SOM_TOPO = {"rectangular": 0, "hexagonal": 1}
SOM_BMU_ACTIVATION, SOM_BMU_QUANTIZATION = 0, 1
SOM_KERNELS = {"bmu": 0, "segsum": 1, "kron": 2, "merge": 3, "prep": 4, "screen": 5}


class SomConfig(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("input_len", C.c_int32),
                ("distance", C.c_int32), ("neighborhood", C.c_int32),
                ("compact_support", C.c_int32), ("precision", C.c_int32), ("device", C.c_int32),
                ("std_coeff", C.c_double), ("stream", C.c_void_p),
                ("topology", C.c_int32), ("norm_p", C.c_int32), ("norm_p_real", C.c_double)]


_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int32)
_H = C.c_void_p

# name -> (restype, argtypes); exactly the symbols include/somhip.h declares
SIGNATURES = {
    "som_version": (C.c_char_p, []),
    "som_device_count": (C.c_int, []),
    "som_last_error": (C.c_char_p, [_H]),
    "som_create": (C.c_int, [C.POINTER(SomConfig), C.POINTER(_H)]),
    "som_destroy": (None, [_H]),
    "som_set_weights": (C.c_int, [_H, _F]),
    "som_get_weights": (C.c_int, [_H, _F]),
    "som_set_data": (C.c_int, [_H, _F, C.c_int64]),
    "som_set_data_device": (C.c_int, [_H, C.c_void_p, C.c_int64]),
    "som_sync_producer": (C.c_int, [_H, C.c_uint64, C.c_int32]),
    "som_copy_to_host": (C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_void_p]),
    "som_epoch_accumulate": (C.c_int, [_H, C.c_double, C.c_double, C.c_int]),
    "som_epoch_accumulate_faithful": (C.c_int, [_H, C.c_double, C.c_double, C.c_int]),
    "som_epoch_merge": (C.c_int, [_H]),
    "som_epoch": (C.c_int, [_H, C.c_double, C.c_double, C.c_int]),
    "som_epoch_accumulate_begin": (C.c_int, [_H, C.c_double, C.c_double, C.c_int]),
    "som_epoch_block_count": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "som_epoch_accumulate_block": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "som_comm_load": (C.c_int, [C.c_char_p]),
    "som_comm_unique_id": (C.c_int, [C.c_void_p]),
    "som_comm_init": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p]),
    "som_epoch_allreduce": (C.c_int, [_H]),
    "som_comm_destroy": (C.c_int, [_H]),
    "som_pinned_alloc": (C.c_int, [C.c_uint64, C.POINTER(C.c_void_p)]),
    "som_pinned_free": (C.c_int, [C.c_void_p]),
    "som_stream_begin": (C.c_int, [_H]),
    "som_stream_rows": (C.c_int, [_H, _F, C.c_int64]),
    "som_stream_end": (C.c_int, [_H, C.c_double, C.c_double, C.c_int]),
    "som_accum_device_ptr": (C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "som_get_stream": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "som_epoch_fetch": (C.c_int, [_H, _F, _F, _I]),
    "som_epoch_accumulate_forced": (C.c_int, [_H, _I, C.c_double, C.c_double, C.c_int]),
    "som_bmu": (C.c_int, [_H, _F, C.c_int64, C.c_int32, _I]),
    "som_bmu_device": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_int32, _I]),
    "som_bmu_top2": (C.c_int, [_H, _F, C.c_int64, _I, _I]),
    "som_bmu_f64": (C.c_int, [_H, C.POINTER(C.c_double), C.c_int64, _I]),
    "som_distance_matrix": (C.c_int, [_H, _F, C.c_int64, C.c_int32, _F]),
    "som_quantization_error": (C.c_int, [_H, _F, C.c_int64, C.POINTER(C.c_double)]),
    "som_quantization_error_device": (C.c_int, [_H, C.c_void_p, C.c_int64, C.POINTER(C.c_double)]),
    "som_set_verify": (C.c_int, [_H, C.c_int32]),
    "som_verify_stats": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "som_debug_corrupt_operands": (C.c_int, [_H, C.c_int32]),
    "som_debug_mfma16": (C.c_int, [_H, C.c_void_p, C.c_void_p, _F, _F, C.c_int32]),
    "som_debug_stamps": (C.c_int, [_H, C.c_int64, C.c_void_p]),
    "som_policy_eval": (C.c_int, [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "som_patch_order": (C.c_int, [C.c_int32, C.c_int32, _I]),
    "som_exact_stats": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "som_exact_skip_stats": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "som_exact_resident_stats": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "som_exact_scout_stats": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "som_exact_refine_stats": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "som_exact_last_counts": (C.c_int, [_H, _I, C.c_int64]),
    "som_sync": (C.c_int, [_H]),
    "som_profile_enable": (C.c_int, [_H, C.c_int32]),
    "som_profile_get": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "som_profile_reset": (C.c_int, [_H]),
}

_lib = None


def torch_lib_file(name):
    """Path of a library bundled with torch (or None): the copy this process must share with torch."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", name)
    return cand if os.path.exists(cand) else None


def _preload_torch_hip_runtime():
    """torch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).
    If torch is (or will be) imported in this process -- it is, for multi-GPU runs --
    both libraries must resolve to ONE HIP runtime, or streams/pointers handed across
    would belong to different runtimes.  Loading torch's copy first makes the loader
    satisfy libsomhip's DT_NEEDED libamdhip64.so.7 with it."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m xpysom_dask_amd.build` "
            "(needs hipcc; there is no CPU fallback)")
    if not os.environ.get("SOM_LIB_PATH"):             # (an explicit A/B build is the caller's business)
        from . import build as _build
        have, want = _build.built_hash(LIB_PATH), _build.source_hash()
        if have != want:
            raise ImportError(
                f"{LIB_PATH} is stale: built from sources {have}, the tree is {want}; "
                "rebuild with `python -m xpysom_dask_amd.build`")
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
