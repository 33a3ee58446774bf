"""ctypes binding of the C ABI in include/vkmr_hip.h -- the stub a Python host would use.

The library must already be built (vk_merkle_roots_amd.build.build_hip); loading
fails loudly when it is missing -- there is no CPU fallback in the product path.
"""
import ctypes as C
import os

from .build import HIP_LIB, HOST_LIB

OK, NOT_READY = 0, 1
ERR_INVALID, ERR_NO_DEVICE, ERR_OOM, ERR_HIP, ERR_COMM = -1, -2, -3, -4, -5
COMM_ID_BYTES = 128


class Metadata(C.Structure):          # vkmr_metadata == VkSha256Metadata
    _fields_ = [("start", C.c_uint32), ("size", C.c_uint32)]


class Digest(C.Structure):            # vkmr_digest == VkSha256Result
    _fields_ = [("data", C.c_uint32 * 8)]


# name -> (restype, argtypes); every symbol include/vkmr_hip.h declares
SIGNATURES = {
    "vkmr_hip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "vkmr_hip_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "vkmr_hip_device_mem_info": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "vkmr_hip_device_geometry": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vkmr_hip_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "vkmr_hip_host_free": (C.c_int, [C.c_void_p]),
    "vkmr_hip_device_alloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "vkmr_hip_device_free": (C.c_int, [C.c_int, C.c_void_p]),
    "vkmr_hip_memset_async": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    "vkmr_hip_memcpy_h2d_async": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vkmr_hip_memcpy_d2h_async": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vkmr_hip_stream_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "vkmr_hip_metadata_from_sizes_async": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "vkmr_hip_sizes_scratch_bytes": (C.c_size_t, [C.c_uint32]),
    "vkmr_hip_warm_up": (C.c_int, [C.c_int, C.c_void_p, C.c_uint, C.c_size_t]),
    "vkmr_hip_stream_destroy": (C.c_int, [C.c_int, C.c_void_p]),
    "vkmr_hip_stream_sync": (C.c_int, [C.c_int, C.c_void_p]),
    "vkmr_hip_event_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "vkmr_hip_event_destroy": (C.c_int, [C.c_int, C.c_void_p]),
    "vkmr_hip_event_record": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "vkmr_hip_event_query": (C.c_int, [C.c_int, C.c_void_p]),
    "vkmr_hip_event_wait": (C.c_int, [C.c_int, C.c_void_p]),
    "vkmr_hip_stream_wait_event": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "vkmr_hip_event_elapsed_ms": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "vkmr_hip_map_async": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "vkmr_hip_reduce_async": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]),
    "vkmr_hip_reduce_scratch_bytes": (C.c_size_t, [C.c_uint64]),
    "vkmr_hip_reduce_slices_async": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32,
                                               C.c_void_p, C.c_void_p]),
    "vkmr_hip_reduce_slices_scratch_bytes": (C.c_size_t, [C.c_uint64, C.c_uint32]),
    "vkmr_hip_proof_async": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "vkmr_hip_reduce_proofs_async": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                               C.c_void_p]),
    "vkmr_hip_reduce_levels_async": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]),
    "vkmr_hip_reduce_levels_scratch_bytes": (C.c_size_t, [C.c_uint64]),
    "vkmr_hip_combine_async": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "vkmr_hip_comm_init_all": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "vkmr_hip_comm_create_id": (C.c_int, [C.c_void_p]),
    "vkmr_hip_comm_init_rank": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "vkmr_hip_comm_destroy": (C.c_int, [C.c_void_p]),
    "vkmr_hip_comm_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vkmr_hip_gather_roots_async": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_void_p)]),
    "vkmr_hip_roots_in_slice_order_async": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "vkmr_hip_digest_hex": (None, [C.c_void_p, C.c_char_p]),
    "vkmr_hip_last_error": (C.c_char_p, []),
    "vkmr_hip_kernel_info": (C.c_char_p, []),
    "vkmr_hip_comm_info": (C.c_char_p, []),
}

HOST_SIGNATURES = {
    "vkmr_host_cpu_leaves": (None, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "vkmr_host_cpu_reduce": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    "vkmr_host_cpu_combine": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "vkmr_host_cpu_fold_proof": (None, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "vkmr_host_rndm_pack": (C.c_int64, [C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p,
                                        C.POINTER(C.c_uint64)]),
    "vkmr_host_rndm_open": (C.c_void_p, [C.c_uint32]),
    "vkmr_host_rndm_close": (None, [C.c_void_p]),
    "vkmr_host_rndm_next": (C.c_int64, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]),
    "vkmr_host_rndm_rand": (None, [C.c_uint32, C.c_void_p, C.c_uint64]),
    "vkmr_host_pack_lines": (C.c_int64, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vkmr_host_pack_lines_portable": (C.c_int64, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                                  C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vkmr_host_pack_indexed": (C.c_int64, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]),
    "vkmr_host_copy_and_count": (None, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vkmr_host_count_lines": (None, [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]),
    "vkmr_host_pack_prefix": (None, [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]),
}

# include/vkmr_hip_experiments.h: exported by the experiments build only (build/ab/libexp.so); bound when present
EXPERIMENT_SIGNATURES = {
    "vkmr_hip_split_text_async": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32,
                                            C.c_void_p]),
    "vkmr_hip_split_scratch_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32]),
}

_lib = None
_host = None


class VkmrError(RuntimeError):
    def __init__(self, status, what):
        super().__init__(f"{what} failed with status {status}: {what_error()}")
        self.status = status


def _bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)          # AttributeError when a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


def experiments_lib(path=None):
    """The experiments build (include/vkmr_hip_experiments.h), loaded BESIDE whatever lib() is: same process, same HIP
    runtime, so device pointers and streams of the one are good in the other.  Tests of the non-shipped entry points only."""
    from .build import EXP_LIB
    path = path or EXP_LIB
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: vk_merkle_roots_amd.build.build_experiments()")
    return _bind(C.CDLL(path), EXPERIMENT_SIGNATURES)   # RTLD_LOCAL: its other symbols must not shadow the product library's


def lib():
    """The HIP extension.  Raises when it is not built -- never falls back."""
    global _lib
    if _lib is None:
        path = os.environ.get("VKMR_HIP_LIB", HIP_LIB)      # alternative build of the SAME extension, for A/B timing
        if path != HIP_LIB and os.path.exists(path):
            _lib = _bind(C.CDLL(path, mode=C.RTLD_GLOBAL), SIGNATURES)
            return _lib
        if not os.path.exists(HIP_LIB):
            raise RuntimeError(f"{HIP_LIB} is missing: run `python -m vk_merkle_roots_amd.build` "
                               "(the HIP extension is required; there is no CPU fallback)")
        _lib = _bind(C.CDLL(HIP_LIB, mode=C.RTLD_GLOBAL), SIGNATURES)
    return _lib


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError(f"{HOST_LIB} is missing: run `python -m vk_merkle_roots_amd.build`")
        _host = _bind(C.CDLL(HOST_LIB), HOST_SIGNATURES)
    return _host


def what_error():
    if _lib is None:
        return ""
    return (_lib.vkmr_hip_last_error() or b"").decode("utf-8", "replace")


def check(status, what):
    if status < 0:
        raise VkmrError(status, what)
    return status
