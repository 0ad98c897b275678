"""ctypes binding of libsplintr_hip.so (include/splintr_hip.h).

The library is the product: if it is missing or cannot be loaded this module raises -- there is
no CPU fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPL_LIB_PATH: development override used to A/B kernel variants built with different -D flags
LIB_PATH = os.environ.get("SPL_LIB_PATH") or os.path.join(_HERE, "libsplintr_hip.so")

SPL_OK = 0
SPL_WITH_SPECIAL = 1
SPL_INTRA_DOC = 2
SPL_MAX_KERNELS = 16

# every symbol include/splintr_hip.h declares (tests/test_abi.py checks the header against this)
SYMBOLS = [
    "spl_last_error", "spl_device_count", "spl_create", "spl_add_special", "spl_vocab_size", "spl_destroy",
    "spl_reserve", "spl_encode_batch", "spl_result_tokens", "spl_result_offsets", "spl_result_n_tokens",
    "spl_result_n_docs", "spl_result_free", "spl_encode_batch_device", "spl_decode_batch", "spl_free",
    "spl_profile_enable", "spl_profile_reset", "spl_profile_read", "spl_kernel_name", "spl_last_queue_counts",
    "spl_debug_phases", "spl_debug_blocks", "spl_gatherv_pack", "spl_gatherv_unpack", "spl_gatherv_unpack_group", "spl_encode_batch_device_packed",
    "spl_set_devices", "spl_n_devices", "spl_set_option", "spl_host_alloc", "spl_host_free",
    "spl_token_bytes", "spl_is_byte_level",
    "spl_comm_unique_id", "spl_comm_create", "spl_comm_destroy", "spl_comm_rank", "spl_comm_world",
    "spl_allgather_slabs", "spl_allgather_slabs_p2p", "spl_gatherv_unpack_at", "spl_allgatherv_csr", "spl_split_host", "spl_encode_chunks_device",
    "spl_split_device", "spl_device_split_fallbacks", "spl_small_path_calls", "spl_pick_stream", "spl_memo_stats",
]
SPL_PATTERN_CUSTOM = 3
SPL_OPT_BYTE_LEVEL = 1


class SplOpts(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("pattern", ctypes.c_int32), ("device", ctypes.c_int32),
                ("flags", ctypes.c_uint32), ("pattern_text", ctypes.c_char_p), ("pattern_len", ctypes.c_uint64)]

    def __init__(self, pattern=0, device=0, flags=0, pattern_text=None):
        super().__init__(ctypes.sizeof(SplOpts), pattern, device, flags, pattern_text, len(pattern_text) if pattern_text else 0)


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). splintr_amd has no CPU fallback.")
    # ONE HIP runtime per process: torch ships its own copy of libamdhip64 and loads it by path; if
    # this library pulled in /opt/rocm's copy first, torch's would come up second and find the GPU's
    # VM already acquired ("No HIP GPUs are available").  With torch imported first both resolve to
    # the copy that is already loaded (same SONAME).  torch is optional for the encode path itself.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, u8p, u32p, u64p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint32), \
        ctypes.POINTER(ctypes.c_uint64)
    L.spl_last_error.restype = ctypes.c_char_p
    L.spl_device_count.restype = ctypes.c_int
    L.spl_create.restype = vp
    L.spl_create.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_size_t, ctypes.POINTER(SplOpts)]
    L.spl_add_special.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32]
    L.spl_vocab_size.restype = ctypes.c_uint32
    L.spl_vocab_size.argtypes = [vp]
    L.spl_destroy.argtypes = [vp]
    L.spl_destroy.restype = None
    L.spl_reserve.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64]
    L.spl_set_devices.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.c_uint32]
    L.spl_n_devices.restype = ctypes.c_uint32
    L.spl_n_devices.argtypes = [vp]
    L.spl_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_int64]
    L.spl_host_alloc.restype = vp
    L.spl_host_alloc.argtypes = [ctypes.c_size_t]
    L.spl_host_free.restype = None
    L.spl_host_free.argtypes = [vp]
    L.spl_token_bytes.argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint32)]
    L.spl_is_byte_level.argtypes = [vp]
    L.spl_encode_batch.argtypes = [vp, vp, vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(vp)]
    L.spl_result_tokens.restype = u32p
    L.spl_result_tokens.argtypes = [vp]
    L.spl_result_offsets.restype = u64p
    L.spl_result_offsets.argtypes = [vp]
    L.spl_result_n_tokens.restype = ctypes.c_uint64
    L.spl_result_n_tokens.argtypes = [vp]
    L.spl_result_n_docs.restype = ctypes.c_uint64
    L.spl_result_n_docs.argtypes = [vp]
    L.spl_result_free.argtypes = [vp]
    L.spl_result_free.restype = None
    L.spl_encode_batch_device.argtypes = [vp, vp, ctypes.c_uint64, vp, ctypes.c_uint64, ctypes.c_uint32, vp,
                                          ctypes.c_uint64, vp, vp]
    L.spl_decode_batch.argtypes = [vp, vp, vp, ctypes.c_uint64, ctypes.POINTER(u8p), ctypes.POINTER(u64p)]
    L.spl_free.argtypes = [vp]
    L.spl_free.restype = None
    L.spl_profile_enable.argtypes = [vp, ctypes.c_int]
    L.spl_profile_reset.argtypes = [vp]
    L.spl_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
    L.spl_kernel_name.restype = ctypes.c_char_p
    L.spl_kernel_name.argtypes = [ctypes.c_int]
    L.spl_last_queue_counts.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32)]
    L.spl_debug_phases.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
    L.spl_debug_blocks.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    L.spl_gatherv_pack.argtypes = [vp, vp, vp, ctypes.c_uint64, vp, ctypes.c_uint64, ctypes.c_uint64, vp]
    L.spl_gatherv_unpack.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, vp, ctypes.c_uint64, vp,
                                     vp, vp]
    L.spl_encode_batch_device_packed.argtypes = [vp, vp, ctypes.c_uint64, vp, ctypes.c_uint64, ctypes.c_uint32, vp,
                                                 ctypes.c_uint64, vp, vp, ctypes.c_uint64, ctypes.c_uint64, vp]
    L.spl_gatherv_unpack_group.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64,
                                           ctypes.c_uint64, vp, ctypes.c_uint64, vp, ctypes.c_uint64, vp, vp]
    L.spl_split_host.argtypes = [vp, vp, vp, ctypes.c_uint64, vp, vp]
    L.spl_split_device.argtypes = [vp, vp, ctypes.c_uint64, vp, ctypes.c_uint64, vp, vp, vp, vp]
    L.spl_device_split_fallbacks.restype = ctypes.c_uint64
    L.spl_device_split_fallbacks.argtypes = [vp]
    L.spl_small_path_calls.restype = ctypes.c_uint64
    L.spl_small_path_calls.argtypes = [vp]
    L.spl_pick_stream.restype = ctypes.c_int
    L.spl_pick_stream.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_double)]
    L.spl_encode_chunks_device.argtypes = [vp, vp, ctypes.c_uint64, vp, ctypes.c_uint64, vp, vp, vp, ctypes.c_uint64, vp, vp]
    L.spl_comm_unique_id.argtypes = [ctypes.c_char_p]
    L.spl_comm_create.restype = vp
    L.spl_comm_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.spl_comm_destroy.restype = None
    L.spl_comm_destroy.argtypes = [vp]
    L.spl_comm_rank.argtypes = [vp]
    L.spl_comm_world.argtypes = [vp]
    L.spl_allgather_slabs.argtypes = [vp, vp, vp, ctypes.c_uint64, vp]
    L.spl_allgather_slabs_p2p.argtypes = [vp, vp, vp, ctypes.c_uint64, vp]
    L.spl_gatherv_unpack_at.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, vp, ctypes.c_uint64, vp, ctypes.c_uint64,
                                        vp, vp, vp]
    L.spl_allgatherv_csr.argtypes = [vp, vp, vp, ctypes.c_uint64, vp, ctypes.c_uint64, vp, ctypes.c_uint64,
                                     u64p, u64p, vp]
    _lib = L
    return L


_shim = None


def shim():
    """The CPython front end (csrc/spl_pyshim.c): list[str] -> pinned UTF-8, list[list[int]] from the
    pinned CSR, one C call per encode.  Built by __graft_entry__.build(); no fallback."""
    global _shim
    if _shim is None:
        lib()                                   # libsplintr_hip.so first: the shim links against it
        from . import _spl_py
        _shim = _spl_py
    return _shim


def last_error() -> str:
    return (lib().spl_last_error() or b"").decode("utf-8", "replace")
