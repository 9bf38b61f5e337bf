"""ctypes binding of libcfx.so (include/cfx.h).  There is NO fallback: if the shared object is missing
or fails to load, importing a codec raises - the product path never runs on anything but the HIP kernels."""
from __future__ import annotations

import ctypes
import os

from .build import LIB, build_lib

CFX_MAX_BATCH = 16

CFX_OK = 0
CFX_ERR_GATE = -8
ERR_NAMES = {
    -1: "CFX_ERR_NULL", -2: "CFX_ERR_SHAPE", -3: "CFX_ERR_ALIGN", -4: "CFX_ERR_CODEC",
    -5: "CFX_ERR_BATCH", -6: "CFX_ERR_LAUNCH", -7: "CFX_ERR_WORKSPACE", -8: "CFX_ERR_GATE", -9: "CFX_ERR_QUEUES",
}

FLAG_UPDATE_CACHE = 1
FLAG_NO_EF = 2


class CompItem(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("base", ctypes.c_void_p), ("new_base", ctypes.c_void_p), ("packet", ctypes.c_void_p)]


class DecompItem(ctypes.Structure):
    _fields_ = [("packet", ctypes.c_void_p), ("base", ctypes.c_void_p), ("recon", ctypes.c_void_p)]


# every symbol include/cfx.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("cfx_abi_version", ctypes.c_int, []),
    ("cfx_create", ctypes.c_void_p, [ctypes.c_int]),
    ("cfx_destroy", None, [ctypes.c_void_p]),
    ("cfx_last_error_string", ctypes.c_char_p, [ctypes.c_void_p]),
    ("cfx_set_rows_per_tile", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("cfx_packet_bytes", ctypes.c_size_t, [ctypes.c_int] * 4),
    ("cfx_workspace_bytes", ctypes.c_size_t, [ctypes.c_int] * 5),
    ("cfx_compress_batch", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.POINTER(CompItem), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("cfx_decompress_batch", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.POINTER(DecompItem), ctypes.c_void_p]),
    ("cfx_compress_batch_ex", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.POINTER(CompItem), ctypes.c_int, ctypes.POINTER(DecompItem),
                                             ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("cfx_compress_batch_gated", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.POINTER(CompItem), ctypes.c_int, ctypes.POINTER(DecompItem),
                                                ctypes.c_int, ctypes.POINTER(DecompItem), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("cfx_gate_errors", ctypes.c_int, [ctypes.c_void_p]),
    ("cfx_gate_recover", ctypes.c_int, [ctypes.c_void_p]),
    ("cfx_prepare", ctypes.c_int, [ctypes.c_void_p]),
    ("cfx_set_fused_finalize", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("cfx_set_stats_rows", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("cfx_set_gated_launch", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("cfx_set_lr_chain", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("cfx_set_lr_decode", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("cfx_hw_queues_ok", ctypes.c_int, []),
    ("cfx_set_allow_shared_queues", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("cfx_ipc_memory_kind", ctypes.c_int, [ctypes.c_void_p]),
    ("cfx_set_ipc_memory_kind", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("cfx_plan_set_pipe_unit_layers", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("cfx_compress", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("cfx_decompress", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    ("cfx_int2_quantize", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(CompItem), ctypes.c_void_p]),
    ("cfx_lr_packet_bytes", ctypes.c_size_t, [ctypes.c_int] * 4),
    ("cfx_lr_workspace_bytes", ctypes.c_size_t, [ctypes.c_int] * 5),
    ("cfx_lr_compress_batch", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.POINTER(CompItem), ctypes.POINTER(ctypes.c_void_p),
                                             ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("cfx_lr_decompress_batch", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.POINTER(DecompItem), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("cfx_profile_enable", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_int]),
    ("cfx_profile_read", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float), ctypes.c_int]),
    ("cfx_kernel_name", ctypes.c_char_p, [ctypes.c_int]),
    ("cfx_plan_create", ctypes.c_void_p, [ctypes.c_void_p]),
    ("cfx_plan_destroy", None, [ctypes.c_void_p]),
    ("cfx_plan_add_compress", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.POINTER(CompItem), ctypes.c_void_p, ctypes.c_size_t]),
    ("cfx_plan_add_compress_ex", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.POINTER(CompItem), ctypes.c_int, ctypes.POINTER(DecompItem),
                                                ctypes.c_void_p, ctypes.c_size_t]),
    ("cfx_plan_add_compress_gated", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, ctypes.POINTER(CompItem), ctypes.c_int, ctypes.POINTER(DecompItem),
                                                   ctypes.c_int, ctypes.POINTER(DecompItem), ctypes.c_void_p, ctypes.c_size_t]),
    ("cfx_plan_add_decompress", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.POINTER(DecompItem)]),
    ("cfx_plan_add_lr_compress", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.POINTER(CompItem), ctypes.POINTER(ctypes.c_void_p),
                                                ctypes.c_void_p, ctypes.c_size_t]),
    ("cfx_plan_add_lr_decompress", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.POINTER(DecompItem), ctypes.c_void_p, ctypes.c_size_t]),
    ("cfx_plan_set_exchange_stream", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("cfx_plan_use_exchange_stream", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    ("cfx_plan_add_exchange_layer", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, ctypes.POINTER(CompItem), ctypes.c_int, ctypes.POINTER(DecompItem),
                                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                                   ctypes.c_void_p, ctypes.c_size_t]),
    ("cfx_plan_add_exchange_layer_p2p", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                       ctypes.c_int, ctypes.POINTER(CompItem), ctypes.c_int, ctypes.POINTER(DecompItem),
                                                       ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_size_t]),
    ("cfx_plan_add_p2p_sync", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    ("cfx_ipc_alloc", ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    ("cfx_ipc_open", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    ("cfx_ipc_close", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    ("cfx_ipc_free", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    ("cfx_plan_add_all_gather", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    ("cfx_plan_add_wait", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("cfx_plan_add_ring_hop", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    ("cfx_plan_set_input", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    ("cfx_comm_ring_hop", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("cfx_rccl_load", ctypes.c_int, [ctypes.c_char_p]),
    ("cfx_comm_unique_id", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    ("cfx_comm_create", ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    ("cfx_comm_destroy", None, [ctypes.c_void_p]),
    ("cfx_comm_all_gather", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("cfx_plan_copy_op", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    ("cfx_plan_size", ctypes.c_int, [ctypes.c_void_p]),
    ("cfx_plan_run", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    ("cfx_plan_run_x", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p]),
    ("cfx_plan_run_async", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p]),
    ("cfx_plan_join", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    ("cfx_plan_run_pipelined", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    ("cfx_plan_finalize", ctypes.c_int, [ctypes.c_void_p]),
    ("cfx_residual2_delta", ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_size_t, ctypes.c_void_p]),
    ("cfx_residual2_update", ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p]),
    ("cfx_attn_merge", ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int] * 6 + [ctypes.c_void_p]),
    ("cfx_copy_probe", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("cfx_binary_rank_packet_bytes", ctypes.c_size_t, [ctypes.c_int] * 3),
    ("cfx_binary_rank_workspace_bytes", ctypes.c_size_t, [ctypes.c_int] * 4),
    ("cfx_binary_rank_compress_batch", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                      ctypes.POINTER(CompItem), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("cfx_binary_rank_decompress_batch", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                        ctypes.POINTER(DecompItem), ctypes.c_void_p]),
    ("cfx_plan_flags", ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]),
    ("cfx_plan_add_flag_wait", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("cfx_plan_add_flag_set", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("cfx_plan_set_pre_flag", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    ("cfx_plan_epoch", ctypes.c_uint, [ctypes.c_void_p]),
    ("cfx_plan_run_lane", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint)]),
    ("cfx_plan_lane_begin", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint)]),
    ("cfx_flag_set", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]),
    ("cfx_flag_wait", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]),
    ("cfx_stream_create_masked", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    ("cfx_stream_destroy", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    ("cfx_set_gate_timeout_ms", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("cfx_attn_merge_wait", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]),
]

# include/cfx_dev.h: exported by the DEVELOPER library only (libcfx_dev.so, -DCFX_DEV_PROBES).  Bound when - and only when - that library
# was loaded in place of the product one (CFX_LIBCFX_PATH, or use_dev_library() before the first load): tools/*_stamps.py, tests/tagwrap_child.py
DEV_SYMBOLS = [
    ("cfx_dev_stamps", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    ("cfx_dev_set_launch_tags", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]),
    ("cfx_dev_set_probe", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
]

_lib = None


class CfxError(RuntimeError):
    pass


def use_dev_library() -> str:
    """Developer tools: make this process load libcfx_dev.so (built on demand) instead of libcfx.so.  Must run before the first load()."""
    if _lib is not None:
        raise CfxError("the library is loaded already: call use_dev_library() first")
    from .build import build_lib
    path = build_lib(dev_probes=True)
    os.environ["CFX_LIBCFX_PATH"] = path
    return path


def load(build_if_missing: bool = True) -> ctypes.CDLL:
    """Load libcfx.so, binding every declared symbol.  Raises (never falls back) when unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        if not build_if_missing:
            raise CfxError(f"{LIB} is missing: run `python -m compactfusion_amd.build`")
        build_lib()
    lib = ctypes.CDLL(os.environ.get("CFX_LIBCFX_PATH") or LIB)     # (developer override: A/B two builds on one box)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)     # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    if lib.cfx_abi_version() != 1:
        raise CfxError("libcfx.so ABI version mismatch")
    if hasattr(lib, "cfx_dev_stamps"):                               # the developer library: its extra entry points too
        for name, res, args in DEV_SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib
