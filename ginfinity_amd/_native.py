"""ctypes binding of libgfy.so (the C ABI in include/gfy.h).

There is no fallback: if the HIP library is missing or a call fails, the
caller gets an exception.  The product never routes around the kernels.
"""
from __future__ import annotations

import ctypes
from ctypes import (POINTER, Structure, c_char_p, c_int, c_int32, c_int64, c_size_t,
                    c_uint32, c_void_p)
import os
from pathlib import Path

#: the in-tree build (python -m ginfinity_amd.build).  The library is opened BY THIS PATH:
#: LD_LIBRARY_PATH does not redirect it.  GFY_LIBRARY names another build of the same sources for
#: a side-by-side measurement (tools/: a diagnostic build next to the default one); it must pass
#: the same ABI and symbol checks.
LIBRARY_PATH = Path(os.environ.get("GFY_LIBRARY") or
                    Path(__file__).resolve().parent / "csrc" / "libgfy.so")
#: gine_host.cpp + gfy_base.cpp built with the host compiler: no HIP runtime behind it
HOST_LIBRARY_PATH = Path(__file__).resolve().parent / "csrc" / "libgfy_host.so"

GFY_OK = 0
GFY_ERR_INVALID, GFY_ERR_UNSUPPORTED, GFY_ERR_HIP, GFY_ERR_WORKSPACE = 1, 2, 3, 4
GFY_F16, GFY_F32, GFY_F64 = 0, 1, 2
GFY_L2, GFY_COSINE = 0, 1
GFY_OPT_SEPARATE_HEAD = 2
GFY_OPT_LAYER_KERNEL = 3
GFY_OPT_STAGGER = 4
GFY_OPT_PRIORITY = 6
GFY_MAX_BATCH_SHARDS = 16
GFY_TAP_H, GFY_TAP_Z, GFY_TAP_V, GFY_TAP_W, GFY_TAP_Y = 0, 1, 2, 3, 4
ABI_VERSION = 4

class GfyShard(Structure):
    """``gfy_shard`` of include/gfy.h: one shard of a batch (device pointers)."""
    _fields_ = [("node_features", c_void_p), ("edge_index", c_void_p),
                ("edge_types", c_void_p), ("out_rows", c_void_p), ("out", c_void_p),
                ("n_nodes", c_int64), ("n_edges", c_int64),
                ("node_ptr", c_void_p), ("edge_ptr", c_void_p), ("n_records", c_int64)]


#: every symbol include/gfy.h declares: (restype, argtypes)
SIGNATURES: dict[str, tuple] = {
    "gfy_last_error": (c_char_p, []),
    "gfy_abi_version": (c_int, []),
    "gfy_weight_pack_bytes": (c_size_t, [c_uint32] * 5),
    "gfy_pack_microbatch": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p,
                                    c_int64, POINTER(c_int64), POINTER(c_int64)]),
    "gfy_upload_ring_create": (c_int, [c_int, POINTER(c_void_p)]),
    "gfy_upload_ring_destroy": (None, [c_void_p]),
    "gfy_upload_async": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p,
                                 c_void_p]),
    "gfy_upload_wait": (c_int, [c_void_p, c_int]),
    "gfy_encoder_create": (c_int, [c_void_p, c_size_t, c_int, c_int,
                                   POINTER(c_void_p)]),
    "gfy_encoder_destroy": (None, [c_void_p]),
    "gfy_build_graphs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                 c_int64, c_int64, c_int, c_int, c_int, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gfy_csr_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "gfy_build_csr": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gfy_encode_workspace_bytes": (c_size_t, [c_void_p, c_int64, c_int64]),
    "gfy_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                           c_int64, c_int64, c_void_p, c_void_p, c_int, c_int,
                           c_void_p, c_size_t, c_void_p]),
    "gfy_encode_hidden": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_int64, c_int64, c_int, c_void_p,
                                  c_void_p, c_size_t, c_void_p]),
    "gfy_encode_coo_workspace_bytes": (c_size_t, [c_void_p, c_int64, c_int64]),
    "gfy_encode_coo_clear_bytes": (c_size_t, [c_int64]),
    "gfy_encode_coo_prepare": (c_int, [c_void_p, c_size_t, c_int64, c_void_p]),
    "gfy_encode_coo": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                               c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t,
                               c_void_p]),
    "gfy_encode_coo_batch_workspace_bytes": (c_size_t, [c_void_p, POINTER(GfyShard), c_int]),
    "gfy_encode_coo_batch_clear_bytes": (c_size_t, [POINTER(GfyShard), c_int]),
    "gfy_encode_coo_batch": (c_int, [c_void_p, POINTER(GfyShard), c_int, c_int, c_int,
                                     c_void_p, c_size_t, c_void_p]),
    "gfy_host_encoder_create": (c_int, [c_void_p, c_size_t, c_int, POINTER(c_void_p)]),
    "gfy_host_encoder_destroy": (None, [c_void_p]),
    "gfy_host_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                c_void_p, c_void_p, c_int, c_int, c_int]),
    "gfy_debug_layer_workspace_bytes": (c_size_t, [c_void_p, c_int64, c_int64]),
    "gfy_debug_layer": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_int64, c_int64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gfy_encoder_set_timing": (c_int, [c_void_p, c_int]),
    "gfy_encoder_set_option": (c_int, [c_void_p, c_int, c_int]),
    "gfy_encoder_last_layer_kernel": (c_int, [c_void_p]),
    "gfy_encoder_get_timing": (c_int, [c_void_p, c_void_p, c_int,
                                       POINTER(c_int)]),
    "gfy_pairwise_dense": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
    "gfy_pairwise_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "gfy_pairwise_nearest": (c_int, [c_void_p, c_int64, c_void_p, c_int64,
                                     c_int, c_int64, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
    "gfy_pairwise_nearest_window": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int64,
                                            c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
}


#: what device="cpu" calls (a subset of SIGNATURES; libgfy.so exports them too)
HOST_SYMBOLS = ("gfy_last_error", "gfy_abi_version", "gfy_weight_pack_bytes", "gfy_pack_microbatch",
                "gfy_host_encoder_create", "gfy_host_encoder_destroy", "gfy_host_encode")


class NativeLibraryError(RuntimeError):
    """libgfy.so is missing, stale or reported a failure."""


_library: ctypes.CDLL | None = None


def library() -> ctypes.CDLL:
    """Load libgfy.so once; raise loudly when it is not built."""
    global _library
    if _library is not None:
        return _library
    if not LIBRARY_PATH.is_file():
        raise NativeLibraryError(
            f"HIP extension not built: {LIBRARY_PATH} is missing. Run "
            "`python -m ginfinity_amd.build` (needs hipcc, gfx950). There is "
            "no CPU fallback in this package.")
    try:
        lib = ctypes.CDLL(str(LIBRARY_PATH), mode=ctypes.RTLD_GLOBAL)
    except OSError as error:
        raise NativeLibraryError(
            f"cannot load {LIBRARY_PATH}: {error}") from error
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            function = getattr(lib, name)
        except AttributeError as error:
            raise NativeLibraryError(
                f"{LIBRARY_PATH} does not export {name}; rebuild it") from error
        function.restype = restype
        function.argtypes = argtypes
    if lib.gfy_abi_version() != ABI_VERSION:
        raise NativeLibraryError(
            f"{LIBRARY_PATH} has ABI {lib.gfy_abi_version()}, "
            f"expected {ABI_VERSION}; rebuild it")
    _library = lib
    return lib


_host_library: ctypes.CDLL | None = None


def host_library() -> ctypes.CDLL:
    """libgfy_host.so: the host implementation by itself (no HIP runtime is loaded)."""
    global _host_library
    if _host_library is not None:
        return _host_library
    if not HOST_LIBRARY_PATH.is_file():
        raise NativeLibraryError(
            f"host library not built: {HOST_LIBRARY_PATH} is missing. Run "
            "`python -m ginfinity_amd.build --host-only` (a C++17 host compiler; no hipcc, no "
            "ROCm runtime).")
    try:
        lib = ctypes.CDLL(str(HOST_LIBRARY_PATH))
    except OSError as error:
        raise NativeLibraryError(f"cannot load {HOST_LIBRARY_PATH}: {error}") from error
    for name in HOST_SYMBOLS:
        function = getattr(lib, name)
        function.restype, function.argtypes = SIGNATURES[name]
    if lib.gfy_abi_version() != ABI_VERSION:
        raise NativeLibraryError(
            f"{HOST_LIBRARY_PATH} has ABI {lib.gfy_abi_version()}, expected {ABI_VERSION}")
    _host_library = lib
    return lib


def check(status: int, where: str, lib: ctypes.CDLL | None = None) -> None:
    """Turn a gfy status into the reference's exception conventions:
    invalid arguments → ValueError, everything else → NativeLibraryError.  ``lib``: the
    library the call went to (its per-thread error string); libgfy.so by default."""
    if status == GFY_OK:
        return
    message = (lib or library()).gfy_last_error().decode("utf-8", "replace")
    text = f"{where}: {message or 'status ' + str(status)}"
    if status == GFY_ERR_INVALID:
        raise ValueError(text)
    raise NativeLibraryError(text)
