"""``device="cpu"``: the reference's default device (src/ginfinity/api.py:64-76), served by
the host implementation (csrc/gine_host.cpp, loaded as libgfy_host.so — built with the host
compiler alone, so this device works on a box without any ROCm runtime): plain C++, the same
rounding-point model as the kernels, threads over node blocks.  It exists so that the
drop-in surface behaves like the reference on a box without a GPU (plumbing, small inputs,
BASELINE configs[0]); it is not a fallback of the GPU path — an encoder loaded for
``"cuda"`` never comes here — and it does not touch ``oracle/``.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _native as native

_GFY_OF_NUMPY = {np.dtype(np.float16): native.GFY_F16, np.dtype(np.float32): native.GFY_F32,
                 np.dtype(np.float64): native.GFY_F64}
EMBEDDING_DIM = 128


class HostEncoder:
    """One gfy_host_encoder.  Same threading contract as the reference: serialized
    inference per instance (docs/OPERATIONS.md:43-47)."""

    def __init__(self, weight_pack: bytes, *, full_precision: bool,
                 threads: int | None = None) -> None:
        self._lib = native.host_library()
        self.full_precision = bool(full_precision)
        self.threads = int(threads) if threads else min(os.cpu_count() or 1, 16)
        handle = ctypes.c_void_p()
        native.check(self._lib.gfy_host_encoder_create(
            weight_pack, len(weight_pack),
            native.GFY_F32 if full_precision else native.GFY_F16, ctypes.byref(handle)),
            "gfy_host_encoder_create", self._lib)
        self._handle = handle

    def close(self) -> None:
        handle, self._handle = getattr(self, "_handle", None), None
        if handle:
            self._lib.gfy_host_encoder_destroy(handle)

    def __del__(self) -> None:  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass

    def encode_arrays(self, node_features: np.ndarray, edge_index: np.ndarray,
                      edge_types: np.ndarray, node_roles: np.ndarray | None, *,
                      embedding_dtype: np.dtype = np.dtype(np.float16),
                      normalise: bool = True) -> np.ndarray:
        """Host arrays of one micro-batch → [core_nodes, 128] in ``embedding_dtype``
        (api.py:236-260: forward, float64 normalise, one rounding, context rows dropped).
        Float kinds the library does not write are produced as float64 and cast, the
        reference's own order of operations."""
        features = np.ascontiguousarray(node_features, dtype=np.float32)
        edges = np.ascontiguousarray(edge_index, dtype=np.int32)
        types = np.ascontiguousarray(edge_types, dtype=np.uint8)
        nodes, count = int(features.shape[0]), int(types.shape[0])
        wanted = np.dtype(embedding_dtype)
        produced = wanted if wanted in _GFY_OF_NUMPY else np.dtype(np.float64)
        rows, kept = None, nodes
        if node_roles is not None and np.any(node_roles):
            core = np.asarray(node_roles) == 0
            kept = int(np.count_nonzero(core))
            rows = np.cumsum(core, dtype=np.int32) - np.int32(1)
            rows[~core] = -1
        out = np.empty((kept, EMBEDDING_DIM), dtype=produced)
        pointer = lambda array: None if array is None else array.ctypes.data_as(ctypes.c_void_p)
        native.check(self._lib.gfy_host_encode(
            self._handle, pointer(features), pointer(edges) if count else None,
            pointer(types) if count else None, nodes, count, pointer(rows), pointer(out),
            _GFY_OF_NUMPY[produced], 1 if normalise else 0, self.threads), "gfy_host_encode",
            self._lib)
        return out if produced == wanted else out.astype(wanted)
