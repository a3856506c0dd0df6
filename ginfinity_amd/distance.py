"""All-pairs L2 / cosine distance over 128-d fp16 embeddings on the matrix cores.

The reference has no implementation of this step (its aligner is the external
``ginfinity-sw`` package; src/ginfinity/api.py:47-50 only exports scoring
parameters), so the semantics are defined here and in include/gfy.h:

    L2      D_ij = sqrt(max(|a_i|² + |b_j|² − 2 a_i·b_j, 0))
    cosine  S_ij = a_i·b_j / (max(|a_i|, 1e-12) · max(|b_j|, 1e-12))

Inputs are fp16 device tensors as produced by ``Ginfinity.encode_graphs_device``;
products are exact, accumulation is fp32 (MFMA).  ``nearest`` never
materialises the N×M matrix.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _native as native

_METRICS = {"l2": native.GFY_L2, "cosine": native.GFY_COSINE}


def _prepare(rows, device: torch.device | None) -> torch.Tensor:
    if isinstance(rows, np.ndarray):
        rows = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float16))
    if rows.dtype != torch.float16:
        raise ValueError("embeddings must be float16")
    if rows.dim() != 2 or rows.shape[1] != 128:
        raise ValueError("embeddings must have shape (rows, 128)")
    if rows.device.type != "cuda":
        rows = rows.to(device if device is not None else "cuda")
    return rows.contiguous()


def _metric(name: str) -> int:
    try:
        return _METRICS[name]
    except KeyError:
        raise ValueError(f"metric must be one of {sorted(_METRICS)}") from None


def pairwise(a, b=None, *, metric: str = "l2") -> torch.Tensor:
    """Dense [n, m] float32 distance (l2) or similarity (cosine) block.
    Meant for blocks that fit comfortably in memory (n·m·4 bytes)."""
    a = _prepare(a, None)
    b = a if b is None else _prepare(b, a.device)
    lib = native.library()
    n, m = a.shape[0], b.shape[0]
    with torch.cuda.device(a.device):
        out = torch.empty((n, m), dtype=torch.float32, device=a.device)
        scratch = torch.empty(lib.gfy_pairwise_workspace_bytes(n, m),
                              dtype=torch.uint8, device=a.device)
        native.check(lib.gfy_pairwise_dense(
            a.data_ptr(), n, b.data_ptr(), m, _metric(metric), out.data_ptr(),
            scratch.data_ptr(), scratch.numel(),
            torch.cuda.current_stream(a.device).cuda_stream),
            "gfy_pairwise_dense")
    return out


class NearestWorkspace:
    """Scratch memory and result arrays of ``nearest`` kept across calls (the chunked
    cross-shard search calls it once per rank's piece and chunk: allocating per call costs
    more than small searches)."""

    def __init__(self) -> None:
        self.scratch: torch.Tensor | None = None
        self.values: torch.Tensor | None = None
        self.indices: torch.Tensor | None = None

    def buffers(self, device, rows: int, scratch_bytes: int):
        if self.scratch is None or self.scratch.numel() < scratch_bytes or \
                self.scratch.device != device:
            self.scratch = torch.empty(max(scratch_bytes, 1), dtype=torch.uint8, device=device)
        if self.values is None or self.values.numel() < rows or self.values.device != device:
            self.values = torch.empty(max(rows, 1), dtype=torch.float32, device=device)
            self.indices = torch.empty(max(rows, 1), dtype=torch.int32, device=device)
        return self.scratch, self.values[:rows], self.indices[:rows]


def nearest(a, b=None, *, metric: str = "l2", exclude_self: bool = False,
            exclude_offset: int | None = None, window_first: int | None = None,
            workspace: NearestWorkspace | None = None
            ) -> tuple[torch.Tensor, torch.Tensor]:
    """For every row of ``a`` the closest row of ``b`` (smallest L2 distance /
    largest cosine): ``(values float32 [n], indices int32 [n])``; ties go to
    the lowest index.  ``exclude_self`` (with ``b`` omitted or identical to
    ``a``) skips the pair (i, i); ``exclude_offset=k`` skips (i, i+k) — ``a`` is a
    row block of ``b`` starting at row k; ``window_first=k`` is the opposite case:
    ``b`` is rows [k, k + m) of ``a`` and every row skips itself (the cross-shard
    search of a rank's own piece, one call).  With ``workspace`` the returned
    tensors are views of its buffers, valid until its next use."""
    a = _prepare(a, None)
    b = a if b is None else _prepare(b, a.device)
    if window_first is not None and (exclude_self or exclude_offset is not None):
        raise ValueError("window_first excludes the other exclusion arguments")
    if exclude_offset is None:
        exclude_offset = 0 if exclude_self else -1
    lib = native.library()
    n, m = a.shape[0], b.shape[0]
    with torch.cuda.device(a.device):
        need = lib.gfy_pairwise_workspace_bytes(n, m)
        if workspace is None:
            values = torch.empty(n, dtype=torch.float32, device=a.device)
            indices = torch.empty(n, dtype=torch.int32, device=a.device)
            scratch = torch.empty(need, dtype=torch.uint8, device=a.device)
        else:
            scratch, values, indices = workspace.buffers(a.device, n, need)
        stream = torch.cuda.current_stream(a.device).cuda_stream
        if window_first is None:
            native.check(lib.gfy_pairwise_nearest(
                a.data_ptr(), n, b.data_ptr(), m, _metric(metric),
                int(exclude_offset), values.data_ptr(), indices.data_ptr(),
                scratch.data_ptr(), scratch.numel(), stream), "gfy_pairwise_nearest")
        else:
            native.check(lib.gfy_pairwise_nearest_window(
                a.data_ptr(), n, b.data_ptr(), m, _metric(metric), int(window_first),
                values.data_ptr(), indices.data_ptr(), scratch.data_ptr(), scratch.numel(),
                stream), "gfy_pairwise_nearest_window")
    return values, indices


__all__ = ["pairwise", "nearest", "NearestWorkspace"]
