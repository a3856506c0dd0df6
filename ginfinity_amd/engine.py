"""Device-side driver of the encode hot path: owns the gfy_encoder handle and
moves one micro-batch through CSR build → GINE forward → normalise.

This is the counterpart of the body of ``Ginfinity._run_graph_shard``
(reference: src/ginfinity/api.py:236-252).  PyTorch is used for device memory,
streams and H2D/D2H copies only; every arithmetic step is a libgfy kernel.
"""
from __future__ import annotations

import ctypes
import warnings
from dataclasses import dataclass

import numpy as np
import torch

from . import _native as native

_TORCH_OF_NUMPY = {np.dtype(np.float16): (torch.float16, native.GFY_F16),
                   np.dtype(np.float32): (torch.float32, native.GFY_F32),
                   np.dtype(np.float64): (torch.float64, native.GFY_F64)}
_GFY_OF_TORCH = {torch.float16: native.GFY_F16, torch.float32: native.GFY_F32,
                 torch.float64: native.GFY_F64}
EMBEDDING_DIM = 128


def device_output_dtype(embedding_dtype: np.dtype) -> tuple[torch.dtype, int, bool]:
    """(torch dtype, gfy code, exact) for a requested numpy embedding dtype.
    Float kinds the kernels do not write directly are produced as float64 on
    the device and cast on the host — the reference's own order of operations
    (api.py:250-259: float64 normalise, then ``astype``)."""
    key = np.dtype(embedding_dtype)
    if key in _TORCH_OF_NUMPY:
        return (*_TORCH_OF_NUMPY[key], True)
    return torch.float64, native.GFY_F64, False


@dataclass
class DeviceCsr:
    row_ptr: torch.Tensor   # int32 [N+1]
    col: torch.Tensor       # int32 [E]
    typ: torch.Tensor       # uint8 [E]
    nodes: int
    edges: int


def _ptr(tensor: torch.Tensor | None) -> int | None:
    return None if tensor is None else tensor.data_ptr()


#: a record's edge list is read by every workgroup that owns 256 of its rows: beyond this many
#: edges in ONE record the counting kernel (global atomics, one pass over the list) is cheaper
MAX_RECORD_EDGES = 1 << 16


def records_pay(node_ptr: np.ndarray, edge_ptr: np.ndarray) -> bool:
    """Whether a micro-batch's record boundaries should travel with it: every record's edge
    list is short enough for the record-range set-up (csrc/csr_records.inc)."""
    if len(node_ptr) < 2 or len(node_ptr) != len(edge_ptr):
        return False
    return int(np.diff(edge_ptr).max()) <= MAX_RECORD_EDGES


def attach_records(edge_index: torch.Tensor, node_ptr: torch.Tensor,
                   edge_ptr: torch.Tensor) -> torch.Tensor:
    """Record boundaries (device int64 tensors of records + 1 entries) ride on the micro-batch's
    edge_index tensor, so the (features, edge_index, edge_types, out_rows, out) tuples every
    caller passes around stay what they are."""
    assert node_ptr.dtype == torch.int64 and edge_ptr.dtype == torch.int64
    assert node_ptr.numel() == edge_ptr.numel() >= 2
    edge_index.gfy_records = (node_ptr, edge_ptr)
    return edge_index


def records_of(edge_index: torch.Tensor):
    return getattr(edge_index, "gfy_records", None)


class DeviceEncoder:
    """One gfy_encoder on one GPU.  Not thread-safe (docs/OPERATIONS.md:43-47
    of the reference: serialized inference per instance)."""

    def __init__(self, weight_pack: bytes, *, full_precision: bool,
                 device: torch.device) -> None:
        self._lib = native.library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("DeviceEncoder needs a HIP ('cuda') device")
        index = (self.device.index if self.device.index is not None
                 else torch.cuda.current_device())
        self.device = torch.device("cuda", index)
        self.full_precision = bool(full_precision)
        self.model_dtype = torch.float32 if full_precision else torch.float16
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            torch.empty(1, device=self.device)   # make sure the HIP context is live
            native.check(self._lib.gfy_encoder_create(
                weight_pack, len(weight_pack),
                native.GFY_F32 if full_precision else native.GFY_F16,
                index, ctypes.byref(handle)), "gfy_encoder_create")
        self._handle = handle
        self._weight_pack = weight_pack            # a second lane's encoder is made from it (twin)
        self._options: dict[int, int] = {}
        self._workspace: torch.Tensor | None = None
        # gfy_encode_coo's own workspace: its leading counters are zero between calls as long
        # as nobody else writes to it (include/gfy.h), so it is never shared with _scratch
        self._coo_workspace: torch.Tensor | None = None
        self._coo_clean_nodes = 0      # the counters are known to be zero for n <= this
        # the zeroed counters are only "zero again" for work enqueued behind the last call:
        # a call on another stream first waits for that call (event), see encode_coo
        self._coo_done: "torch.cuda.Event | None" = None
        self._coo_stream: int | None = None

    # -- lifetime ---------------------------------------------------------------
    def close(self) -> None:
        handle, self._handle = getattr(self, "_handle", None), None
        if handle:
            self._lib.gfy_encoder_destroy(handle)

    def __del__(self) -> None:  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ------------------------------------------------------------------
    def _scratch(self, nbytes: int) -> torch.Tensor:
        if self._workspace is None or self._workspace.numel() < nbytes:
            self._workspace = torch.empty(
                max(nbytes, 1 << 20), dtype=torch.uint8, device=self.device)
        return self._workspace

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    # -- kernels ----------------------------------------------------------------------
    def build_csr(self, edge_index: torch.Tensor, edge_types: torch.Tensor,
                  nodes: int, *, out: DeviceCsr | None = None) -> DeviceCsr:
        """COO (int32 [2,E], uint8 [E]) → destination-major CSR on the device.
        ``out`` reuses the buffers of an earlier CSR of the same size (steady-state
        loops: no allocator traffic)."""
        edges = int(edge_types.numel())
        assert edge_index.dtype == torch.int32 and edge_index.is_contiguous()
        assert edge_types.dtype == torch.uint8 and edge_types.is_contiguous()
        assert tuple(edge_index.shape) == (2, edges)
        with torch.cuda.device(self.device):
            if out is not None and (out.nodes, out.edges) == (nodes, edges):
                row_ptr, col, typ = out.row_ptr, out.col, out.typ
            else:
                row_ptr = torch.empty(nodes + 1, dtype=torch.int32, device=self.device)
                col = torch.empty(max(edges, 1), dtype=torch.int32, device=self.device)
                typ = torch.empty(max(edges, 1), dtype=torch.uint8, device=self.device)
            need = self._lib.gfy_csr_workspace_bytes(nodes, edges)
            scratch = self._scratch(need)
            native.check(self._lib.gfy_build_csr(
                _ptr(edge_index) if edges else None,
                _ptr(edge_types) if edges else None, nodes, edges,
                _ptr(row_ptr), _ptr(col), _ptr(typ), _ptr(scratch),
                scratch.numel(), self._stream()), "gfy_build_csr")
        return DeviceCsr(row_ptr, col, typ, nodes, edges)

    def build_graphs(self, bases: torch.Tensor, marks: torch.Tensor,
                     node_ptr: torch.Tensor, edge_ptr: torch.Tensor,
                     positional: torch.Tensor | None, nodes: int, edges: int, *,
                     struct_states: int, skip2: bool
                     ) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """Sequence / dot-bracket text of unsliced records (device uint8 [N] each, int64
        [R+1] offsets) → (node_features f32 [N,F], edge_index int32 [2,E], edge_types uint8
        [E], first_invalid int32 [1]) on the device: ``GraphBuilder._build_full`` for every
        record (graph.py:494-561), bit-identical arrays.  ``first_invalid`` is -1 when every
        record was a balanced structure consistent with ``edge_ptr``; read it after the
        stream has run."""
        records = int(node_ptr.numel()) - 1
        columns = 0 if positional is None else int(positional.shape[1])
        assert bases.dtype == torch.uint8 and marks.dtype == torch.uint8
        assert int(bases.numel()) == nodes == int(marks.numel())
        assert node_ptr.dtype == torch.int64 and edge_ptr.dtype == torch.int64
        assert int(edge_ptr.numel()) == records + 1
        assert positional is None or (positional.dtype == torch.float32
                                      and positional.is_contiguous()
                                      and positional.shape[0] == nodes)
        with torch.cuda.device(self.device):
            features = torch.empty((nodes, 4 + struct_states + columns),
                                   dtype=torch.float32, device=self.device)
            edge_index = torch.empty((2, edges), dtype=torch.int32, device=self.device)
            edge_types = torch.empty(edges, dtype=torch.uint8, device=self.device)
            first_invalid = torch.empty(1, dtype=torch.int32, device=self.device)
            native.check(self._lib.gfy_build_graphs(
                _ptr(bases), _ptr(marks), _ptr(node_ptr), _ptr(edge_ptr), records, nodes,
                edges, struct_states, columns, 1 if skip2 else 0, _ptr(positional),
                _ptr(features), _ptr(edge_index), _ptr(edge_types), _ptr(first_invalid),
                self._stream()), "gfy_build_graphs")
        return features, edge_index, edge_types, first_invalid

    def encode(self, node_features: torch.Tensor, csr: DeviceCsr, *,
               out_rows: torch.Tensor | None = None,
               n_out: int | None = None,
               out_dtype: torch.dtype = torch.float16,
               normalise: bool = True,
               out: torch.Tensor | None = None) -> torch.Tensor:
        """GINE forward (+ float64 L2 normalise) → [n_out, 128] on the device."""
        nodes = csr.nodes
        assert node_features.dtype == torch.float32 and node_features.is_contiguous()
        assert node_features.shape[0] == nodes
        rows = nodes if n_out is None else int(n_out)
        with torch.cuda.device(self.device):
            if out is None:
                out = torch.empty((rows, EMBEDDING_DIM), dtype=out_dtype,
                                  device=self.device)
            assert out.is_contiguous() and out.shape == (rows, EMBEDDING_DIM)
            need = self._lib.gfy_encode_workspace_bytes(
                self._handle, nodes, csr.edges)
            scratch = self._scratch(need)
            native.check(self._lib.gfy_encode(
                self._handle, _ptr(node_features), _ptr(csr.row_ptr),
                _ptr(csr.col), _ptr(csr.typ), nodes, csr.edges, _ptr(out_rows),
                _ptr(out), _GFY_OF_TORCH[out.dtype], 1 if normalise else 0,
                _ptr(scratch), scratch.numel(), self._stream()), "gfy_encode")
        return out

    def _coo_scratch(self, need: int, rows: int, stream) -> torch.Tensor:
        """The encoder's ONE workspace of ``gfy_encode_coo`` / ``gfy_encode_coo_batch`` calls,
        with its leading counters zero for a call over ``rows`` padded rows (include/gfy.h: they
        are zero again after every call, so only a call LARGER than every call since the last
        clearing finds other arrays where its counters will be), and ordered behind the last
        call when that ran on another stream."""
        if self._coo_workspace is None or self._coo_workspace.numel() < need:
            self._coo_workspace = torch.zeros(max(need, 1 << 20), dtype=torch.uint8,
                                              device=self.device)
            self._coo_clean_nodes = 1 << 62
            self._coo_done, self._coo_stream = None, None   # zero-filled on THIS stream
        elif self._coo_stream is not None and self._coo_stream != stream.cuda_stream:
            stream.wait_event(self._coo_done)     # the last call ran on another stream
        scratch = self._coo_workspace
        if rows > self._coo_clean_nodes:
            native.check(self._lib.gfy_encode_coo_prepare(
                _ptr(scratch), scratch.numel(), rows, stream.cuda_stream),
                "gfy_encode_coo_prepare")
        self._coo_clean_nodes = 0               # until the call has been enqueued
        return scratch

    def _coo_enqueued(self, rows: int, stream) -> None:
        self._coo_clean_nodes = rows
        if self._coo_done is None:
            self._coo_done = torch.cuda.Event()
        self._coo_done.record(stream)
        self._coo_stream = stream.cuda_stream

    def encode_coo(self, node_features: torch.Tensor, edge_index: torch.Tensor,
                   edge_types: torch.Tensor, *, out_rows: torch.Tensor | None = None,
                   n_out: int | None = None, out_dtype: torch.dtype = torch.float16,
                   normalise: bool = True, out: torch.Tensor | None = None) -> torch.Tensor:
        """COO in → [n_out, 128] embeddings on the device: the body of
        ``Ginfinity._run_graph_shard`` (api.py:236-252) as ONE C-ABI call, the CSR build
        fused into the encoder's setup (3 + 4 launches for the fp16 model)."""
        nodes = int(node_features.shape[0])
        edges = int(edge_types.numel())
        assert node_features.dtype == torch.float32 and node_features.is_contiguous()
        assert edge_index.dtype == torch.int32 and edge_index.is_contiguous()
        assert edge_types.dtype == torch.uint8 and edge_types.is_contiguous()
        assert tuple(edge_index.shape) == (2, edges)
        rows = nodes if n_out is None else int(n_out)
        lib = self._lib
        with torch.cuda.device(self.device):
            if out is None:
                out = torch.empty((rows, EMBEDDING_DIM), dtype=out_dtype, device=self.device)
            assert out.is_contiguous() and out.shape == (rows, EMBEDDING_DIM)
            need = lib.gfy_encode_coo_workspace_bytes(self._handle, nodes, edges)
            stream = torch.cuda.current_stream(self.device)
            scratch = self._coo_scratch(need, -(-nodes // 32) * 32, stream)
            native.check(lib.gfy_encode_coo(
                self._handle, _ptr(node_features), _ptr(edge_index) if edges else None,
                _ptr(edge_types) if edges else None, nodes, edges, _ptr(out_rows), _ptr(out),
                _GFY_OF_TORCH[out.dtype], 1 if normalise else 0, _ptr(scratch),
                scratch.numel(), stream.cuda_stream), "gfy_encode_coo")
            self._coo_enqueued(-(-nodes // 32) * 32, stream)
        return out

    def prepare_step(self, node_features: torch.Tensor, edge_index: torch.Tensor,
                     edge_types: torch.Tensor, out: torch.Tensor):
        """CSR build + encode of one device-resident shard as a pre-bound callable
        ``step(stream_handle)``: buffers, workspaces and pointers are resolved once, each
        call is ONE C-ABI call (gfy_encode_coo on a workspace of its own, cleared here) and
        nothing else.  For steady-state loops over same-sized
        shards (bench.py): per-step Python overhead drops from ~50 us to a few us.  The
        calling thread's current device must be this encoder's device (the C ABI launches on
        the caller's device; ``torch.cuda.set_device`` once, as bench.py does)."""
        nodes = int(node_features.shape[0])
        edges = int(edge_types.numel())
        assert node_features.dtype == torch.float32 and node_features.is_contiguous()
        assert edge_index.dtype == torch.int32 and tuple(edge_index.shape) == (2, edges)
        assert edge_types.dtype == torch.uint8 and edge_index.is_contiguous()
        assert out.is_contiguous() and tuple(out.shape) == (nodes, EMBEDDING_DIM)
        lib, handle = self._lib, self._handle
        with torch.cuda.device(self.device):
            need = lib.gfy_encode_coo_workspace_bytes(handle, nodes, edges)
            scratch = torch.zeros(need, dtype=torch.uint8, device=self.device)   # cleared once
            # the zero-fill ran on torch's current stream; the step will be launched on streams
            # of the caller's choice (non-blocking ones do not wait for it): finish it now
            torch.cuda.current_stream(self.device).synchronize()
        encode = lib.gfy_encode_coo
        p_ei = _ptr(edge_index) if edges else None
        p_et = _ptr(edge_types) if edges else None
        p_x, p_out, p_ws, ws_bytes = _ptr(node_features), _ptr(out), _ptr(scratch), scratch.numel()
        out_code = _GFY_OF_TORCH[out.dtype]
        keep = (node_features, edge_index, edge_types, out, scratch)   # keep buffers alive

        def step(stream_handle: int, _keep=keep) -> None:
            status = encode(handle, p_x, p_ei, p_et, nodes, edges, None, p_out, out_code, 1,
                            p_ws, ws_bytes, stream_handle)
            if status != 0:
                native.check(status, "gfy_encode_coo")
        return step

    # -- a batch of shards in one sequence of launches (gfy_encode_coo_batch) -------------------
    def _shard_array(self, shards) -> "ctypes.Array":
        """``shards``: sequence of (node_features, edge_index, edge_types, out_rows or None,
        out) device tensors -> the host array of gfy_shard descriptors."""
        if not 1 <= len(shards) <= native.GFY_MAX_BATCH_SHARDS:
            raise ValueError(f"a batch holds 1..{native.GFY_MAX_BATCH_SHARDS} shards, "
                             f"got {len(shards)}")
        array = (native.GfyShard * len(shards))()
        for slot, (x, ei, et, rows, out) in zip(array, shards):
            nodes, edges = int(x.shape[0]), int(et.numel())
            assert x.dtype == torch.float32 and x.is_contiguous()
            assert ei.dtype == torch.int32 and ei.is_contiguous() and tuple(ei.shape) == (2, edges)
            assert et.dtype == torch.uint8 and et.is_contiguous()
            assert out.is_contiguous() and out.shape[1] == EMBEDDING_DIM
            assert rows is None or (rows.dtype == torch.int32 and rows.numel() == nodes)
            slot.node_features, slot.out = _ptr(x), _ptr(out)
            slot.edge_index = _ptr(ei) if edges else None
            slot.edge_types = _ptr(et) if edges else None
            slot.out_rows = _ptr(rows)
            slot.n_nodes, slot.n_edges = nodes, edges
            records = records_of(ei)
            if records is not None and edges:
                node_ptr, edge_ptr = records
                slot.node_ptr, slot.edge_ptr = _ptr(node_ptr), _ptr(edge_ptr)
                slot.n_records = int(node_ptr.numel()) - 1
        return array

    def prepare_batch_step(self, shards, *, normalise: bool = True):
        """``gfy_encode_coo_batch`` over device-resident shards as a pre-bound callable
        ``step(stream_handle)``: one C-ABI call per batch, descriptors, workspace (cleared
        here, once) and pointers resolved up front.  ``shards`` as for ``_shard_array``; all
        outputs share one dtype."""
        array = self._shard_array(shards)
        count = len(shards)
        out_code = _GFY_OF_TORCH[shards[0][4].dtype]
        assert all(s[4].dtype == shards[0][4].dtype for s in shards)
        lib, handle = self._lib, self._handle
        with torch.cuda.device(self.device):
            need = lib.gfy_encode_coo_batch_workspace_bytes(handle, array, count)
            if need == 0:
                native.check(native.GFY_ERR_INVALID, "gfy_encode_coo_batch_workspace_bytes")
            scratch = torch.zeros(need, dtype=torch.uint8, device=self.device)
            torch.cuda.current_stream(self.device).synchronize()      # see prepare_step
        encode, p_ws, ws_bytes = lib.gfy_encode_coo_batch, _ptr(scratch), scratch.numel()
        flag = 1 if normalise else 0
        keep = (shards, array, scratch)

        def step(stream_handle: int, _keep=keep) -> None:
            status = encode(handle, array, count, out_code, flag, p_ws, ws_bytes, stream_handle)
            if status != 0:
                native.check(status, "gfy_encode_coo_batch")
        return step

    def encode_coo_batch(self, shards, *, out_dtype: torch.dtype = torch.float16,
                         normalise: bool = True) -> list[torch.Tensor]:
        """``shards``: sequence of (node_features, edge_index, edge_types, out_rows or None,
        n_out or None) -> one [n_out, 128] device tensor per shard, from ONE sequence of
        launches on the current stream (a fresh workspace per call: steady-state loops use
        ``prepare_batch_step``)."""
        with torch.cuda.device(self.device):
            outs = [torch.empty((int(x.shape[0]) if kept is None else int(kept), EMBEDDING_DIM),
                                dtype=out_dtype, device=self.device)
                    for x, _ei, _et, _rows, kept in shards]
            step = self.prepare_batch_step(
                [(x, ei, et, rows, out) for (x, ei, et, rows, _k), out in zip(shards, outs)],
                normalise=normalise)
            step(self._stream())
        return outs

    def encode_coo_group(self, shards, *, normalise: bool = True) -> None:
        """Up to 16 micro-batches in ONE sequence of launches (``gfy_encode_coo_batch``) on the
        current stream and on the encoder's own workspace: what ``encode_graphs`` issues for a
        group of consecutive micro-batches (reference: the micro-batch loop api.py:211-230).
        ``shards``: (node_features, edge_index, edge_types, out_rows or None, out) device
        tensors, every ``out`` preallocated ([n_out, 128], one dtype)."""
        array = self._shard_array(shards)
        count = len(shards)
        out_code = _GFY_OF_TORCH[shards[0][4].dtype]
        assert all(s[4].dtype == shards[0][4].dtype for s in shards)
        lib = self._lib
        with torch.cuda.device(self.device):
            need = lib.gfy_encode_coo_batch_workspace_bytes(self._handle, array, count)
            if need == 0:
                native.check(native.GFY_ERR_INVALID, "gfy_encode_coo_batch_workspace_bytes")
            rows = sum(-(-int(x.shape[0]) // 32) * 32 for x, *_rest in shards)
            stream = torch.cuda.current_stream(self.device)
            scratch = self._coo_scratch(need, rows, stream)
            native.check(lib.gfy_encode_coo_batch(
                self._handle, array, count, out_code, 1 if normalise else 0, _ptr(scratch),
                scratch.numel(), stream.cuda_stream), "gfy_encode_coo_batch")
            self._coo_enqueued(rows, stream)

    # -- the same from device ADDRESSES (no tensor per array: encode_shards_device) -------------
    @staticmethod
    def pointer_shard(pointers, *, nodes: int, edges: int, records: int, out: torch.Tensor,
                      keep=None) -> tuple:
        """One micro-batch for ``encode_coo_group_pointers``: ``pointers`` = device addresses of
        (node_features f32 [nodes][7], edge_index i32 [2][edges], edge_types u8 [edges], out_rows
        i32 [nodes] or 0, node_ptr i64 [records + 1] or 0, edge_ptr or 0) — one upload block cut
        by offsets (``api._Uploader.send_block``); ``keep``: what owns that memory."""
        assert out.is_contiguous() and out.shape[1] == EMBEDDING_DIM
        return (tuple(pointers), nodes, edges, records, out, keep)

    def encode_coo_group_pointers(self, members, *, normalise: bool = True) -> None:
        """``encode_coo_group`` for micro-batches described by addresses (``pointer_shard``)."""
        if not 1 <= len(members) <= native.GFY_MAX_BATCH_SHARDS:
            raise ValueError(f"a batch holds 1..{native.GFY_MAX_BATCH_SHARDS} shards, "
                             f"got {len(members)}")
        count = len(members)
        array = (native.GfyShard * count)()
        rows = 0
        for slot, (pointers, nodes, edges, records, out, _keep) in zip(array, members):
            p_x, p_ei, p_et, p_rows, p_np, p_ep = pointers
            slot.node_features, slot.out = p_x, out.data_ptr()
            slot.edge_index = p_ei if edges else None
            slot.edge_types = p_et if edges else None
            slot.out_rows = p_rows or None
            slot.n_nodes, slot.n_edges = nodes, edges
            if records > 0 and p_np and p_ep and edges:
                slot.node_ptr, slot.edge_ptr, slot.n_records = p_np, p_ep, records
            rows += -(-nodes // 32) * 32
        out_code = _GFY_OF_TORCH[members[0][4].dtype]
        lib = self._lib
        with torch.cuda.device(self.device):
            need = lib.gfy_encode_coo_batch_workspace_bytes(self._handle, array, count)
            if need == 0:
                native.check(native.GFY_ERR_INVALID, "gfy_encode_coo_batch_workspace_bytes")
            stream = torch.cuda.current_stream(self.device)
            scratch = self._coo_scratch(need, rows, stream)
            native.check(lib.gfy_encode_coo_batch(
                self._handle, array, count, out_code, 1 if normalise else 0, _ptr(scratch),
                scratch.numel(), stream.cuda_stream), "gfy_encode_coo_batch")
            self._coo_enqueued(rows, stream)

    def hidden(self, node_features: torch.Tensor, csr: DeviceCsr,
               stage: int) -> torch.Tensor:
        """Parity tap: hidden state after ``stage`` (0 = input Linear,
        l+1 = after layer l), [N,128] in the model dtype."""
        with torch.cuda.device(self.device):
            out = torch.empty((csr.nodes, EMBEDDING_DIM), dtype=self.model_dtype,
                              device=self.device)
            need = self._lib.gfy_encode_workspace_bytes(
                self._handle, csr.nodes, csr.edges)
            scratch = self._scratch(need)
            native.check(self._lib.gfy_encode_hidden(
                self._handle, _ptr(node_features), _ptr(csr.row_ptr),
                _ptr(csr.col), _ptr(csr.typ), csr.nodes, csr.edges, stage,
                _ptr(out), _ptr(scratch), scratch.numel(), self._stream()),
                "gfy_encode_hidden")
        return out

    def debug_layer(self, hidden: torch.Tensor, csr: DeviceCsr, layer: int,
                    tap: int = native.GFY_TAP_H) -> torch.Tensor:
        """Parity tap (``gfy_debug_layer``): GINE layer ``layer`` run on the given hidden
        state (fp16 [N,128], natural channel order — e.g. the reference's own recorded
        tensor), one of its phase tensors back: ``native.GFY_TAP_H`` h', ``_Z`` z, ``_V`` v
        ([N,256]), ``_W`` w, ``_Y`` y."""
        assert hidden.dtype == torch.float16 and hidden.is_contiguous()
        assert tuple(hidden.shape) == (csr.nodes, EMBEDDING_DIM)
        width = 2 * EMBEDDING_DIM if tap == native.GFY_TAP_V else EMBEDDING_DIM
        with torch.cuda.device(self.device):
            out = torch.empty((csr.nodes, width), dtype=torch.float16, device=self.device)
            need = self._lib.gfy_debug_layer_workspace_bytes(self._handle, csr.nodes, csr.edges)
            scratch = self._scratch(need)
            native.check(self._lib.gfy_debug_layer(
                self._handle, int(layer), _ptr(hidden), _ptr(csr.row_ptr), _ptr(csr.col),
                _ptr(csr.typ), csr.nodes, csr.edges, int(tap), _ptr(out), _ptr(scratch),
                scratch.numel(), self._stream()), "gfy_debug_layer")
        return out

    # -- diagnostics -------------------------------------------------------------------
    def set_option(self, option: int, value: int) -> None:
        """Diagnostic switches of include/gfy.h (``native.GFY_OPT_*``): head fused into the
        last layer launch or not."""
        native.check(self._lib.gfy_encoder_set_option(
            self._handle, int(option), int(value)), "gfy_encoder_set_option")
        self._options[int(option)] = int(value)

    def twin(self) -> "DeviceEncoder":
        """A second gfy_encoder of the same weights on the same device, with this one's
        options: a call in flight owns its encoder's hidden-state buffers and workspace, so two
        groups of micro-batches in flight (``Ginfinity.encode_staged``: what ``bench.py`` times)
        need two."""
        other = DeviceEncoder(self._weight_pack, full_precision=self.full_precision,
                              device=self.device)
        for option, value in self._options.items():
            other.set_option(option, value)
        return other

    def last_layer_kernel(self) -> int:
        """Layer kernel of the last fp16-model encode (``native.GFY_OPT_LAYER_KERNEL`` values:
        1 one round per launch, 3 persistent rounds, 4 windowed rounds; 0: none yet)."""
        return int(self._lib.gfy_encoder_last_layer_kernel(self._handle))

    def set_timing(self, enabled: bool | int) -> None:
        """True / 1: an event after every launch; 2: none between layer launches
        1 .. L-1, whose mean is reported (agrees with rocprof's kernel durations for one
        stream); 3: no events, the layer launches time themselves on the device clock —
        ``kernel_times_ms`` then returns one duration per layer launch, valid with any number
        of streams in flight."""
        native.check(self._lib.gfy_encoder_set_timing(
            self._handle, int(enabled)), "gfy_encoder_set_timing")

    def kernel_times_ms(self) -> list[float]:
        """Device time between the marks of the last ``encode`` (timing enabled):
        [setup (tile plans + input Linear), layer 1 .. layer L, stand-alone head].
        For fp16 output the last layer's launch runs the head + normalise as well
        and the final entry is ~0."""
        buffer = (ctypes.c_float * 16)()
        count = ctypes.c_int()
        native.check(self._lib.gfy_encoder_get_timing(
            self._handle, buffer, 16, ctypes.byref(count)),
            "gfy_encoder_get_timing")
        return [float(buffer[i]) for i in range(count.value)]

    # -- one micro-batch, host arrays in → device embeddings out ----------------------
    def _upload(self, array: np.ndarray) -> torch.Tensor:
        """Host array -> device tensor; read-only arrays (a memory-mapped shard) are only read."""
        with warnings.catch_warnings():
            warnings.filterwarnings("ignore", message="The given NumPy array is not writable")
            return torch.from_numpy(np.ascontiguousarray(array)).to(self.device)

    def upload_arrays(self, node_features: np.ndarray, edge_index: np.ndarray,
                      edge_types: np.ndarray, node_roles: np.ndarray | None, *,
                      node_ptr: np.ndarray | None = None,
                      edge_ptr: np.ndarray | None = None) -> tuple:
        """Host arrays of one micro-batch → ``(features, edge_index, edge_types, out_rows or
        None, kept)`` on the device: context nodes (role != 0) take part in message passing
        and are dropped at the head's store through ``out_rows`` (api.py:253-260).
        ``node_ptr`` / ``edge_ptr``: the micro-batch's record boundaries (graph.py:268-271);
        where they pay (``records_pay``) they go up too and ride on the edge_index tensor
        (``attach_records``): the batch call then builds its tile plans without global atomics
        (include/gfy.h, gfy_shard.node_ptr)."""
        nodes = int(node_features.shape[0])
        out_rows, kept = None, nodes
        if node_roles is not None:
            core = node_roles == 0
            kept = int(np.count_nonzero(core))
            if kept != nodes:
                rows = np.cumsum(core, dtype=np.int32) - np.int32(1)
                rows[~core] = -1
                out_rows = torch.from_numpy(rows).to(self.device)
        x, ei, et = (self._upload(a) for a in (node_features, edge_index, edge_types))
        if node_ptr is not None and edge_ptr is not None and records_pay(node_ptr, edge_ptr):
            attach_records(ei, self._upload(np.asarray(node_ptr, dtype=np.int64)),
                           self._upload(np.asarray(edge_ptr, dtype=np.int64)))
        return x, ei, et, out_rows, kept

    def encode_arrays(self, node_features: np.ndarray, edge_index: np.ndarray,
                      edge_types: np.ndarray, node_roles: np.ndarray | None,
                      *, out_dtype: torch.dtype = torch.float16,
                      normalise: bool = True,
                      out: torch.Tensor | None = None) -> torch.Tensor:
        """Host arrays of one shard slice → [core_nodes, 128] device tensor.
        Context nodes (role != 0) take part in message passing and are dropped
        in the head kernel's store (api.py:253-260)."""
        x, ei, et, out_rows, n_out = self.upload_arrays(node_features, edge_index, edge_types,
                                                        node_roles)
        return self.encode_coo(x, ei, et, out_rows=out_rows, n_out=n_out,
                               out_dtype=out_dtype, normalise=normalise, out=out)
