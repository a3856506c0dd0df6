"""Public embedding API — the drop-in for ``ginfinity.Ginfinity``.

Same method names, keyword arguments, return types and error behaviour as the
reference (src/ginfinity/api.py:53-260); the compute under
``_run_graph_shard`` is the HIP path of libgfy instead of a ``torch.nn``
module.  Differences a user can observe, all deliberate:

* ``device="cuda"``/``"cuda:i"`` is the HIP device under PyTorch-ROCm, i.e. an
  MI355X; it needs ``allow_nondeterministic_cuda=True`` exactly as the
  reference's CUDA path does (api.py:69-76) although these kernels are
  deterministic (no atomics in the float path);
* ``device="cpu"`` (the default, as in the reference) runs the host
  implementation inside libgfy (csrc/gine_host.cpp: the same rounding-point
  model in plain C++) — for boxes without a GPU and small inputs; an encoder
  loaded for ``"cuda"`` never routes there, and nothing comes from ``oracle/``;
* the per-record arrays returned by one call are views into one host block
  (C-contiguous, independent rows) instead of separate allocations, unless
  ``encoder.independent_outputs = True``;
* on the GPU that host block is page-locked memory from torch's caching host allocator and the
  embeddings are DMA'd straight into it — no staging copy on the host, the call runs at PCIe
  speed (``load(..., pinned_outputs=...)``, not in the reference: ``None`` = this default,
  ``False`` = pageable memory through a staging ring, ``True`` = always).  The arrays are
  ordinary numpy views; the block returns to the allocator's cache when the last of them is
  dropped, and while results worth more than ``PINNED_RESULT_LIMIT`` bytes are alive the default
  falls back to pageable memory (page-locked memory cannot be swapped).
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
import collections
import ctypes
import os
import queue
import threading

import json
from pathlib import Path
from typing import Sequence

import numpy as np
import torch

from . import _native as native
from .engine import DeviceEncoder, attach_records, device_output_dtype, records_pay
from .host import HostEncoder
from .graph import Graph, GraphBuilder, GraphShard, shard_text
from .records import RNA
from .spec import (DATA_DIRECTORY, GraphCompatibilityError, GraphSpec,
                   GraphValidationError)
from .weights import LoadedCheckpoint, ModelIntegrityError, load_checkpoint


def _embedding_dtype(value: np.dtype | str) -> np.dtype:
    try:
        dtype = np.dtype(value)
    except TypeError as error:
        raise ValueError(f"unsupported embedding dtype {value!r}") from error
    if dtype.kind != "f":
        raise ValueError(f"embedding dtype must be floating-point, got {dtype}")
    return dtype


def default_alignment_parameters() -> dict[str, float]:
    """Model-versioned scoring parameters for the separate aligner package
    (passthrough; reference api.py:47-50)."""
    try:
        data = json.loads((DATA_DIRECTORY / "alignment.json").read_text())
    except (OSError, json.JSONDecodeError) as error:
        raise ModelIntegrityError(
            f"cannot read model metadata: {error}") from error
    return dict(data["scoring_parameters"])


def microbatch_bounds(lengths: Sequence[int], edge_counts: Sequence[int],
                      max_batch_nodes: int, max_batch_edges: int
                      ) -> list[tuple[int, int]]:
    """Greedy contiguous packing of records into micro-batches, as
    ``encode_graphs`` does it (api.py:211-230): a record joins the current
    batch unless that would exceed either limit; a batch always holds at least
    one record.  Same boundaries as the reference's record-by-record loop, found
    per micro-batch by two binary searches over the running sums (the loop over
    5,840 records was 1 ms of every call)."""
    total = len(lengths)
    if total == 0:
        return []
    nodes = np.zeros(total + 1, dtype=np.int64)
    edges = np.zeros(total + 1, dtype=np.int64)
    np.cumsum(lengths, out=nodes[1:])           # (ndarrays welcome: a 5,840-tuple costs
    np.cumsum(edge_counts, out=edges[1:])       #  0.15 ms to convert)
    bounds: list[tuple[int, int]] = []
    start = 0
    while start < total:
        # the largest stop with nodes[stop] - nodes[start] <= limit (and the same for edges)
        by_nodes = int(np.searchsorted(nodes, nodes[start] + max_batch_nodes, side="right")) - 1
        by_edges = int(np.searchsorted(edges, edges[start] + max_batch_edges, side="right")) - 1
        stop = max(start + 1, min(by_nodes, by_edges, total))
        bounds.append((start, stop))
        start = stop
    return bounds


def _advise_huge_pages(block: np.ndarray) -> None:
    """Ask for transparent huge pages under a large, freshly allocated result block: its first
    touch (by the copier threads) then takes a fault per 2 MB instead of one per 4 KB — 56,000
    faults, 11 ms on one thread, for the 230 MB of config 2.  Best effort: not Linux, THP off
    or a small block → nothing happens."""
    try:
        import ctypes
        libc = ctypes.CDLL(None, use_errno=True)
        huge = 2 << 20
        begin = block.ctypes.data
        first = -(-begin // huge) * huge
        last = (begin + block.nbytes) // huge * huge
        if last > first:
            libc.madvise(ctypes.c_void_p(first), ctypes.c_size_t(last - first), 14)  # MADV_HUGEPAGE
    except Exception:       # pragma: no cover - advisory only
        pass


#: consecutive micro-batches of a call that share one sequence of launches
#: (``gfy_encode_coo_batch``, at most 16): four 60,000-node micro-batches give every CU 3.7
#: rounds of tiles, and a group's embeddings (61 MB) come back as one copy
MICROBATCH_GROUP = 4


#: ``encode_shards_device`` packs its micro-batches with ``gfy_pack_microbatch`` (False: the numpy
#: form, which the tests compare it with)
NATIVE_PACKER = True


def _packable(shard: GraphShard) -> bool:
    """The arrays ``gfy_pack_microbatch`` reads by address: the dtypes of the interchange format,
    C-contiguous (what ``GraphShard`` holds unless a caller built it from views)."""
    wanted = ((shard.node_features, np.float32), (shard.edge_index, np.int32),
              (shard.edge_types, np.uint8), (shard.node_roles, np.uint8),
              (shard.node_ptr, np.int64), (shard.edge_ptr, np.int64))
    return all(isinstance(a, np.ndarray) and a.dtype == d and a.flags.c_contiguous
               for a, d in wanted)


#: groups of micro-batches ``encode_staged`` keeps in flight (2: as ``bench.py --streams 2``)
STAGED_LANES = 2
#: ... and ``encode_shards_device`` (host arrays in): 1 — the call is bound by the uploads, a
#: second lane costs 2–4 % there (128 shards: 12.8–13.0 against 13.1–13.7 ms on one box,
#: tools/bench_host_feed.py --lanes)
HOST_FEED_LANES = 1


def _groups(count: int, size: int | None = None, *, ramp: bool = False) -> list[range]:
    """Consecutive micro-batches per launch sequence.  ``ramp``: the first groups hold 1, 1, 2
    micro-batches — a call that returns host arrays is bound by the copy engine (230 MB at
    49.6 GB/s for config 2), which has nothing to do until the first group's embeddings exist."""
    size = MICROBATCH_GROUP if size is None else size
    groups, first, opening = [], 0, [1, 1, 2] if ramp and size > 2 else []
    while first < count:
        take = min(opening.pop(0) if opening else size, size, count - first)
        groups.append(range(first, first + take))
        first += take
    return groups


def _settle(jobs) -> None:
    """A call is leaving through an exception: no helper job of it may run (and write into a
    staging slot the next call could be handed) after that."""
    for job in jobs:
        job.cancel()
    for job in jobs:
        if not job.cancelled():
            try:
                job.exception()
            except BaseException:       # pragma: no cover - cancelled meanwhile
                pass


#: packer threads of an encoder (numpy releases the GIL in the copies they make); one per
#: staging slot of the uploader's ring at most
_PACKERS = int(os.environ.get("GFY_PACKERS", "8"))
#: staging slots of an uploader's ring (micro-batches packed ahead of the launching thread)
_STAGING_SLOTS = int(os.environ.get("GFY_STAGING_SLOTS", "8"))

#: default mode only: results alive beyond this many page-locked bytes go to pageable memory
PINNED_RESULT_LIMIT = 4 << 30
_pinned_alive = [0]                     # bytes of page-locked result blocks not yet dropped
_pinned_lock = threading.Lock()         # (finalizers run on whatever thread drops a block)


def _pinned_block(shape, dtype: torch.dtype) -> torch.Tensor:
    """A page-locked host tensor whose bytes count against PINNED_RESULT_LIMIT while it lives
    (the numpy views handed to the caller keep it alive)."""
    import weakref
    block = torch.empty(shape, dtype=dtype, pin_memory=True)
    size = block.numel() * block.element_size()
    with _pinned_lock:
        _pinned_alive[0] += size
    weakref.finalize(block, _pinned_released, size)
    return block


def _pinned_released(size: int) -> None:
    with _pinned_lock:
        _pinned_alive[0] -= size


class _ParameterView:
    """``parameters()`` / ``state_dict()`` of the loaded checkpoint as torch tensors."""

    def __init__(self, state: dict, dtype: torch.dtype) -> None:
        self._state, self._dtype = state, dtype

    def state_dict(self) -> dict:
        return {name: (torch.from_numpy(np.array(value)).to(self._dtype)
                       if np.issubdtype(value.dtype, np.floating)
                       else torch.from_numpy(np.array(value)))
                for name, value in self._state.items()}

    def parameters(self):
        for name, value in self.state_dict().items():
            if "running_" not in name and "num_batches_tracked" not in name:
                yield value


class _Downloader:
    """Device block → the caller's (pageable) host memory through a ring of pinned staging
    buffers: the DMA runs at PCIe speed into pinned memory and the copy out of it is a plain
    memcpy (numpy releases the GIL for it) — six worker threads, each with its own stream, so
    several memcpys and DMAs overlap (one pageable ``tensor.cpu()`` is staged by a single
    runtime thread, ≈14 GB/s; two workers reached 17–19 GB/s, profiles/r02_api_bench.json).
    The destination is a row range of ONE host block per ``encode_graphs`` call, first touched
    (page-faulted) by the workers themselves, in parallel.  The caller gets ordinary numpy
    memory: nothing stays pinned on its behalf."""

    def __init__(self, device: torch.device, *, threads: int = 6, slots: int = 8) -> None:
        self._device = device
        self._free: "queue.SimpleQueue[torch.Tensor | None]" = queue.SimpleQueue()
        for _ in range(slots):
            self._free.put(None)                      # allocated at first use, sized then
        self._pool = ThreadPoolExecutor(
            max_workers=threads, thread_name_prefix="ginfinity-d2h",
            initializer=self._enter_stream)

    def _enter_stream(self) -> None:                  # torch's current stream is thread-local
        torch.cuda.set_device(self._device)
        torch.cuda.set_stream(torch.cuda.Stream(device=self._device))

    def submit(self, block: torch.Tensor, ready: "torch.cuda.Event", finish,
               destination: np.ndarray | None = None):
        """Future of ``finish(host_array)``; ``block`` is read once ``ready`` has happened.
        ``destination``: where the rows go (same shape and dtype as ``block``); a fresh array
        otherwise."""
        return self._pool.submit(self._run, block, ready, finish, destination)

    def _run(self, block: torch.Tensor, ready: "torch.cuda.Event", finish, destination):
        nbytes = block.numel() * block.element_size()
        staging = self._free.get()
        try:
            if staging is None or staging.numel() < nbytes:
                staging = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8,
                                      pin_memory=True)
            stream = torch.cuda.current_stream(self._device)
            ready.synchronize()     # not stream.wait_event: see _DirectDownloader (a copy queued
            #                         behind a pending cross-stream dependency runs at 1/4 rate)
            view = staging[:nbytes].view(block.dtype).view(block.shape)
            view.copy_(block, non_blocking=True)
            stream.synchronize()
            host = destination
            if host is None:
                host = np.empty(tuple(block.shape), dtype=view.numpy().dtype)
            np.copyto(host, view.numpy())
        finally:
            self._free.put(staging)
        return finish(host)


class _DirectDownloader:
    """Device block → a row range of a PINNED host block, one DMA per micro-batch on one copy
    stream, no host memcpy, no worker thread.  A copy is ENQUEUED only once the kernels that
    produce its rows have finished: a copy enqueued behind a still-pending cross-stream
    dependency was, on most boxes, served at a quarter of the rate (15.4 MB in 1.23 instead of
    0.31 ms, one to four such copies per call; profiles/README.md, "D2H").  The launching thread
    therefore keeps the copies of unfinished micro-batches in a list and enqueues those whose
    event has happened whenever it passes by (``submit``, ``result``); a helper thread doing
    the same cost the packers and the launcher 1 ms of a call."""

    class _Landed:
        def __init__(self, owner: "_DirectDownloader", block, ready, destination) -> None:
            self._owner = owner
            self._job = (block, ready, destination)
            self._done: "torch.cuda.Event | None" = None

        def result(self) -> None:
            self._owner._pump(until=self)
            if self._done is None:
                raise RuntimeError("this copy was abandoned (the call that queued it raised)")
            self._done.synchronize()

    def __init__(self, device: torch.device) -> None:
        self._stream = torch.cuda.Stream(device=device)
        self._pending: "collections.deque[_DirectDownloader._Landed]" = collections.deque()

    def _pump(self, until: "_DirectDownloader._Landed | None" = None) -> None:
        """Enqueue, in order, every pending copy whose rows are ready; with ``until``, wait for
        the events in front of (and of) that copy."""
        pending = self._pending
        while pending:
            head = pending[0]
            block, ready, destination = head._job
            if not ready.query():
                if until is None or until._done is not None:
                    return
                ready.synchronize()
            with torch.cuda.stream(self._stream):
                destination.copy_(block, non_blocking=True)
                done = torch.cuda.Event()
                done.record(self._stream)
            head._done, head._job = done, None
            pending.popleft()

    def submit(self, block: torch.Tensor, ready: "torch.cuda.Event",
               destination: torch.Tensor) -> "_DirectDownloader._Landed":
        # (no record_stream: `block` is a row range of the encoder's own device block, which is
        # next written by a later call — after this call's copies have landed)
        landed = self._Landed(self, block, ready, destination)
        self._pending.append(landed)
        self._pump()
        return landed

    def abandon(self) -> None:
        """Forget the copies not yet enqueued (a call that raised)."""
        self._pending.clear()


class _Uploader:
    """Several small host arrays → device tensors with ONE copy: they are packed into a
    pinned staging buffer (ring, each slot guarded by the event of its last copy) and go up
    as a single asynchronous H2D; the device tensors are views of one allocation.  Five
    pageable ``tensor.to(device)`` per micro-batch were 0.4 ms of the launching thread.

    ``pack`` (host memcpy into pinned memory, edge rebasing, the edge range check of the
    slice) may run on a worker thread ahead of the launching thread, which then only calls
    ``send``: with 61 MB of inputs for the 897,588-node shard the packing was 4 of the
    launching thread's 12 ms."""

    _TORCH = {np.dtype(np.uint8): torch.uint8, np.dtype(np.int64): torch.int64,
              np.dtype(np.int32): torch.int32, np.dtype(np.float32): torch.float32}

    def __init__(self, device: torch.device, slots: int | None = None) -> None:
        slots = _STAGING_SLOTS if slots is None else slots
        self._device = device
        self._staging: list[torch.Tensor | None] = [None] * slots
        self._copied: list["torch.cuda.Event | None"] = [None] * slots
        self._next = 0
        self.slots = slots
        self._ring: int | None = None        # gfy_upload_ring (send_group), made on first use

    def __del__(self) -> None:              # pragma: no cover - interpreter teardown order
        try:
            if self._ring:
                native.library().gfy_upload_ring_destroy(self._ring)
        except Exception:
            pass

    def reserve(self) -> int:
        """Next staging slot (called by the launching thread, in micro-batch order)."""
        slot = self._next
        self._next = (slot + 1) % len(self._staging)
        return slot

    def pack(self, slot: int, arrays: Sequence):
        """``arrays``: numpy arrays, ``None``, or ``(array, scalar, lo, hi)`` = pack
        ``array - scalar`` after checking that every value lies in [lo, hi) (edge indices
        rebased to the micro-batch's first node, graph.py:414-444, with the range check of
        graph.py:318-321 on the slice).  Thread-safe per slot."""
        rebases = [item[1:] if isinstance(item, tuple) else None for item in arrays]
        arrays = [item[0] if isinstance(item, tuple) else item for item in arrays]
        offsets, total = [], 0
        for array in arrays:
            offsets.append(total)
            if array is not None:
                total += -(-array.nbytes // 256) * 256
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()          # the slot's last copy has left it
        staging = self._staging[slot]
        if staging is None or staging.numel() < total:
            staging = self._staging[slot] = torch.empty(
                max(total, 1 << 20), dtype=torch.uint8, pin_memory=True)
        host = staging.numpy()
        for array, offset, rebase in zip(arrays, offsets, rebases):
            if array is not None and array.nbytes:
                # ONE pass from the source (a view of the shard — for a loaded shard a view of
                # the file mapping) into pinned memory, rebasing on the way where asked
                target = host[offset:offset + array.nbytes].view(array.dtype).reshape(array.shape)
                if rebase is None:
                    np.copyto(target, array)
                else:
                    base, low, high = rebase
                    if base:
                        np.subtract(array, base, out=target)
                    else:                      # a shard's first micro-batch: nothing to rebase
                        np.copyto(target, array)
                    # one pass: as unsigned, a value below `low` wraps to ≥ 2^31 > high - low
                    if target.size and int(target.view(np.uint32).max()) >= high - low:
                        raise GraphValidationError("edge index outside shard node range")
        return slot, arrays, offsets, total

    def send(self, packed, *, mapped: bool = False) -> list[torch.Tensor | None]:
        """The packed slot → device tensors (views of one allocation), on the current stream.
        ``mapped``: no copy at all — the tensors returned are the page-locked staging memory
        itself, which the device reads over PCIe (hipHostMalloc memory is mapped into the
        device's address space; each input array is read once, by the counting and the set-up
        kernel).  The caller then says with ``hold`` which event ends those reads.  Used with
        page-locked results: an H2D copy per micro-batch can land on the copy engine that is
        busy bringing the embeddings back, and the kernels then wait 0.3 ms per micro-batch for
        4 MB of input (profiles/README.md, "D2H": 7.3-7.7 ms per call instead of 6.2-6.6)."""
        slot, arrays, offsets, total = packed
        staging = self._staging[slot]
        if mapped:
            on_device = staging
        else:
            on_device = torch.empty(max(total, 1), dtype=torch.uint8, device=self._device)
            on_device[:total].copy_(staging[:total], non_blocking=True)
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self._device))
            self._copied[slot] = done
        views: list[torch.Tensor | None] = []
        for array, offset in zip(arrays, offsets):
            if array is None:
                views.append(None)
                continue
            flat = on_device[offset:offset + array.nbytes].view(self._TORCH[array.dtype])
            views.append(flat.view(array.shape))
        return views

    # -- a GROUP of micro-batches in one slot: one H2D copy, one event per group -----------------
    @staticmethod
    def padded(nbytes: int) -> int:
        return -(-nbytes // 256) * 256

    def prepare_slot(self, slot: int, total: int) -> None:
        """Launching thread, before the group's packers are submitted: the slot's last copy has
        left it and its staging buffer holds ``total`` bytes."""
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()
        if self._ring:
            native.check(native.library().gfy_upload_wait(self._ring, slot), "gfy_upload_wait")
        staging = self._staging[slot]
        if staging is None or staging.numel() < total:
            self._staging[slot] = torch.empty(max(total, 1 << 20), dtype=torch.uint8,
                                              pin_memory=True)

    def pack_at(self, slot: int, base: int, arrays: Sequence) -> list[int]:
        """``pack`` into a prepared slot at byte ``base`` (packer threads; the ranges of a
        group's micro-batches are disjoint): the offsets of the arrays inside the slot."""
        rebases = [item[1:] if isinstance(item, tuple) else None for item in arrays]
        arrays = [item[0] if isinstance(item, tuple) else item for item in arrays]
        host = self._staging[slot].numpy()
        offsets, at = [], base
        for array, rebase in zip(arrays, rebases):
            offsets.append(at)
            if array is None:
                continue
            if array.nbytes:
                target = host[at:at + array.nbytes].view(array.dtype).reshape(array.shape)
                if rebase is None:
                    np.copyto(target, array)
                else:
                    shift, low, high = rebase
                    if shift:
                        np.subtract(array, shift, out=target)
                    else:
                        np.copyto(target, array)
                    if target.size and int(target.view(np.uint32).max()) >= high - low:
                        raise GraphValidationError("edge index outside shard node range")
            at += self.padded(array.nbytes)
        return offsets

    def send_range(self, slot: int, total: int) -> torch.Tensor:
        """Bytes [0, total) of a packed slot → ONE device allocation by one asynchronous copy on
        the current stream."""
        block = torch.empty(max(total, 1), dtype=torch.uint8, device=self._device)
        block[:total].copy_(self._staging[slot][:total], non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self._device))
        self._copied[slot] = done
        return block

    def send_group(self, slot: int, total: int, copies: "torch.cuda.Stream",
                   consumer: "torch.cuda.Stream") -> torch.Tensor:
        """``send_range`` on ``copies`` with ``consumer`` made to wait for it, in ONE native call
        (``gfy_upload_async``: copy, event, stream wait): per group that was a ``copy_`` (89 us
        of the calling thread), two event records, an event wait and a stream switch."""
        lib = native.library()
        if self._ring is None:
            ring = ctypes.c_void_p()
            with torch.cuda.device(self._device):
                native.check(lib.gfy_upload_ring_create(self.slots, ctypes.byref(ring)),
                             "gfy_upload_ring_create")
            self._ring = ring.value
        with torch.cuda.stream(copies):           # the block belongs to the copy stream's pool
            block = torch.empty(max(total, 1), dtype=torch.uint8, device=self._device)
        block.record_stream(consumer)
        native.check(lib.gfy_upload_async(self._ring, slot, block.data_ptr(),
                                          self._staging[slot].data_ptr(), total,
                                          copies.cuda_stream, consumer.cuda_stream),
                     "gfy_upload_async")
        return block

    def hold(self, packed, event: "torch.cuda.Event") -> None:
        """The slot of ``packed`` (sent ``mapped``) is read by the device until ``event``."""
        self._copied[packed[0]] = event

    def __call__(self, arrays: Sequence) -> list[torch.Tensor | None]:
        return self.send(self.pack(self.reserve(), arrays))


class Ginfinity:
    """Loaded GINFINITY encoder, resident on one MI355X, ready for repeated
    inference."""

    def __init__(self, engine: "DeviceEncoder | None", checkpoint: LoadedCheckpoint,
                 device: str, *, full_precision: bool,
                 host: "HostEncoder | None" = None) -> None:
        self._engine = engine
        self._host = host
        self._copier: _Downloader | None = None
        self._direct: _DirectDownloader | None = None
        self._device_block: torch.Tensor | None = None
        self._preparer: ThreadPoolExecutor | None = None
        self._uploader: _Uploader | None = None
        self._copy_stream: "torch.cuda.Stream | None" = None
        self._lanes: "list[tuple[DeviceEncoder, torch.cuda.Stream]] | None" = None
        self._metadata = checkpoint.metadata
        self._state = checkpoint.state
        self._config = checkpoint.config
        self._graph_spec = checkpoint.graph_spec
        self.device = device
        self.full_precision = full_precision
        #: True: every returned per-record array owns its memory, as the reference's do
        #: (api.py:253-260: one tensor per record); False (default): the arrays of one
        #: micro-batch are row ranges of ONE host block (no per-record copy — keeping a single
        #: record alive keeps its block alive)
        self.independent_outputs = False
        #: None (default): the host block of a call is page-locked memory (torch's caching
        #: host allocator) and the device writes into it directly, unless more than
        #: PINNED_RESULT_LIMIT bytes of earlier results are still alive; True: always;
        #: False: pageable memory through the staging ring.  See the module docstring.
        self.pinned_outputs: bool | None = None

    @classmethod
    def load(cls, device: str = "cpu", *,
             allow_nondeterministic_cuda: bool = False,
             model_dir: str | Path | None = None,
             full_precision: bool = False,
             pinned_outputs: bool | None = None) -> "Ginfinity":
        """Same signature, defaults and device policy as the reference
        (src/ginfinity/api.py:64-76); ``pinned_outputs`` is this build's (module docstring)."""
        if not isinstance(device, str) or (device != "cpu" and not device.startswith("cuda")):
            raise ValueError("device must be 'cpu' or a CUDA device")
        if device.startswith("cuda"):
            if not allow_nondeterministic_cuda:
                raise ValueError("CUDA requires allow_nondeterministic_cuda=True")
            if not torch.cuda.is_available():
                raise ValueError("CUDA was requested but is unavailable")
        checkpoint = load_checkpoint(model_dir)
        if device == "cpu":
            host = HostEncoder(checkpoint.weight_pack, full_precision=full_precision)
            return cls(None, checkpoint, device, full_precision=full_precision, host=host)
        engine = DeviceEncoder(checkpoint.weight_pack,
                               full_precision=full_precision,
                               device=torch.device(device))
        loaded = cls(engine, checkpoint, device, full_precision=full_precision)
        loaded.pinned_outputs = None if pinned_outputs is None else bool(pinned_outputs)
        return loaded

    # -- metadata -----------------------------------------------------------------
    @property
    def _model(self) -> "_ParameterView":
        """What remains of the reference's ``torch.nn`` module here: its parameters, in the
        dtype the encoder computes with (api.py:110-113: ``model.half()`` unless
        ``full_precision``) — read-only copies; the compute uses the weight pack in libgfy."""
        return _ParameterView(self._state, torch.float32 if self.full_precision else torch.float16)

    @property
    def embedding_dimension(self) -> int:
        return self._config.out_dim

    @property
    def graph_spec(self) -> GraphSpec:
        """The graph contract accepted by this encoder."""
        return self._graph_spec

    def info(self) -> dict:
        return json.loads(json.dumps(self._metadata))

    # -- records in → embeddings out ---------------------------------------------------
    def encode(self, record: RNA, *, keep_paired_neighbours: bool = False,
               context_hops: int = 1,
               embedding_dtype: np.dtype | str = np.float16) -> np.ndarray:
        return self.encode_many(
            [record], keep_paired_neighbours=keep_paired_neighbours,
            context_hops=context_hops, embedding_dtype=embedding_dtype)[0]

    def encode_many(self, records: Sequence[RNA], *,
                    max_batch_nodes: int = 60_000,
                    max_batch_edges: int = 300_000,
                    keep_paired_neighbours: bool = False,
                    context_hops: int = 1,
                    embedding_dtype: np.dtype | str = np.float16
                    ) -> list[np.ndarray]:
        """Build graphs and encode them.  For sliced records the context
        nucleotides take part in message passing and are dropped afterwards:
        each returned array holds the core nucleotides only, 5'→3'."""
        records = list(records)
        if not records:
            return []
        if self._host is None and not any(record.sliced for record in records):
            return self._encode_records(records, max_batch_nodes, max_batch_edges,
                                        _embedding_dtype(embedding_dtype))
        shard = GraphBuilder(
            self._graph_spec, keep_paired_neighbours=keep_paired_neighbours,
            context_hops=context_hops).build_shard(records)
        return self.encode_graphs(
            shard, max_batch_nodes=max_batch_nodes,
            max_batch_edges=max_batch_edges, embedding_dtype=embedding_dtype)

    def _encode_records(self, records: Sequence[RNA], max_batch_nodes: int,
                        max_batch_edges: int, embedding_dtype: np.dtype
                        ) -> list[np.ndarray]:
        """Unsliced records: the graphs are built ON THE DEVICE (``gfy_build_graphs``,
        the arrays ``GraphBuilder`` would produce, bit for bit) — only the text, the
        record offsets and the positional columns cross PCIe, 2 + 8 bytes per nucleotide
        instead of 46.  Same packing, limits and errors as ``encode_graphs``."""
        text = shard_text(records, self._graph_spec)
        if max_batch_nodes <= 0 or max_batch_edges <= 0:
            raise ValueError("batch node and edge limits must be positive")
        lengths, edge_counts = text.lengths.tolist(), text.edge_counts.tolist()
        if max(lengths) > max_batch_nodes:
            raise ValueError("max_batch_nodes is smaller than the longest graph")
        if max(edge_counts) > max_batch_edges:
            raise ValueError("max_batch_edges is smaller than the largest graph")
        torch_dtype, _code, exact = device_output_dtype(embedding_dtype)
        engine, device = self._engine, self._engine.device
        spec = self._graph_spec
        if self._uploader is None:
            self._uploader = _Uploader(device)
        pending, verdicts = [], []
        bounds = microbatch_bounds(lengths, edge_counts, max_batch_nodes, max_batch_edges)
        produced = np.dtype(embedding_dtype) if exact else np.dtype(np.float64)
        total_rows = int(text.node_ptr[-1] - text.node_ptr[0])
        host_block, fetch, direct = self._landing(total_rows, produced, torch_dtype, exact)
        device_rows = self._device_rows(total_rows, torch_dtype)
        # the positional columns (numpy sin / cos, GIL released) of later micro-batches are
        # computed on a second helper thread while this one uploads and launches
        if self._preparer is None:
            self._preparer = ThreadPoolExecutor(max_workers=_PACKERS,
                                                thread_name_prefix="ginfinity-prep")
        columns_of = [self._preparer.submit(text.positional, a, b) for a, b in bounds]
        assert MICROBATCH_GROUP <= self._uploader.slots   # a group's inputs live in the ring
        try:
            for group in _groups(len(bounds), ramp=True):
                members, group_row = [], None
                for index in group:
                    start, stop = bounds[index]
                    n0, n1 = int(text.node_ptr[start]), int(text.node_ptr[stop])
                    e0, e1 = int(text.edge_ptr[start]), int(text.edge_ptr[stop])
                    columns = columns_of[index].result()
                    packed = self._uploader.pack(self._uploader.reserve(), (
                        text.bases[n0:n1], text.marks[n0:n1], text.node_ptr[start:stop + 1],
                        text.edge_ptr[start:stop + 1], columns))
                    bases, marks, node_ptr, edge_ptr, positional = self._uploader.send(
                        packed, mapped=direct)
                    features, edge_index, edge_types, first_invalid = engine.build_graphs(
                        bases, marks, node_ptr, edge_ptr, positional, n1 - n0, e1 - e0,
                        struct_states=1 if spec.struct_feature == "A" else 3,
                        skip2=spec.has_skip2)
                    row = n0 - int(text.node_ptr[0])
                    group_row = row if group_row is None else group_row
                    members.append((packed, (features, edge_index, edge_types, None,
                                             device_rows[row:row + n1 - n0])))
                    verdicts.append((start, first_invalid))
                    pending.append((row, n1 - n0, lengths[start:stop]))
                engine.encode_coo_group([tensors for _packed, tensors in members])
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(device))
                if direct:
                    for packed, _tensors in members:
                        self._uploader.hold(packed, ready)
                last_row, last_kept, _counts = pending[-1]
                landing = fetch(device_rows[group_row:last_row + last_kept], ready, group_row,
                                last_row + last_kept - group_row)
                for slot in range(len(pending) - len(members), len(pending)):
                    pending[slot] = (landing,) + pending[slot]
        except BaseException:
            _settle(columns_of)
            raise
        outputs: list[np.ndarray] = []
        for job, row, kept, counts in pending:     # views cut here, as the copies land
            job.result()
            outputs.extend(self._splitter(counts, embedding_dtype, exact)(
                host_block[row:row + kept]))
        for start, first_invalid in verdicts:
            bad = int(first_invalid.item())
            if bad >= 0:
                raise GraphValidationError(
                    f"record {records[start + bad].identifier!r}: sequence or structure "
                    "text is not a balanced dot-bracket string over A, C, G, U")
        return outputs

    def encode_graph(self, graph: Graph, *,
                     embedding_dtype: np.dtype | str = np.float16) -> np.ndarray:
        return self.encode_graphs([graph], embedding_dtype=embedding_dtype)[0]

    def _checked_shard(self, graphs: Sequence[Graph] | GraphShard,
                       max_batch_nodes: int, max_batch_edges: int
                       ) -> GraphShard | None:
        if isinstance(graphs, GraphShard):
            shard = graphs
        else:
            graphs = list(graphs)
            if not graphs:
                return None
            shard = GraphShard.from_graphs(graphs)
        if shard.spec.sha256 != self._graph_spec.sha256:
            raise GraphCompatibilityError(
                "graphs were built with a specification incompatible with "
                "this encoder")
        if max_batch_nodes <= 0 or max_batch_edges <= 0:
            raise ValueError("batch node and edge limits must be positive")
        # (a shard that fits a micro-batch as a whole has no record that does not: the per-record
        # maxima are looked at only above that — 7 us per shard of a 128-shard call's prologue)
        if (int(shard.node_ptr[-1]) > max_batch_nodes
                and int(np.diff(shard.node_ptr).max()) > max_batch_nodes):
            raise ValueError("max_batch_nodes is smaller than the longest graph")
        if (int(shard.edge_ptr[-1]) > max_batch_edges
                and int(np.diff(shard.edge_ptr).max()) > max_batch_edges):
            raise ValueError("max_batch_edges is smaller than the largest graph")
        return shard

    @staticmethod
    def _shard_bounds(shard: GraphShard, max_batch_nodes: int, max_batch_edges: int
                      ) -> list[tuple[int, int]]:
        """``microbatch_bounds`` of a shard; one that fits the limits as a whole is one
        micro-batch without the running sums (what the greedy packing gives for it)."""
        records = shard.record_count
        if (records and int(shard.node_ptr[-1]) <= max_batch_nodes
                and int(shard.edge_ptr[-1]) <= max_batch_edges):
            return [(0, records)]
        return microbatch_bounds(shard.lengths, shard.edge_counts, max_batch_nodes,
                                 max_batch_edges)

    def encode_graphs(self, graphs: Sequence[Graph] | GraphShard, *,
                      max_batch_nodes: int = 60_000,
                      max_batch_edges: int = 300_000,
                      embedding_dtype: np.dtype | str = np.float16
                      ) -> list[np.ndarray]:
        """Encode prebuilt graphs, micro-batching a persistent shard."""
        shard = self._checked_shard(graphs, max_batch_nodes, max_batch_edges)
        if shard is None:
            return []
        embedding_dtype = _embedding_dtype(embedding_dtype)
        bounds = microbatch_bounds(np.diff(shard.node_ptr), np.diff(shard.edge_ptr),
                                   max_batch_nodes, max_batch_edges)
        if len(bounds) == 1:
            return self._run_graph_shard(shard, embedding_dtype)
        if self._host is not None:   # device="cpu": micro-batch after micro-batch, as api.py:211-230
            outputs: list[np.ndarray] = []
            for start, stop in bounds:
                outputs.extend(self._run_graph_shard(shard.slice(start, stop), embedding_dtype))
            return outputs
        # Several micro-batches, three kinds of thread: PACKERS slice the shard, rebase and
        # range-check the edges and copy the micro-batch's arrays into pinned staging (up to
        # `slots - 1` micro-batches ahead); THIS thread uploads, launches and records events
        # (~0.1 ms per micro-batch); COPIERS bring the embeddings back through their pinned ring
        # into one host block.  PCIe is full duplex and the host copies run in parallel, so the
        # call is bound by the 230 MB of D2H (config 2), not by a thread.  Compute stays on ONE
        # stream in micro-batch order (one encoder, one workspace).
        torch_dtype, _code, exact = device_output_dtype(embedding_dtype)
        if self._uploader is None:
            self._uploader = _Uploader(self._engine.device)
        if self._preparer is None:
            self._preparer = ThreadPoolExecutor(max_workers=_PACKERS,
                                                thread_name_prefix="ginfinity-prep")
        uploader = self._uploader
        core_counts = shard.core_count_array()
        produced = np.dtype(embedding_dtype) if exact else np.dtype(np.float64)
        total_rows = int(core_counts.sum())
        host_block, fetch, direct = self._landing(total_rows, produced, torch_dtype, exact)
        device_rows = self._device_rows(total_rows, torch_dtype)

        def prepare(slot: int, start: int, stop: int):
            return self._pack_microbatch(uploader, slot, shard, start, stop)

        assert MICROBATCH_GROUP <= uploader.slots
        jobs: list = []
        pending = []
        first_row = 0
        try:
            for group in _groups(len(bounds), ramp=True):
                # the micro-batches of a group share every launch (gfy_encode_coo_batch): a
                # 60,000-node micro-batch by itself gives a CU less than one round of tiles
                members, group_row = [], first_row
                # packers run ahead of this thread, but never into a staging slot whose last
                # user has not been launched: a slot is guarded by the event of the GROUP that
                # read it (`hold`, below), so micro-batch j may be packed once micro-batch
                # j - slots belongs to a group in front of this one
                while len(jobs) < len(bounds) and len(jobs) < group.start + uploader.slots:
                    a, b = bounds[len(jobs)]
                    jobs.append(self._preparer.submit(prepare, uploader.reserve(), a, b))
                for index in group:
                    packed, kept = jobs[index].result()
                    features, edge_index, edge_types, out_rows, node_ptr, edge_ptr = \
                        uploader.send(packed, mapped=direct)
                    # mapped inputs are read over PCIe: the record-range set-up reads a record's
                    # destinations once per 256 of its rows, the counting kernel every array once
                    if node_ptr is not None and not direct:
                        attach_records(edge_index, node_ptr, edge_ptr)
                    members.append((packed, (features, edge_index, edge_types, out_rows,
                                             device_rows[first_row:first_row + kept])))
                    start, stop = bounds[index]
                    pending.append((first_row, kept, core_counts[start:stop]))
                    first_row += kept
                self._engine.encode_coo_group([tensors for _packed, tensors in members])
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(self._engine.device))
                if direct:
                    for packed, _tensors in members:
                        uploader.hold(packed, ready)
                # (the workers run no interpreter-level loops: a worker cutting 400 views holds
                # the GIL for 0.3 ms at a time and this thread, which needs it between every two
                # enqueues, took 0.6 ms per micro-batch instead of 0.1)
                landing = fetch(device_rows[group_row:first_row], ready, group_row,
                                first_row - group_row)
                for slot in range(len(pending) - len(members), len(pending)):
                    pending[slot] = (landing,) + pending[slot]
        except BaseException:
            _settle(jobs)      # no packer may still be writing a staging slot after we leave
            raise
        # the per-record views are cut here, group by group as the copies land
        outputs: list[np.ndarray] = []
        for job, row, kept, counts in pending:
            job.result()
            outputs.extend(self._splitter(counts, embedding_dtype, exact)(
                host_block[row:row + kept]))
        return outputs

    @staticmethod
    def _pack_microbatch(uploader: "_Uploader", slot: int, shard: GraphShard, start: int,
                         stop: int):
        """Records [start, stop) of ``shard`` → pinned staging slot ``slot`` (packer threads):
        the arrays of GraphShard.slice(start, stop) (graph.py:414-444: edge indices rebased to the
        first node of the range) without building — and re-validating — a GraphShard per
        micro-batch; the one check of GraphShard.__post_init__ that depends on the slice
        (graph.py:318-321 after the rebasing: an edge that leaves the micro-batch's node range,
        which the whole-shard range check cannot see) is made on the way into pinned memory and
        refused exactly as the reference refuses it.  Returns ``(packed, kept core rows)``."""
        n0, n1 = int(shard.node_ptr[start]), int(shard.node_ptr[stop])
        e0, e1 = int(shard.edge_ptr[start]), int(shard.edge_ptr[stop])
        roles = shard.node_roles[n0:n1]
        rows, kept = None, n1 - n0
        if roles.any():                      # context nodes: dropped at the head's store
            core = roles == 0
            kept = int(np.count_nonzero(core))
            rows = np.cumsum(core, dtype=np.int32) - np.int32(1)
            rows[~core] = -1
        # the record boundaries go up as they are (the kernels subtract the first entry): two
        # arrays of records + 1 int64, and COO -> tile plans then needs no global atomics
        node_ptr = edge_ptr = None
        if records_pay(shard.node_ptr[start:stop + 1], shard.edge_ptr[start:stop + 1]):
            node_ptr, edge_ptr = shard.node_ptr[start:stop + 1], shard.edge_ptr[start:stop + 1]
        packed = uploader.pack(slot, (
            shard.node_features[n0:n1], (shard.edge_index[:, e0:e1], np.int32(n0), n0, n1),
            shard.edge_types[e0:e1], rows, node_ptr, edge_ptr))
        return packed, kept

    @staticmethod
    def _microbatch_bytes(shard: GraphShard, start: int, stop: int) -> int:
        """Staging bytes of records [start, stop) as ``_pack_microbatch_at`` lays them out."""
        n = int(shard.node_ptr[stop]) - int(shard.node_ptr[start])
        e = int(shard.edge_ptr[stop]) - int(shard.edge_ptr[start])
        pad = _Uploader.padded
        total = pad(n * shard.node_features.shape[1] * 4) + pad(2 * e * 4) + pad(e)
        total += pad(n * 4)                                   # out_rows (used or not)
        total += 2 * pad((stop - start + 1) * 8)              # node_ptr, edge_ptr (used or not)
        return total

    @classmethod
    def _pack_items(cls, uploader: "_Uploader", slot: int, items) -> list:
        """Packer thread: ``_pack_microbatch_at`` for ``(base, shard, start, stop)`` items of one slot."""
        return [cls._pack_microbatch_at(uploader, slot, base, shard, start, stop)
                for base, shard, start, stop in items]

    @staticmethod
    def _pack_microbatch_at(uploader: "_Uploader", slot: int, base: int, shard: GraphShard,
                            start: int, stop: int):
        """``_pack_microbatch`` into a prepared group slot at byte ``base``: ``(offsets of the six
        arrays in the slot — 0 where an array is absent —, nodes, edges, records or 0, kept)``."""
        if NATIVE_PACKER and _packable(shard):
            # one call without the interpreter lock (csrc/gfy_base.cpp): the numpy form below
            # kept a pool of packers behind one lock — 128 micro-batches, 561 MB: 19-24 ms
            offsets, counts = (ctypes.c_int64 * 6)(), (ctypes.c_int64 * 4)()
            lib = native.library()
            status = lib.gfy_pack_microbatch(
                shard.node_features.ctypes.data, int(shard.node_features.shape[1]),
                shard.edge_index.ctypes.data, int(shard.edge_index.shape[1]),
                shard.edge_types.ctypes.data, shard.node_roles.ctypes.data,
                shard.node_ptr.ctypes.data, shard.edge_ptr.ctypes.data, int(start), int(stop), 1,
                uploader._staging[slot].data_ptr(), int(base), offsets, counts)
            if status != native.GFY_OK:
                message = lib.gfy_last_error().decode("utf-8", "replace")
                if "edge index outside" in message:
                    raise GraphValidationError(message)
                native.check(status, "gfy_pack_microbatch")
            return list(offsets), int(counts[0]), int(counts[1]), int(counts[2]), int(counts[3])
        n0, n1 = int(shard.node_ptr[start]), int(shard.node_ptr[stop])
        e0, e1 = int(shard.edge_ptr[start]), int(shard.edge_ptr[stop])
        roles = shard.node_roles[n0:n1]
        rows, kept = None, n1 - n0
        if roles.any():
            core = roles == 0
            kept = int(np.count_nonzero(core))
            rows = np.cumsum(core, dtype=np.int32) - np.int32(1)
            rows[~core] = -1
        node_ptr = edge_ptr = None
        if records_pay(shard.node_ptr[start:stop + 1], shard.edge_ptr[start:stop + 1]):
            node_ptr, edge_ptr = shard.node_ptr[start:stop + 1], shard.edge_ptr[start:stop + 1]
        arrays = (shard.node_features[n0:n1], (shard.edge_index[:, e0:e1], np.int32(n0), n0, n1),
                  shard.edge_types[e0:e1], rows, node_ptr, edge_ptr)
        offsets = uploader.pack_at(slot, base, arrays)
        present = (True, True, True, rows is not None, node_ptr is not None, edge_ptr is not None)
        return ([offset if here else -1 for offset, here in zip(offsets, present)],
                n1 - n0, e1 - e0, stop - start if node_ptr is not None else 0, kept)

    def _device_rows(self, rows: int, torch_dtype: torch.dtype) -> torch.Tensor:
        """[rows, 128] of ``torch_dtype`` on the device, a view of ONE block the encoder keeps
        (grown when a call needs more): the micro-batches of a call write their embeddings into
        row ranges of it.  Per-micro-batch output tensors came from torch's caching allocator
        with ``record_stream`` on the copy stream, so their reuse waited for events and a call
        could run into ``hipMalloc`` — device-wide, and 15 MB D2H copies next to it took 1.2 ms
        instead of 0.3 (profiles/README.md, "D2H").  Safe to reuse call after call: a call returns
        only when its last copy has landed (one encoder = serialized inference)."""
        width = self.embedding_dimension
        need = rows * width * torch.empty((), dtype=torch_dtype).element_size()
        if self._device_block is None or self._device_block.numel() < need:
            self._device_block = None                      # release before growing
            self._device_block = torch.empty(max(need, 1 << 20), dtype=torch.uint8,
                                             device=self._engine.device)
        return self._device_block[:need].view(torch_dtype).view(rows, width)

    def _landing(self, rows: int, produced: np.dtype, torch_dtype: torch.dtype, exact: bool):
        """The host block of one call and the function that brings a micro-batch's device
        block into rows [first, first + count) of it: ``(host_block, fetch, direct)``;
        ``fetch(block, ready, first, count)`` returns an object whose ``result()`` waits for
        the rows; ``direct``: the block is page-locked and written by the device."""
        width = self.embedding_dimension
        pinned = self.pinned_outputs
        if pinned is None:
            pinned = _pinned_alive[0] + rows * width * produced.itemsize <= PINNED_RESULT_LIMIT
        if pinned and exact and not self.independent_outputs:
            if self._direct is None:
                self._direct = _DirectDownloader(self._engine.device)
            landing = _pinned_block((rows, width), torch_dtype)
            direct = self._direct
            direct.abandon()                  # (copies a call that raised never enqueued)
            return landing.numpy(), lambda block, ready, first, count: direct.submit(
                block, ready, landing[first:first + count]), True
        if self._copier is None:
            self._copier = _Downloader(self._engine.device)
        host_block = np.empty((rows, width), dtype=produced)
        _advise_huge_pages(host_block)
        copier = self._copier
        return host_block, lambda block, ready, first, count: copier.submit(
            block, ready, lambda host: None, host_block[first:first + count]), False

    def _splitter(self, core_counts, embedding_dtype: np.dtype, exact: bool):
        """host block → the per-record arrays of one micro-batch (views of the block, or
        copies with ``independent_outputs``)."""
        ends = np.cumsum(core_counts).tolist()
        starts = [0] + ends[:-1]
        independent = self.independent_outputs

        def finish(host: np.ndarray) -> list[np.ndarray]:
            if not exact:
                host = host.astype(embedding_dtype)
            parts = [host[a:b] for a, b in zip(starts, ends)]   # (np.split: 3x the time)
            return [part.copy() for part in parts] if independent else parts
        return finish

    # -- the seam (reference: api.py:232-260) -----------------------------------------
    def _encode_shard_device(self, shard: GraphShard, out_dtype: torch.dtype
                             ) -> torch.Tensor:
        """One micro-batch → [core_nodes,128] tensor left on the device."""
        return self._engine.encode_arrays(
            shard.node_features, shard.edge_index, shard.edge_types,
            shard.node_roles, out_dtype=out_dtype, normalise=True)

    def _run_graph_shard(self, shard: GraphShard, embedding_dtype: np.dtype
                         ) -> list[np.ndarray]:
        if self._host is not None:
            block = self._host.encode_arrays(
                shard.node_features, shard.edge_index, shard.edge_types, shard.node_roles,
                embedding_dtype=embedding_dtype)
            return self._splitter(shard.core_counts, embedding_dtype, True)(block)
        torch_dtype, _code, exact = device_output_dtype(embedding_dtype)
        block = self._encode_shard_device(shard, torch_dtype).cpu().numpy()
        return self._splitter(shard.core_counts, embedding_dtype, exact)(block)

    # -- device-resident results (MI355X-side extension; no reference counterpart) ----------
    def stage_shards(self, shards: "GraphShard | Sequence[GraphShard]", *,
                     max_batch_nodes: int = 60_000, max_batch_edges: int = 300_000
                     ) -> tuple[list[tuple], list[tuple[int, ...]]]:
        """Upload the micro-batches of one shard or of several: ``(staged, counts)`` — the
        device arrays of every micro-batch, in order (the input of ``encode_staged``), and
        the per-record core row counts of every shard.  Same packing, limits and errors as
        ``encode_graphs`` (api.py:196-230), the slices validated as the reference validates
        them (``GraphShard.slice``)."""
        if self._engine is None:
            raise ValueError("device-resident encoding needs a GPU encoder (device='cuda')")
        if isinstance(shards, GraphShard):
            shards = [shards]
        staged: list[tuple] = []
        counts: list[tuple[int, ...]] = []
        for shard in shards:
            shard = self._checked_shard(shard, max_batch_nodes, max_batch_edges)
            counts.append(shard.core_counts)
            for a, b in microbatch_bounds(shard.lengths, shard.edge_counts,
                                          max_batch_nodes, max_batch_edges):
                piece = shard if (a, b) == (0, shard.record_count) else shard.slice(a, b)
                staged.append(self._engine.upload_arrays(
                    piece.node_features, piece.edge_index, piece.edge_types, piece.node_roles,
                    node_ptr=piece.node_ptr, edge_ptr=piece.edge_ptr))
        return staged, counts

    def encode_staged(self, staged: Sequence[tuple], *,
                      out: torch.Tensor | None = None) -> torch.Tensor:
        """fp16 embeddings of staged micro-batches as ONE device tensor ([total core rows, 128]),
        the micro-batches issued in groups of ``MICROBATCH_GROUP`` through
        ``gfy_encode_coo_batch`` — nothing crosses PCIe, nothing returns to the host.  ``out``:
        a tensor of that shape to write into (steady-state loops: no allocation per call)."""
        if self._engine is None:
            raise ValueError("device-resident encoding needs a GPU encoder (device='cuda')")
        rows = sum(kept for *_arrays, kept in staged)
        block = out
        if block is None:
            block = torch.empty((rows, self.embedding_dimension), dtype=torch.float16,
                                device=self._engine.device)
        if (tuple(block.shape) != (rows, self.embedding_dimension)
                or block.dtype != torch.float16 or not block.is_contiguous()):
            raise ValueError("out must be a contiguous float16 [total core rows, 128] tensor")
        groups = _groups(len(staged))
        # Two groups in flight (fp16 model): the tail of one group's layer launches — the last,
        # partial round of tiles — runs under the next group's workgroups.  Each lane is an
        # encoder of its own (hidden-state buffers, workspace) on a stream of its own; the
        # caller's stream waits for both.  ``STAGED_LANES = 1``: one group after the other.
        lanes = self._two_lanes() if (STAGED_LANES > 1 and len(groups) > 1
                                      and not self.full_precision) else None
        current = torch.cuda.current_stream(self._engine.device)
        if lanes:
            ready = torch.cuda.Event()
            ready.record(current)                 # the staged arrays, `out`'s previous readers
            for _engine, stream in lanes:
                stream.wait_event(ready)
        first = 0
        for number, group in enumerate(groups):
            members = []
            for index in group:
                features, edge_index, edge_types, out_rows, kept = staged[index]
                members.append((features, edge_index, edge_types, out_rows,
                                block[first:first + kept]))
                first += kept
            if lanes:
                engine, stream = lanes[number % len(lanes)]
                with torch.cuda.stream(stream):
                    engine.encode_coo_group(members)
            else:
                self._engine.encode_coo_group(members)
        if lanes:
            for _engine, stream in lanes:
                done = torch.cuda.Event()
                done.record(stream)
                current.wait_event(done)
        return block

    def _two_lanes(self) -> "list[tuple[DeviceEncoder, torch.cuda.Stream]]":
        if self._lanes is None:
            device = self._engine.device
            self._lanes = [(self._engine, torch.cuda.Stream(device=device)),
                           (self._engine.twin(), torch.cuda.Stream(device=device))]
        twin = self._lanes[1][0]
        for option, value in self._engine._options.items():   # switches set since the twin was made
            if twin._options.get(option) != value:
                twin.set_option(option, value)
        return self._lanes

    def encode_shards_device(self, shards: Sequence[GraphShard], *,
                             max_batch_nodes: int = 60_000, max_batch_edges: int = 300_000,
                             out: torch.Tensor | None = None
                             ) -> tuple[torch.Tensor, list[tuple[int, ...]]]:
        """Several shards → one device block of all their core rows (shard after shard) and
        the per-record row counts of every shard: what a rank of ``parallel`` encodes.  The
        host arrays STREAM in: packer threads slice and rebase the micro-batches into a ring of
        page-locked staging slots, a copy stream brings group g + 1 up while group g (four
        micro-batches, one ``gfy_encode_coo_batch``) is computed — the call is bound by the
        larger of the two (4.4 MB per 60,000-node micro-batch against ≈ 75 µs of compute:
        level), not by their sum.  Nothing returns to the host."""
        if self._engine is None:
            raise ValueError("device-resident encoding needs a GPU encoder (device='cuda')")
        engine, device = self._engine, self._engine.device
        plan: list[tuple[GraphShard, int, int]] = []
        counts: list[tuple[int, ...]] = []
        for shard in shards:
            shard = self._checked_shard(shard, max_batch_nodes, max_batch_edges)
            counts.append(shard.core_counts)
            plan += [(shard, a, b) for a, b in self._shard_bounds(
                shard, max_batch_nodes, max_batch_edges)]
        rows = sum(sum(per_record) for per_record in counts)
        block = out
        if block is None:
            block = torch.empty((rows, self.embedding_dimension), dtype=torch.float16,
                                device=device)
        if (tuple(block.shape) != (rows, self.embedding_dimension)
                or block.dtype != torch.float16 or not block.is_contiguous()):
            raise ValueError("out must be a contiguous float16 [total core rows, 128] tensor")
        if self._uploader is None:
            self._uploader = _Uploader(device)
        if self._preparer is None:
            self._preparer = ThreadPoolExecutor(max_workers=_PACKERS,
                                                thread_name_prefix="ginfinity-prep")
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=device)
        uploader, copies = self._uploader, self._copy_stream
        compute = torch.cuda.current_stream(device)
        # A GROUP of micro-batches shares one staging slot, one H2D copy and one event: per
        # micro-batch the launching thread then only hands four packers their byte ranges and
        # turns offsets into addresses (one copy and one event per micro-batch were ~50 us each
        # of this thread: 128 micro-batches could not start faster than one per 180 us).
        groups = list(_groups(len(plan)))
        submitted: list = []              # per group: (slot, total bytes, [jobs])

        def submit(group) -> None:
            slot, at, jobs_of = uploader.reserve(), 0, []
            bases = []
            for index in group:
                shard, a, b = plan[index]
                bases.append(at)
                at += self._microbatch_bytes(shard, a, b)
            uploader.prepare_slot(slot, at)
            items = [(base, *plan[index]) for index, base in zip(group, bases)]
            # the first groups: a packer per micro-batch (the copy engine is waiting for them);
            # behind them a packer per GROUP — enough of them are packed ahead, and a job
            # handed to the pool is 11 us of this thread
            pieces = [[item] for item in items] if len(submitted) < 2 else [items]
            for piece in pieces:
                jobs_of.append(self._preparer.submit(self._pack_items, uploader, slot, piece))
            submitted.append((slot, at, jobs_of))

        # ``HOST_FEED_LANES = 2``: two groups in flight as in ``encode_staged`` — a lane = an
        # encoder and a stream of its own; group g runs on lane g mod 2 behind ITS upload only,
        # the caller's stream waits for both at the end.  Not the default: see HOST_FEED_LANES.
        lanes = self._two_lanes() if (HOST_FEED_LANES > 1 and len(groups) > 1
                                      and not self.full_precision) else None
        if lanes:
            ready = torch.cuda.Event()
            ready.record(compute)                 # `out`'s previous readers
            for _engine, stream in lanes:
                stream.wait_event(ready)
        first = 0
        ahead = max(1, uploader.slots - 1)   # groups packed ahead of the one being launched
        try:
            for number, group in enumerate(groups):
                while len(submitted) < len(groups) and len(submitted) <= number + ahead - 1:
                    submit(groups[len(submitted)])
                slot, total, jobs_of = submitted[number]
                packed = [result for job in jobs_of for result in job.result()]
                lane_engine, lane_stream = lanes[number % len(lanes)] if lanes else (engine, compute)
                # one native call: H2D copy on the copy stream, the lane waits for it
                inputs = uploader.send_group(slot, total, copies, lane_stream)
                address = inputs.data_ptr()
                members = []
                for offsets, nodes, edges, records, kept in packed:
                    members.append(engine.pointer_shard(
                        [address + offset if offset >= 0 else 0 for offset in offsets],
                        nodes=nodes, edges=edges, records=records,
                        out=block[first:first + kept], keep=inputs))
                    first += kept
                with torch.cuda.stream(lane_stream):
                    lane_engine.encode_coo_group_pointers(members)
        except BaseException:
            _settle([job for _slot, _total, jobs_of in submitted for job in jobs_of])
            raise
        finally:
            if lanes:
                for _engine, stream in lanes:
                    done = torch.cuda.Event()
                    done.record(stream)
                    compute.wait_event(done)
        return block, counts

    def encode_graphs_device(self, shard: GraphShard, *,
                             max_batch_nodes: int = 60_000,
                             max_batch_edges: int = 300_000
                             ) -> tuple[torch.Tensor, tuple[int, ...]]:
        """fp16 embeddings of the whole shard as ONE device tensor ([total core nodes, 128])
        plus the per-record row counts — the input format of ``ginfinity_amd.distance``."""
        block, counts = self.encode_shards_device([shard], max_batch_nodes=max_batch_nodes,
                                                  max_batch_edges=max_batch_edges)
        return block, counts[0]


__all__ = ["Ginfinity", "ModelIntegrityError", "default_alignment_parameters",
           "microbatch_bounds"]
