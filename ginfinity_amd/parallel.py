"""Multi-GPU operation: one process per GPU, ``torch.distributed`` over RCCL.

The reference has no distributed code — it delegates to an external scheduler
running one process per shard (docs/GRAPH_PIPELINE.md:22-24).  On an 8-GPU
MI355X node the same decomposition maps to ranks:

* **encode is shard-parallel and needs no collective** — graphs never share
  edges across records (graph.py:392-395), weights (0.6 MB) are replicated,
  rank r encodes shards r, r+W, r+2W, …;
* **the one exchange step** is the cross-shard nearest-neighbour search: the fp16
  embedding blocks are all-gathered in chunks (``all_gather_into_tensor`` per chunk into
  one of two staging buffers — RCCL over xGMI on GPUs, gloo on CPU tensors in the tests)
  and every chunk is searched against the rank's own rows while the next one is in
  flight; per row the best (value, global row) is merged on the device.  Row reductions
  stay on the rank that owns the row: no second collective.

With world size 1 (or no process group) every function degenerates to the
single-GPU path without touching ``torch.distributed``.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.distributed as dist


def world(group=None) -> tuple[int, int]:
    """(rank, world_size); (0, 1) when no process group is initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_assignment(n_shards: int, world_size: int, rank: int) -> list[int]:
    """Round-robin shard ownership: shard s belongs to rank s mod W."""
    if not 0 <= rank < world_size:
        raise ValueError("rank outside the world")
    return list(range(rank, n_shards, world_size))


def encode_owned_shards(encoder, shards: Sequence, *, group=None,
                        max_batch_nodes: int = 60_000,
                        max_batch_edges: int = 300_000
                        ) -> tuple[torch.Tensor, list[int], list[tuple[int, ...]]]:
    """Encode this rank's share of ``shards`` (no communication).

    Returns (fp16 device block of all owned core rows, owned shard indices,
    per-shard per-record row counts)."""
    rank, size = world(group)
    owned = shard_assignment(len(shards), size, rank)
    if owned:   # all owned shards in one call: their micro-batches share launches in groups
        merged, counts = encoder.encode_shards_device(
            [shards[index] for index in owned], max_batch_nodes=max_batch_nodes,
            max_batch_edges=max_batch_edges)
    else:
        counts = []
        merged = torch.empty((0, 128), dtype=torch.float16,
                             device=encoder._engine.device)
    return merged, owned, counts


def all_gather_rows(block: torch.Tensor, group=None
                    ) -> tuple[torch.Tensor, list[int]]:
    """Gather ``[rows_r, 128]`` blocks of all ranks into one ``[sum rows, 128]``
    tensor in rank order; returns it with the row offset of every rank's block
    (length W+1).  Blocks may differ in size: they are padded to the largest
    for one equal-count ``all_gather_into_tensor`` and compacted afterwards.
    For SMALL exchanges (manifests, tests): the search path, ``cross_shard_nearest``,
    gathers in chunks and never holds all rows at once."""
    rank, size = world(group)
    rows = int(block.shape[0])
    if size == 1:
        return block, [0, rows]
    counts = torch.tensor([rows], dtype=torch.int64, device=block.device)
    all_counts = torch.empty(size, dtype=torch.int64, device=block.device)
    dist.all_gather_into_tensor(all_counts, counts, group=group)
    sizes = [int(v) for v in all_counts.tolist()]
    widest = max(sizes)
    padded = block
    if rows < widest:
        padded = torch.zeros((widest, block.shape[1]), dtype=block.dtype,
                             device=block.device)
        padded[:rows] = block
    gathered = torch.empty((size * widest, block.shape[1]), dtype=block.dtype,
                           device=block.device)
    dist.all_gather_into_tensor(gathered, padded.contiguous(), group=group)
    offsets = [0]
    for value in sizes:
        offsets.append(offsets[-1] + value)
    if all(value == widest for value in sizes):
        return gathered, offsets
    pieces = [gathered[r * widest:r * widest + sizes[r]] for r in range(size)]
    return torch.cat(pieces, dim=0), offsets


def _search_block(search, local: torch.Tensor, other: torch.Tensor, metric: str,
                  own_first: int | None, workspace=None) -> tuple[torch.Tensor, torch.Tensor]:
    """``search(local, other)``; when ``other`` is rows [own_first, own_first + len(other)) of
    ``local`` itself, every row skips itself (``window_first``: one call for the rows in front
    of the window, inside it and behind it)."""
    extra = {} if workspace is None else {"workspace": workspace}
    if own_first is None:
        return search(local, other, metric=metric, **extra)
    return search(local, other, metric=metric, window_first=own_first, **extra)


def cross_shard_nearest(block: torch.Tensor, *, metric: str = "l2", group=None,
                        chunk_rows: int = 1 << 20, search=None
                        ) -> tuple[torch.Tensor, torch.Tensor, list[int]]:
    """Nearest other embedding, over ALL ranks' rows, of every local row.

    The blocks are exchanged in chunks of ``chunk_rows`` rows per rank — one equal-count
    ``all_gather_into_tensor`` per chunk into one of two staging buffers (RCCL over xGMI;
    the gathered 15.7 GB of BASELINE configs[4] never exist at once, nothing is compacted
    or copied a second time) — and chunk k is searched (``distance.nearest`` on the matrix
    cores, one call per rank's piece) while chunk k+1 is in flight.  Per local row the
    best (value, global row) is merged on the device; ties go to the lowest global row.
    Rows of other ranks never leave the rank that owns the query row, so there is no second
    collective.  A rank without rows takes part in the collectives and returns empty
    results.  With world size 1 the same code runs without ``torch.distributed``.

    Returns (values float32 [rows_r], global row indices int64 [rows_r], rank offsets W+1).
    ``search`` (tests) replaces ``distance.nearest``."""
    workspace = None
    if search is None:
        from . import distance
        search = distance.nearest
        workspace = distance.NearestWorkspace()     # one set of buffers for every piece
    if metric not in ("l2", "cosine"):
        raise ValueError("metric must be 'l2' or 'cosine'")
    rank, size = world(group)
    rows = int(block.shape[0])
    device = block.device
    if size > 1:
        mine = torch.tensor([rows], dtype=torch.int64, device=device)
        everyone = torch.empty(size, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(everyone, mine, group=group)
        sizes = [int(v) for v in everyone.tolist()]
    else:
        sizes = [rows]
    offsets = [0]
    for value in sizes:
        offsets.append(offsets[-1] + value)
    widest = max(sizes)
    chunk_rows = max(1, min(int(chunk_rows), max(widest, 1)))
    chunks = (widest + chunk_rows - 1) // chunk_rows
    worse = float("inf") if metric == "l2" else float("-inf")
    best_value = torch.full((rows,), worse, dtype=torch.float32, device=device)
    best_index = torch.full((rows,), -1, dtype=torch.int64, device=device)
    if size > 1:
        staging = [torch.empty((size, chunk_rows, block.shape[1]), dtype=block.dtype, device=device)
                   for _ in range(min(2, max(chunks, 1)))]
        outgoing = [torch.zeros((chunk_rows, block.shape[1]), dtype=block.dtype, device=device)
                    for _ in range(len(staging))]

    def start(chunk: int):
        first = chunk * chunk_rows
        have = max(0, min(rows - first, chunk_rows))
        buffer = outgoing[chunk % len(outgoing)]
        if have:
            buffer[:have] = block[first:first + have]
        return dist.all_gather_into_tensor(
            staging[chunk % len(staging)].view(size * chunk_rows, block.shape[1]), buffer,
            group=group, async_op=True)

    pending = start(0) if size > 1 and chunks else None
    for chunk in range(chunks):
        first = chunk * chunk_rows
        if size > 1:
            arriving = pending
            pending = start(chunk + 1) if chunk + 1 < chunks else None   # in flight under the search
            arriving.wait()
        for other in range(size):
            valid = max(0, min(sizes[other] - first, chunk_rows))
            if valid == 0 or rows == 0:
                continue
            piece = (staging[chunk % len(staging)][other, :valid] if size > 1
                     else block[first:first + valid])
            values, indices = _search_block(search, block, piece, metric,
                                            first if other == rank else None, workspace)
            indices = indices.to(torch.int64) + (offsets[other] + first)
            better = (values < best_value) if metric == "l2" else (values > best_value)
            better |= (values == best_value) & (indices < best_index)
            best_value = torch.where(better, values, best_value)
            best_index = torch.where(better, indices, best_index)
    return best_value, best_index, offsets


__all__ = ["world", "shard_assignment", "encode_owned_shards",
           "all_gather_rows", "cross_shard_nearest"]
