"""Multi-GPU operation: one process per GPU, ``torch.distributed`` over RCCL.

The reference has no distributed code — it delegates to an external scheduler
running one process per shard (docs/GRAPH_PIPELINE.md:22-24).  On an 8-GPU
MI355X node the same decomposition maps to ranks:

* **encode is shard-parallel and needs no collective** — graphs never share
  edges across records (graph.py:392-395), weights (0.6 MB) are replicated,
  rank r encodes shards r, r+W, r+2W, …;
* **the one exchange step** is the cross-shard nearest-neighbour search: every
  rank all-gathers the fp16 embedding blocks (``all_gather_into_tensor`` —
  RCCL over xGMI on GPUs, gloo on CPU tensors in the tests), then searches its
  own rows against all rows with the (i, i + own offset) pair excluded.  Row
  reductions stay on the rank that owns the row: no second collective.

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
    blocks, counts = [], []
    for index in owned:
        block, per_record = encoder.encode_graphs_device(
            shards[index], max_batch_nodes=max_batch_nodes,
            max_batch_edges=max_batch_edges)
        blocks.append(block)
        counts.append(per_record)
    if blocks:
        merged = blocks[0] if len(blocks) == 1 else torch.cat(blocks, dim=0)
    else:
        merged = torch.empty((0, 128), dtype=torch.float16,
                             device=encoder._engine.device)
    return merged, owned, counts


def all_gather_rows(block: torch.Tensor, group=None
                    ) -> tuple[torch.Tensor, list[int]]:
    """Gather ``[rows_r, 128]`` blocks of all ranks into one ``[sum rows, 128]``
    tensor in rank order; returns it with the row offset of every rank's block
    (length W+1).  Blocks may differ in size: they are padded to the largest
    for one equal-count ``all_gather_into_tensor`` and compacted afterwards."""
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


def cross_shard_nearest(block: torch.Tensor, *, metric: str = "l2", group=None
                        ) -> tuple[torch.Tensor, torch.Tensor, list[int]]:
    """Nearest other embedding, over ALL ranks' rows, of every local row.

    Returns (values [rows_r], global row indices [rows_r], rank offsets)."""
    from . import distance
    rank, _size = world(group)
    everything, offsets = all_gather_rows(block, group)
    values, indices = distance.nearest(
        block, everything, metric=metric, exclude_offset=offsets[rank])
    return values, indices, offsets


__all__ = ["world", "shard_assignment", "encode_owned_shards",
           "all_gather_rows", "cross_shard_nearest"]
