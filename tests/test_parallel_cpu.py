"""Multi-rank paths on CPU: gloo backend, world size 2 (and the world-size-1
degenerate path).  The collective layer is exercised here; the kernels behind
it are covered by the single-GPU tests."""
from __future__ import annotations

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ginfinity_amd import parallel


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rows(rank: int, count: int) -> torch.Tensor:
    rng = np.random.default_rng(100 + rank)
    return torch.from_numpy(rng.standard_normal((count, 128)).astype(np.float16))


def _worker(rank: int, size: int, port: int, sizes: list[int], queue) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        block = _rows(rank, sizes[rank])
        gathered, offsets = parallel.all_gather_rows(block)
        want = torch.cat([_rows(r, sizes[r]) for r in range(size)])
        ok = (offsets == list(np.concatenate(([0], np.cumsum(sizes))))
              and gathered.shape == want.shape and torch.equal(gathered, want)
              and torch.equal(gathered[offsets[rank]:offsets[rank + 1]], block))
        owned = parallel.shard_assignment(7, size, rank)
        queue.put((rank, bool(ok), owned, parallel.world()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [[5, 5], [3, 9], [0, 4]])
def test_all_gather_rows_world2_gloo(sizes):
    context = mp.get_context("spawn")
    queue = context.Queue()
    port = _free_port()
    procs = [context.Process(target=_worker, args=(r, 2, port, sizes, queue))
             for r in range(2)]
    for p in procs:
        p.start()
    results = [queue.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    assert [r[1] for r in results] == [True, True]
    assert results[0][2] == [0, 2, 4, 6] and results[1][2] == [1, 3, 5]
    assert [r[3] for r in results] == [(0, 2), (1, 2)]


def test_world_size_one_needs_no_process_group():
    assert parallel.world() == (0, 1)
    block = _rows(0, 6)
    gathered, offsets = parallel.all_gather_rows(block)
    assert gathered is block and offsets == [0, 6]
    assert parallel.shard_assignment(5, 1, 0) == [0, 1, 2, 3, 4]
    with pytest.raises(ValueError):
        parallel.shard_assignment(5, 2, 2)


def test_shard_assignment_is_a_partition():
    for size in (1, 2, 3, 8):
        seen = sorted(s for r in range(size)
                      for s in parallel.shard_assignment(1024, size, r))
        assert seen == list(range(1024))
        loads = [len(parallel.shard_assignment(1024, size, r)) for r in range(size)]
        assert max(loads) - min(loads) <= 1


# ---- chunked, overlapped cross-shard search (the collective + merge logic; the kernel behind
# ---- `search` is covered by the single-GPU tests) ------------------------------------------

def _oracle_search(a, b, *, metric="l2", exclude_offset=None, window_first=None):
    """float64 definition (oracle/gine_numpy.py) with the library's signature: test stand-in
    for distance.nearest on CPU tensors."""
    from oracle import gine_numpy as G
    an, bn = a.numpy(), b.numpy()
    full = G.pairwise_l2(an, bn) if metric == "l2" else G.pairwise_cosine(an, bn)
    if window_first is not None:           # b = rows [window_first, ...) of a: skip (k + j, j)
        exclude_offset = -int(window_first)
    if exclude_offset is not None and (exclude_offset >= 0 or window_first is not None):
        for i in range(an.shape[0]):
            if 0 <= i + exclude_offset < bn.shape[0]:
                full[i, i + exclude_offset] = np.inf if metric == "l2" else -np.inf
    index = full.argmin(axis=1) if metric == "l2" else full.argmax(axis=1)
    value = full[np.arange(an.shape[0]), index] if an.shape[0] else np.zeros(0)
    return (torch.from_numpy(value.astype(np.float32)), torch.from_numpy(index.astype(np.int32)))


def _nearest_worker(rank: int, size: int, port: int, sizes: list[int], metric: str,
                    chunk_rows: int, queue) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        block = _rows(rank, sizes[rank])
        values, indices, offsets = parallel.cross_shard_nearest(
            block, metric=metric, chunk_rows=chunk_rows, search=_oracle_search)
        queue.put((rank, values.numpy(), indices.numpy(), offsets))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sizes,chunk_rows", [([7, 7], 3), ([5, 11], 4), ([0, 6], 4),
                                              ([9, 0, 4], 5), ([3, 8, 5], 100)])
@pytest.mark.parametrize("metric", ["l2", "cosine"])
def test_cross_shard_nearest_chunked_gloo(sizes, chunk_rows, metric):
    """Unequal and empty blocks, chunks smaller than / larger than the blocks, two and three
    ranks: every rank's rows must find their nearest OTHER row among all ranks' rows."""
    from oracle import gine_numpy as G
    size = len(sizes)
    context = mp.get_context("spawn")
    queue = context.Queue()
    port = _free_port()
    procs = [context.Process(target=_nearest_worker,
                             args=(r, size, port, sizes, metric, chunk_rows, queue))
             for r in range(size)]
    for p in procs:
        p.start()
    results = sorted((queue.get(timeout=180) for _ in procs), key=lambda item: item[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    everything = np.concatenate([_rows(r, sizes[r]).numpy() for r in range(size)])
    full = G.pairwise_l2(everything, everything) if metric == "l2" else G.pairwise_cosine(everything, everything)
    np.fill_diagonal(full, np.inf if metric == "l2" else -np.inf)
    want = full.argmin(axis=1) if metric == "l2" else full.argmax(axis=1)
    starts = np.concatenate(([0], np.cumsum(sizes)))
    for rank, values, indices, offsets in results:
        assert offsets == list(starts)
        lo, hi = starts[rank], starts[rank + 1]
        assert indices.shape == (hi - lo,) and indices.dtype == np.int64
        np.testing.assert_array_equal(indices, want[lo:hi])
        np.testing.assert_allclose(values, full[np.arange(lo, hi), want[lo:hi]], rtol=1e-6)


def test_cross_shard_nearest_world_size_one_needs_no_process_group():
    block = _rows(3, 13)
    values, indices, offsets = parallel.cross_shard_nearest(
        block, metric="cosine", chunk_rows=4, search=_oracle_search)
    direct_v, direct_i = _oracle_search(block, block, metric="cosine", exclude_offset=0)
    assert offsets == [0, 13]
    np.testing.assert_array_equal(indices.numpy(), direct_i.numpy().astype(np.int64))
    np.testing.assert_allclose(values.numpy(), direct_v.numpy())


# ---- bench.py --workload cross-shard: the DRIVER logic under gloo, GPU pieces stubbed ----------

class _StandInEncoder:
    """``Ginfinity.stage_shards`` / ``encode_staged`` on CPU tensors: a shard is its own rows, a
    staged micro-batch is two of them at most (so shards split into ragged pieces), "encoding"
    returns the rows — issued group by group with the product's own grouping (``api._groups``),
    the group sizes kept for the test."""

    def __init__(self) -> None:
        self.groups: list[list[int]] = []

    def stage_shards(self, shards, **_limits):
        staged, counts = [], []
        for shard in shards:
            counts.append((int(shard.shape[0]),))
            staged += [(shard[a:a + 2], None, None, None, int(shard[a:a + 2].shape[0]))
                       for a in range(0, int(shard.shape[0]), 2)]
        return staged, counts

    def encode_staged(self, staged, *, out=None):
        from ginfinity_amd import api
        rows = sum(kept for *_arrays, kept in staged)
        block = torch.empty((rows, 128), dtype=torch.float16) if out is None else out
        assert tuple(block.shape) == (rows, 128)
        first, sizes = 0, []
        for group in api._groups(len(staged)):
            sizes.append(len(group))
            for index in group:
                piece, *_rest, kept = staged[index]
                block[first:first + kept] = piece
                first += kept
        self.groups.append(sizes)
        return block


def _bench_cross_shard_worker(rank: int, size: int, port: int, shards: int, queue) -> None:
    import argparse
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    if str(root) not in sys.path:
        sys.path.insert(0, str(root))
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        args = argparse.Namespace(shards=shards, chunk_rows=5)
        # shard s = 4 + s rows, every row recognisably its shard's; "encoding" returns them
        make_shard = lambda s: _rows(100 + s, 4 + s)
        encoder = _StandInEncoder()
        line = bench.cross_shard(args, rank, rank, size, True, device=torch.device("cpu"),
                                 encoder=encoder, search=_oracle_search, make_shard=make_shard)
        queue.put((rank, line, encoder.groups))
    finally:
        dist.destroy_process_group()


def _run_cross_shard_driver(size: int, shards: int):
    context = mp.get_context("spawn")
    queue = context.Queue()
    port = _free_port()
    procs = [context.Process(target=_bench_cross_shard_worker, args=(r, size, port, shards, queue))
             for r in range(size)]
    for p in procs:
        p.start()
    results = {rank: (line, groups) for rank, line, groups in
               (queue.get(timeout=240) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def test_bench_cross_shard_driver_logic_world_size_two():
    """bench.py's cross-shard workload with the encoder and the matrix-core search replaced by
    CPU stand-ins: shard s goes to rank s mod W, every rank stages its shards and encodes them in
    groups of four micro-batches (the last group ragged), rank 0 prints ONE line whose totals
    and rank offsets describe all ranks' rows, and the sample it reports is the true nearest
    other row."""
    from oracle import gine_numpy as G
    size, shards = 2, 5
    results = _run_cross_shard_driver(size, shards)
    assert results[1][0] is None and results[0][0] is not None      # one line, from rank 0
    line = results[0][0]
    per_rank = [sum(4 + s for s in range(shards) if s % size == r) for r in range(size)]
    assert line["config"]["rows_total"] == sum(per_rank)
    assert line["config"]["rank_offsets"] == [0, per_rank[0], per_rank[0] + per_rank[1]]
    assert line["n_gpus"] == size and line["config"]["rccl_ranks"] == size
    assert line["unit"] == "pairs/s" and line["value"] > 0
    assert line["encode"]["seconds"] > 0 and line["stage"]["seconds"] > 0
    # rank 0: shards 0, 2, 4 = 4 + 6 + 8 rows = 2 + 3 + 4 staged pieces of two rows -> groups of
    # 4, 4, 1 (warm-up call, then the timed one); rank 1: shards 1, 3 = 3 + 4 pieces -> 4, 3
    assert results[0][1] == [[4, 4, 1], [4, 4, 1]]
    assert results[1][1] == [[4, 3], [4, 3]]
    everything = np.concatenate(
        [_rows(100 + s, 4 + s).numpy() for r in range(size) for s in range(shards) if s % size == r])
    full = G.pairwise_cosine(everything, everything)
    np.fill_diagonal(full, -np.inf)
    assert line["sample"][1] == int(full[0].argmax())
    assert abs(line["sample"][0] - full[0].max()) < 1e-6


def test_bench_cross_shard_driver_with_a_rank_that_owns_nothing():
    """Three ranks, two shards: rank 2 has no shard — it stages nothing, warms on a shard it does
    not keep, contributes zero rows to the exchange and still walks every fence and reduction."""
    size, shards = 3, 2
    results = _run_cross_shard_driver(size, shards)
    line = results[0][0]
    assert results[1][0] is None and results[2][0] is None
    assert line["config"]["rows_total"] == 4 + 5
    assert line["config"]["rank_offsets"] == [0, 4, 9, 9]
    assert results[2][1] == [[2]]            # the warm-up on make_shard(0) only (4 rows: 2 pieces)
    assert results[0][1] == [[2], [2]] and results[1][1] == [[3], [3]]
