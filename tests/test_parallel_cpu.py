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
