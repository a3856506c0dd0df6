"""BASELINE configs[4] in miniature: shard-parallel encode + RCCL all-gather of the
embedding blocks + cross-shard nearest-neighbour search.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port 29511 tools/bench_cross_shard.py --shards 64

Each rank encodes shards r, r+N, ... (no collective), all ranks all-gather the fp16
blocks (equal counts), each rank searches ITS rows against ALL rows with the
(i, i + own offset) pair excluded.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ginfinity_amd import Ginfinity, parallel, synthetic  # noqa: E402


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--shards", type=int, default=16)
    parser.add_argument("--metric", default="cosine")
    args = parser.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    if "RANK" in os.environ:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    encoder = Ginfinity.load(f"cuda:{local_rank}")
    owned = parallel.shard_assignment(args.shards, world, rank)
    shards = {s: synthetic.roofline_shard(s) for s in owned}          # built per rank
    everything = [shards.get(s) for s in range(args.shards)]

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    blocks = [encoder.encode_graphs_device(shards[s])[0] for s in owned]
    block = torch.cat(blocks) if blocks else torch.empty((0, 128), dtype=torch.float16,
                                                         device=f"cuda:{local_rank}")
    fence()
    t1 = time.perf_counter()
    values, indices, offsets = parallel.cross_shard_nearest(block, metric=args.metric)
    fence()
    t2 = time.perf_counter()
    total_rows = offsets[-1]
    if rank == 0:
        print(json.dumps({
            "workload": f"{args.shards} synthetic 60k-node shards over {world} GPU(s)",
            "rows_total": total_rows, "rows_this_rank": int(block.shape[0]),
            "encode_s": t1 - t0, "encode_nodes_per_s": total_rows / (t1 - t0),
            "gather_and_nearest_s": t2 - t1,
            "pair_rate_per_s": float(block.shape[0]) * total_rows * world / (t2 - t1),
            "metric": args.metric, "sample": [float(values[0]), int(indices[0])]}))
    del everything
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
