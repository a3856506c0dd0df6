"""A rank's share of BASELINE configs[4] from HOST arrays: ``Ginfinity.encode_shards_device`` on
``--shards`` synthetic 60,000-node shards (numpy in, one device block out; packers ->
page-locked staging ring -> H2D under the compute), the call timed ``--repeats`` times.  The
environment selects what is compared on one box (one process per setting):
GFY_PACKERS (packer threads), GFY_PACK_STREAM (0: memcpy, else streaming stores in
gfy_pack_microbatch), GFY_STAGING_SLOTS.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ginfinity_amd import Ginfinity, synthetic  # noqa: E402


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--shards", type=int, default=128)
    parser.add_argument("--repeats", type=int, default=7)
    parser.add_argument("--skip", choices=("none", "pack", "copy", "both"), default="none",
                        help="timing diagnostics (every shard has the same sizes, so a slot's "
                             "earlier content is a valid stand-in): pack = the packers return the "
                             "warm call's offsets without writing; copy = a group's device block "
                             "is the one its slot was sent to in the warm call, no H2D copy")
    parser.add_argument("--lanes", type=int, default=0, help="api.HOST_FEED_LANES (0: as shipped)")
    parser.add_argument("--profile", action="store_true",
                        help="cProfile of the calling thread over the timed calls (stderr)")
    args = parser.parse_args()
    from ginfinity_amd import api
    if args.lanes:
        api.HOST_FEED_LANES = args.lanes
    encoder = Ginfinity.load("cuda:0", allow_nondeterministic_cuda=True)
    distinct = [synthetic.roofline_shard(s) for s in range(min(args.shards, 16))]
    # distinct arrays per shard (a copy, not a view): the packers read 128 different buffers
    shards = [distinct[s] if s < len(distinct) else _copy(distinct[s % len(distinct)])
              for s in range(args.shards)]
    nodes = sum(int(shard.node_features.shape[0]) for shard in shards)
    block, _counts = encoder.encode_shards_device(shards)          # warm: ring, workspace
    torch.cuda.synchronize()
    if args.skip != "none":
        _skip(args.skip, encoder, shards)
    seconds = []
    if args.profile:
        import cProfile
        import pstats
        profile = cProfile.Profile()
        profile.enable()
        for _ in range(args.repeats):
            encoder.encode_shards_device(shards, out=block)
        profile.disable()
        torch.cuda.synchronize()
        pstats.Stats(profile, stream=sys.stderr).sort_stats("tottime").print_stats(22)
    for _ in range(args.repeats):
        t0 = time.perf_counter()
        encoder.encode_shards_device(shards, out=block)
        torch.cuda.synchronize()
        seconds.append(time.perf_counter() - t0)
    best = min(seconds)
    print(json.dumps({
        "workload": f"encode_shards_device, {args.shards} host shards, {nodes} nodes",
        "packers": os.environ.get("GFY_PACKERS", "default"),
        "pack_stream": os.environ.get("GFY_PACK_STREAM", "default"),
        "staging_slots": os.environ.get("GFY_STAGING_SLOTS", "default"), "skip": args.skip,
        "lanes": api.HOST_FEED_LANES,
        "seconds_all": [round(s, 5) for s in seconds],
        "seconds_best": best, "seconds_median": sorted(seconds)[len(seconds) // 2],
        "nodes_per_s_best": nodes / best,
        "nodes_per_s_median": nodes / sorted(seconds)[len(seconds) // 2],
        "checksum": float(block[::4001].float().sum().item())}))


def _skip(what: str, encoder, shards) -> None:
    from ginfinity_amd import api
    if what in ("pack", "both"):
        real_pack, packed = api.Ginfinity._pack_microbatch_at, {}

        def pack(uploader, slot, base, shard, start, stop):
            key = (base, start, stop)                     # equal-sized shards: the same answer
            if key not in packed:
                packed[key] = real_pack(uploader, slot, base, shard, start, stop)
            return packed[key]
        api.Ginfinity._pack_microbatch_at = staticmethod(pack)
    if what in ("copy", "both"):
        real_send, sent = api._Uploader.send_group, {}

        def send(self, slot, total, copies, consumer):
            if slot not in sent:
                sent[slot] = real_send(self, slot, total, copies, consumer)
            return sent[slot]
        api._Uploader.send_group = send
    block, _ = encoder.encode_shards_device(shards)       # fills the memo tables
    torch.cuda.synchronize()


def _copy(shard):
    import copy
    import numpy as np
    twin = copy.copy(shard)
    for name in ("node_features", "edge_index", "edge_types", "node_roles"):
        object.__setattr__(twin, name, np.array(getattr(shard, name), copy=True))
    return twin


if __name__ == "__main__":
    main()
