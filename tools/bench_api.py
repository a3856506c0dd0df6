"""API-level (PCIe-inclusive) throughput of BASELINE configs[1]: encode_graphs on
tests/golden/rouskin_sample_6k.tsv built into one shard, numpy in -> numpy out."""
from __future__ import annotations

import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from ginfinity_amd import (Ginfinity, GraphBuilder, load_graph_shard, read_rna_table,  # noqa: E402
                           save_graph_shard)


def d2h_probe() -> None:
    """``--d2h-probe``: what the copy engine does with the 15 result copies of config 2 by
    themselves — 15 chunks of one device block into one page-locked host block on one copy
    stream, idle, beside compute, and beside 240 MB of 4-MB uploads; then the block as ONE copy.
    (Round 3's one-off probes, folded in here: profiles/README.md, "D2H".)"""
    import torch
    dev = torch.device("cuda:0")
    rows, width, chunks = 897588, 128, 15
    per = rows // chunks
    landing = torch.empty((rows, width), dtype=torch.float16, pin_memory=True)
    blocks = [torch.randn((per, width), device=dev).half() for _ in range(chunks)]
    stream, up_stream = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    busy = torch.randn((4096, 4096), device=dev)
    up_src = [torch.empty(4 << 20, dtype=torch.uint8, pin_memory=True) for _ in range(15)]
    up_dst = [torch.empty(4 << 20, dtype=torch.uint8, device=dev) for _ in range(15)]
    report = {}
    for label, compute_ms, uploads in (("idle", 0.0, False), ("compute_3ms", 3.0, False),
                                       ("uploads", 0.0, True), ("uploads_compute_3ms", 3.0, True)):
        runs = []
        for _ in range(3):
            torch.cuda.synchronize()
            marks, t0 = [], time.perf_counter()
            if uploads:
                with torch.cuda.stream(up_stream):
                    for _ in range(4):
                        for a, b in zip(up_src, up_dst):
                            b.copy_(a, non_blocking=True)
            with torch.cuda.stream(stream):
                for i, block in enumerate(blocks):
                    a = torch.cuda.Event(enable_timing=True)
                    a.record(stream)
                    landing[i * per:(i + 1) * per].copy_(block, non_blocking=True)
                    b = torch.cuda.Event(enable_timing=True)
                    b.record(stream)
                    marks.append((a, b))
            until = time.perf_counter() + compute_ms * 1e-3
            while time.perf_counter() < until:
                torch.mm(busy, busy)
            stream.synchronize()
            total = time.perf_counter() - t0
            torch.cuda.synchronize()
            runs.append({"total_ms": total * 1e3, "gbytes_per_s": rows * 256 / total / 1e9,
                         "per_copy_ms": [round(a.elapsed_time(b), 3) for a, b in marks]})
        report[label] = runs
    one = torch.empty((rows, width), dtype=torch.float16, device=dev)
    torch.cuda.synchronize()
    t = time.perf_counter()
    landing.copy_(one, non_blocking=True)
    torch.cuda.synchronize()
    t = time.perf_counter() - t
    report["one_copy"] = {"ms": t * 1e3, "gbytes_per_s": rows * 256 / t / 1e9}
    print(json.dumps({"workload": "15 x 15.4 MB device -> page-locked host, one copy stream",
                      **report}))


def main() -> None:
    if "--d2h-probe" in sys.argv[1:]:
        return d2h_probe()
    t0 = time.perf_counter()
    records = read_rna_table(ROOT / "tests" / "golden" / "rouskin_sample_6k.tsv")
    t1 = time.perf_counter()
    shard = GraphBuilder().build_shard(records)
    t2 = time.perf_counter()
    encoder = Ginfinity.load("cuda", allow_nondeterministic_cuda=True)
    encoder.encode_graphs(shard.slice(0, 50))            # warm
    best_all = []
    outputs = None
    for _ in range(6):
        outputs = None                                   # (freeing 230 MB of results is the
        a = time.perf_counter()                          # caller's time, not the call's)
        outputs = encoder.encode_graphs(shard)
        best_all.append(time.perf_counter() - a)
    best = min(best_all)
    nodes = shard.node_count
    # pinned_outputs=False: pageable result memory through the staging ring (round 2's path)
    encoder.pinned_outputs = False
    encoder.encode_graphs(shard.slice(0, 50))
    pageable_all = []
    outputs_pageable = None
    for _ in range(6):
        outputs_pageable = None
        a = time.perf_counter()
        outputs_pageable = encoder.encode_graphs(shard)
        pageable_all.append(time.perf_counter() - a)
    assert all(x.tobytes() == y.tobytes() for x, y in zip(outputs, outputs_pageable))
    outputs_pageable = None
    encoder.pinned_outputs = None
    encoder.encode_many(records[:50])
    many = 1e9
    outputs_many = None
    for _ in range(3):
        outputs_many = None
        a = time.perf_counter()
        outputs_many = encoder.encode_many(records)
        many = min(many, time.perf_counter() - a)
    assert all(x.tobytes() == y.tobytes() for x, y in zip(outputs, outputs_many))
    # the `embed-graphs` path: shard file (mapped, not read: shard_io._map_tensors) -> embeddings
    import tempfile
    with tempfile.TemporaryDirectory() as scratch:
        tensor_path, _ = save_graph_shard(shard, Path(scratch) / "rouskin.safetensors")
        load_graph_shard(tensor_path)                    # page cache warm, as after a build step
        from_file = 1e9
        load_only = 1e9
        outputs_file = None
        for _ in range(3):
            outputs_file = None
            a = time.perf_counter()
            loaded = load_graph_shard(tensor_path, expected_spec=encoder.graph_spec)
            b = time.perf_counter()
            outputs_file = encoder.encode_graphs(loaded)
            from_file = min(from_file, time.perf_counter() - a)
            load_only = min(load_only, b - a)
        assert all(x.tobytes() == y.tobytes() for x, y in zip(outputs, outputs_file))
    # device-resident leg (Ginfinity.stage_shards + encode_staged: what parallel.encode_owned_shards
    # and bench.py --workload cross-shard run): inputs in HBM, embeddings left there; the
    # micro-batches in groups of MICROBATCH_GROUP per launch sequence vs one by one
    import torch
    from ginfinity_amd import api
    staged, _counts = encoder.stage_shards(shard)
    device_leg = {}
    for group in (api.MICROBATCH_GROUP, 1):
        api.MICROBATCH_GROUP, keep = group, api.MICROBATCH_GROUP
        block = encoder.encode_staged(staged)
        torch.cuda.synchronize()
        took = 1e9
        for _ in range(5):
            a = time.perf_counter()
            encoder.encode_staged(staged, out=block)
            torch.cuda.synchronize()
            took = min(took, time.perf_counter() - a)
        api.MICROBATCH_GROUP = keep
        device_leg[f"group_of_{group}"] = {"seconds": took, "nodes_per_s": nodes / took}
    print(json.dumps({
        "workload": "encode_graphs(rouskin shard) numpy->numpy, fp16, default limits",
        "encode_staged_device_resident": device_leg,
        "records": shard.record_count, "nodes": nodes, "edges": shard.edge_count,
        "read_table_s": t1 - t0, "build_shard_s": t2 - t1, "encode_graphs_s": best,
        "encode_many_s_device_built_graphs": many, "nodes_per_s_encode_many": nodes / many,
        "load_graph_shard_s_mapped": load_only, "load_and_encode_graphs_s": from_file,
        "nodes_per_s_from_shard_file": nodes / from_file,
        "nodes_per_s_api": nodes / best,
        "d2h_gbytes_per_s_encode_graphs": nodes * 256 / best / 1e9,
        "encode_graphs_s_all": best_all,
        "result_memory": "page-locked block, written by the device (the default)",
        "encode_graphs_s_pageable_results_all": pageable_all,
        "encode_graphs_s_pageable_results": min(pageable_all),
        "d2h_gbytes_per_s_pageable_results": nodes * 256 / min(pageable_all) / 1e9,
        "gpu_max_hw_queues": __import__("os").environ.get("GPU_MAX_HW_QUEUES"),
        "h2d_d2h_bytes": int(shard.node_features.nbytes + shard.edge_index.nbytes
                             + shard.edge_types.nbytes + nodes * 256),
        "outputs": len(outputs), "dtype": str(outputs[0].dtype)}))


if __name__ == "__main__":
    main()
