"""bench.py prints ONE JSON line with the keys the driver reads (the contract in the task
statement), plus the `roofline` and `cpu_baseline` objects.  Run as a child process, the way
the driver runs it."""
from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]


def _run(*flags: str) -> dict:
    env = dict(os.environ, GFY_BENCH_SETTLE_S="0.05")
    done = subprocess.run([sys.executable, str(ROOT / "bench.py"), *flags], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [line for line in done.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, done.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract():
    line = _run("--steps", "40", "--warmup", "8", "--cpu-seconds", "2", "--distance-rows", "65536")
    assert line["metric"] == "encoded nodes/sec on 60k-node/300k-edge shards"
    assert line["unit"] == "nodes/s" and line["higher_is_better"] is True
    assert (line["n_gpus"], line["steps"], line["warmup"]) == (1, 40, 8)
    assert line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["dtype"] == "f16" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["value"] > 1e8                      # an MI355X does hundreds of M nodes/s
    assert abs(line["value"] - 60000 / (line["ms_per_step"] * 1e-3)) < 1e-3 * line["value"]
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert 0.02 < roof["frac"] < 1.0
    assert roof["traffic"] is None or roof["traffic"] >= 0.9 * roof["algorithmic_bytes_per_launch"]
    # the top-level figure is ONE launch by itself (what profiles/ holds); the span with other
    # batches in flight is kept aside and can only be longer
    assert "one batch at a time" in roof["configuration"] and roof["kernel_ms"] > 0
    assert roof["isolated"]["frac"] == roof["frac"]
    assert "as timed" in roof["in_flight"]["configuration"]
    assert roof["in_flight"]["kernel_ms"] >= 0.9 * roof["kernel_ms"]
    assert roof["kernel"] == "k_gine_layer_w"
    for key in ("mfma_busy_frac", "coexec_frac"):
        assert roof[key] is None or 0.0 < roof[key] < 1.0
    assert roof["counter_source"]
    dist = line["distance"]
    assert dist["roofline"]["bound"] == "mfma" and 0.0 < dist["roofline"]["frac"] < 1.0
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "nodes/s"
    assert cpu["cores"] >= 1 and 0 < cpu["value"] < line["value"] and cpu["sample"]


def test_bench_line_one_stream_without_cpu_baseline():
    line = _run("--steps", "20", "--warmup", "4", "--streams", "1", "--no-cpu-baseline",
                "--distance-rows", "0")
    assert line["cpu_baseline"] is None and line["steps"] == 20 and line["distance"] is None
    assert line["config"]["streams_per_gpu"] == 1


def test_bench_through_the_rank_launcher_world_size_one():
    """`--spawn` = what `--gpus N>1` does: the parent starts torch.distributed.run, the one
    rank creates the RCCL group, rank 0's line is relayed."""
    line = _run("--gpus", "1", "--spawn", "--steps", "20", "--warmup", "4", "--no-cpu-baseline",
                "--distance-rows", "0")
    assert line["n_gpus"] == 1 and line["config"]["rccl_ranks"] == 1
    assert line["value"] > 1e8 and line["cpu_baseline"] is None


def test_cross_shard_workload_through_the_launcher():
    """BASELINE configs[4] in miniature, one rank under torch.distributed.run (RCCL group
    created): encode, chunked exchange + search, one JSON line."""
    line = _run("--gpus", "1", "--spawn", "--workload", "cross-shard", "--shards", "2",
                "--chunk-rows", "50000")
    assert line["n_gpus"] == 1 and line["config"]["rccl_ranks"] == 1
    assert line["config"]["rows_total"] == 120_000 and line["unit"] == "pairs/s"
    assert line["encode"]["nodes_per_s"] > 1e6 and line["exchange_and_search"]["tflops"] > 1.0


def test_two_ranks_rehearsed_on_one_gpu_over_gloo():
    """No multi-GPU box was available to this build: the control flow of N > 1 — two rank
    processes under torch.distributed.run, barrier fences, MAX-reduction of the elapsed time,
    the repeated leg with the same count on both ranks, rank 0 alone printing, the ranks
    leaving together — is run with both ranks on this box's ONE GPU over gloo
    (GFY_BENCH_BACKEND / GFY_BENCH_ONE_DEVICE in bench.py).  `value` counts both ranks' shards."""
    env = dict(os.environ, GFY_BENCH_SETTLE_S="0.05", GFY_BENCH_BACKEND="gloo",
               GFY_BENCH_ONE_DEVICE="1")
    done = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--standalone", "--local-addr", "127.0.0.1", str(ROOT / "bench.py"),   # (a free port)
         "--gpus", "2", "--steps", "40", "--warmup", "8"],
        cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [line for line in done.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, done.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 0
    assert "sharing ONE GPU" in line["config"]["rehearsal"]
    assert line["cpu_baseline"] is None and line["distance"] is None      # N = 1 only
    assert abs(line["value"] - 2 * 60000 / (line["ms_per_step"] * 1e-3)) < 1e-3 * line["value"]
    assert line["repeated"]["repeats"] >= 1 and line["value"] > 1e8
