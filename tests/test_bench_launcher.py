"""`python bench.py --gpus N` starts its own ranks: the parent counts the
visible devices without initialising HIP and refuses a world it cannot give one GPU per rank —
it must never report a smaller world as if it were the requested one."""
from __future__ import annotations

import subprocess
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]


def test_launcher_refuses_more_ranks_than_devices():
    wanted = torch.cuda.device_count() + 1
    done = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(max(wanted, 2)),
                           "--steps", "2", "--warmup", "1"], cwd=ROOT, capture_output=True,
                          text=True, timeout=300)
    assert done.returncode == 3, (done.returncode, done.stderr[-500:])
    assert "HIP device(s) visible" in done.stderr
    assert not [line for line in done.stdout.splitlines() if line.startswith("{")]


def test_mismatched_world_is_rejected():
    import os
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    done = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], cwd=ROOT,
                          env=env, capture_output=True, text=True, timeout=300)
    assert done.returncode != 0 and "WORLD_SIZE=1" in done.stderr
