"""Shared fixtures.  ``-m "not gpu"`` runs everywhere; ``-m gpu`` needs an MI355X.

The oracle (``oracle/``) is imported here and in the test modules only — it is
the checker, never the code under test.
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP device)")


@pytest.fixture(scope="session")
def golden():
    cache: dict[str, object] = {}

    def load(name: str):
        if name not in cache:
            path = GOLDEN / name
            cache[name] = (json.loads(path.read_text()) if name.endswith(".json")
                           else dict(np.load(path)))
        return cache[name]
    return load


@pytest.fixture(scope="session")
def checkpoint():
    from ginfinity_amd.weights import load_checkpoint
    return load_checkpoint()


@pytest.fixture(scope="session")
def oracle_weights(checkpoint):
    from oracle import gine_numpy
    return gine_numpy.Weights.from_state_dict(
        checkpoint.state, layers=checkpoint.config.layers,
        residual=checkpoint.config.residual)


@pytest.fixture(scope="session")
def rouskin_records():
    from ginfinity_amd import read_rna_table
    return read_rna_table(GOLDEN / "rouskin_sample_6k.tsv")


@pytest.fixture(scope="session")
def rouskin_shard(rouskin_records):
    from ginfinity_amd import GraphBuilder
    return GraphBuilder().build_shard(rouskin_records)


@pytest.fixture(scope="session")
def gpu_encoder():
    """fp16-model ``Ginfinity`` on cuda:0 (GPU tests only)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from ginfinity_amd import Ginfinity
    return Ginfinity.load("cuda", allow_nondeterministic_cuda=True)


@pytest.fixture(scope="session")
def gpu_encoder_fp32():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from ginfinity_amd import Ginfinity
    return Ginfinity.load("cuda", allow_nondeterministic_cuda=True, full_precision=True)
