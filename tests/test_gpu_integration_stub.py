"""INTEGRATION.md's reference-side binding (integration/_hip.py, the file a maintainer of the
reference would add) is executed as written: weight_pack / create / run_graph_shard against
this repository's GraphShard must return the bytes Ginfinity.encode_graphs returns."""
from __future__ import annotations

import importlib.util
import os
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _load_stub():
    from ginfinity_amd import _native
    os.environ["GFY_LIBRARY"] = str(_native.LIBRARY_PATH)
    spec = importlib.util.spec_from_file_location("reference_hip_stub", ROOT / "integration" / "_hip.py")
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


@pytest.mark.parametrize("dtype", [np.float16, np.float32])
def test_reference_side_stub_equals_the_package(gpu_encoder, checkpoint, rouskin_shard, dtype):
    import torch
    from ginfinity_amd import synthetic
    stub = _load_stub()
    state = {name: torch.from_numpy(np.asarray(value)) for name, value in checkpoint.state.items()}
    pack = stub.weight_pack(state, checkpoint.config)
    handle = stub.create(pack, False, 0)
    try:
        for shard in (rouskin_shard.slice(0, 50),
                      synthetic.arbitrary_shard(3, nodes=2_000, edges=9_000)):   # context rows
            got = stub.run_graph_shard(handle, shard, dtype, torch.device("cuda", 0))
            want = gpu_encoder.encode_graphs(shard, embedding_dtype=dtype)
            assert len(got) == len(want)
            for a, b in zip(got, want):
                assert a.dtype == b.dtype and a.shape == b.shape
                np.testing.assert_array_equal(a, b)
    finally:
        stub.destroy(handle)
