"""One rank's share of BASELINE configs[4] at full size on one MI355X: 128 of the 1,024 synthetic
60k-node / 300k-edge shards (shard s -> rank s mod 8: /root/reference/docs/GRAPH_PIPELINE.md:22-24
has one process per shard and no exchange between them), encoded through the product path of a
rank (``parallel.encode_owned_shards`` -> ``Ginfinity.encode_shards_device`` ->
``gfy_encode_coo_batch``) and searched with ``parallel.cross_shard_nearest``.  No multi-GPU box
was available to this build, so the all-gather between ranks is covered by the gloo tests
(tests/test_parallel_cpu.py); everything a rank does by itself runs here at the real size.

The encode leg is pinned: shards 0 and 1 against the rows the reference itself produced
(tests/golden/synthetic.npz), a spread of the others bit for bit against the lone-micro-batch
path, the resident-input path bit for bit against the host-array one.  The nearest-row leg has
no reference symbol (parity unpinned) and is checked against the float64
definition on sampled rows."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from ginfinity_amd import parallel, synthetic

pytestmark = pytest.mark.gpu

SHARDS = 128
ROWS = SHARDS * 60_000
TOLERANCE = 4e-6      # fp16 rows, 128 products summed in fp32 (the 1M-row test: 2e-6 on random rows)


@pytest.fixture(scope="module")
def rank_share(gpu_encoder):
    shards = [synthetic.roofline_shard(seed) for seed in range(SHARDS)]
    block, owned, counts = parallel.encode_owned_shards(gpu_encoder, shards)
    torch.cuda.synchronize()
    return shards, block, owned, counts


def test_a_rank_s_128_shards_encode_to_the_reference_s_rows(gpu_encoder, golden, rank_share):
    shards, block, owned, counts = rank_share
    engine = gpu_encoder._engine
    assert owned == list(range(SHARDS)) and block.shape == (ROWS, 128)
    assert block.dtype == torch.float16 and engine.last_layer_kernel() == 4
    assert [sum(c) for c in counts] == [60_000] * SHARDS
    g = golden("synthetic.npz")
    for seed in (0, 1):                                  # rows the reference wrote
        rows = g[f"seed{seed}.rows"]
        got = block[seed * 60_000:(seed + 1) * 60_000].cpu().numpy()[rows].astype(np.float64)
        assert np.abs(got - g[f"seed{seed}.out.m16"].astype(np.float64)).max() <= 1e-3
    for seed in (2, 63, 64, 126, 127):                   # one shard by itself: other kernels
        alone, _ = gpu_encoder.encode_graphs_device(shards[seed])
        assert engine.last_layer_kernel() == 1
        piece = block[seed * 60_000:(seed + 1) * 60_000]
        assert alone.cpu().numpy().tobytes() == piece.cpu().numpy().tobytes(), seed
    # inputs resident in HBM (what bench.py times) = the same bytes; and again into the same block
    staged, staged_counts = gpu_encoder.stage_shards(shards)
    assert staged_counts == counts
    resident = gpu_encoder.encode_staged(staged)
    assert torch.equal(resident, block)
    again, _ = gpu_encoder.encode_shards_device(shards, out=resident)
    assert again.data_ptr() == resident.data_ptr() and torch.equal(again, block)
    assert bool(torch.isfinite(block).all())


def test_a_rank_s_7_68_million_rows_find_their_nearest_other_row(rank_share):
    """7,680,000 x 7,680,000 cosine pairs in 1M-row chunks (the matrix is never materialised);
    sampled rows — first, last, both sides of chunk and shard seams — against float64."""
    _, block, _, _ = rank_share
    values, indices, offsets = parallel.cross_shard_nearest(block, metric="cosine",
                                                            chunk_rows=1 << 20)
    torch.cuda.synchronize()
    assert offsets == [0, ROWS] and values.shape == (ROWS,) and indices.shape == (ROWS,)
    assert int(indices.min()) >= 0 and int(indices.max()) < ROWS
    assert not bool((indices == torch.arange(ROWS, device=indices.device)).any())
    rng = np.random.default_rng(11)
    seams = np.concatenate([np.array([k - 1, k]) for k in
                            [1 << 20, 7 << 20, 60_000, 64 * 60_000, 127 * 60_000]])
    sample = np.unique(np.concatenate([np.arange(0, 8), np.arange(ROWS - 8, ROWS), seams,
                                       rng.integers(0, ROWS, 230)]))
    picked_rows = torch.from_numpy(sample).to(block.device)
    unit = torch.nn.functional.normalize(block.double(), dim=1, eps=1e-12)   # 7.9 GB
    worst = 0.0
    for start in range(0, sample.size, 32):                              # 32 x 7.68M float64
        rows = picked_rows[start:start + 32]
        full = unit[rows] @ unit.T
        full[torch.arange(rows.numel(), device=rows.device), rows] = -np.inf        # self
        best = full.max(dim=1).values
        chosen = full[torch.arange(rows.numel(), device=rows.device), indices[rows]]
        assert float((best - chosen).abs().max()) <= TOLERANCE
        assert float((values[rows].double() - chosen).abs().max()) <= TOLERANCE
        worst = max(worst, float((values[rows].double() - best).abs().max()))
    print(f"one rank of config 5: {sample.size} sampled rows of {ROWS}, "
          f"max |value - float64| = {worst:.2e}")
