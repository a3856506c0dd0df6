"""All-pairs distance against the float64 oracle definition.
Parity is unpinned against the reference — it has no implementation — so the
oracle is the mathematical definition (oracle/gine_numpy.py)."""
from __future__ import annotations

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rows(seed, count, unit=True):
    rng = np.random.default_rng(seed)
    data = rng.standard_normal((count, 128))
    if unit:
        data /= np.linalg.norm(data, axis=1, keepdims=True)
    else:
        data *= rng.uniform(0.2, 3.0, size=(count, 1))
    return data.astype(np.float16)


@pytest.mark.parametrize("n,m", [(1, 1), (130, 257), (300, 500)])
@pytest.mark.parametrize("unit", [True, False])
def test_dense_block_matches_oracle(n, m, unit):
    from oracle import gine_numpy as G
    from ginfinity_amd import distance
    a, b = _rows(1, n, unit), _rows(2, m, unit)
    d = distance.pairwise(a, b, metric="l2").cpu().numpy().astype(np.float64)
    want = G.pairwise_l2(a, b)
    scale = ((a.astype(np.float64) ** 2).sum(1)[:, None]
             + (b.astype(np.float64) ** 2).sum(1)[None, :])
    # tolerance is stated on d² (cancellation for near-duplicates)
    assert np.abs(d ** 2 - want ** 2).max() <= 4e-6 * scale.max()
    s = distance.pairwise(a, b, metric="cosine").cpu().numpy().astype(np.float64)
    assert np.abs(s - G.pairwise_cosine(a, b)).max() <= 2e-6


def test_dense_self_distance_is_small():
    from ginfinity_amd import distance
    a = _rows(3, 200)
    d = distance.pairwise(a, metric="l2").cpu().numpy()
    assert np.abs(np.diag(d)).max() <= 2e-3          # sqrt of ~1e-7 cancellation noise
    s = distance.pairwise(a, metric="cosine").cpu().numpy()
    np.testing.assert_allclose(np.diag(s), 1.0, atol=2e-6)


@pytest.mark.parametrize("n,m", [(1000, 3000), (64, 20000), (129, 127)])
@pytest.mark.parametrize("metric", ["l2", "cosine"])
def test_nearest_matches_oracle(n, m, metric):
    from oracle import gine_numpy as G
    from ginfinity_amd import distance
    a, b = _rows(4, n), _rows(5, m)
    values, indices = distance.nearest(a, b, metric=metric)
    values, indices = values.cpu().numpy().astype(np.float64), indices.cpu().numpy()
    full = G.pairwise_l2(a, b) if metric == "l2" else G.pairwise_cosine(a, b)
    best = full.min(axis=1) if metric == "l2" else full.max(axis=1)
    picked = full[np.arange(n), indices]
    assert indices.min() >= 0 and indices.max() < m
    np.testing.assert_allclose(picked, best, atol=2e-6 if metric == "cosine" else 2e-5)
    np.testing.assert_allclose(values, picked, atol=2e-6 if metric == "cosine" else 2e-5)


def test_nearest_excludes_self_and_offset():
    from oracle import gine_numpy as G
    from ginfinity_amd import distance
    rows = _rows(6, 700)
    values, indices = distance.nearest(rows, metric="cosine", exclude_self=True)
    indices = indices.cpu().numpy()
    assert not np.any(indices == np.arange(700))
    full = G.pairwise_cosine(rows, rows)
    np.fill_diagonal(full, -np.inf)
    np.testing.assert_allclose(values.cpu().numpy(), full.max(axis=1), atol=2e-6)
    # a = rows[200:328] as a block of b = rows: skip (i, i + 200)
    block = rows[200:328]
    _, idx = distance.nearest(block, rows, metric="l2", exclude_offset=200)
    idx = idx.cpu().numpy()
    assert not np.any(idx == np.arange(128) + 200)
    _, idx_plain = distance.nearest(block, rows, metric="l2")
    np.testing.assert_array_equal(idx_plain.cpu().numpy(), np.arange(128) + 200)


def test_nearest_ties_go_to_lowest_index():
    from ginfinity_amd import distance
    base = _rows(7, 300)
    b = np.concatenate([base, base[:50], base])          # every row appears 2-3 times
    _, idx = distance.nearest(base, b, metric="cosine")
    np.testing.assert_array_equal(idx.cpu().numpy(), np.arange(300))


def test_distance_on_real_embeddings(gpu_encoder, rouskin_shard):
    from oracle import gine_numpy as G
    from ginfinity_amd import distance
    block, counts = gpu_encoder.encode_graphs_device(rouskin_shard.slice(0, 40))
    assert block.shape[0] == sum(counts)
    host = block.cpu().numpy()
    sample = host[:512]
    d = distance.pairwise(block[:512], block, metric="l2").cpu().numpy()
    np.testing.assert_allclose(d.astype(np.float64) ** 2,
                               G.pairwise_l2(sample, host) ** 2, atol=1e-5)


def test_cross_shard_nearest_world_size_one(gpu_encoder):
    """The multi-GPU search with one rank equals the single-GPU search with
    self-pairs excluded (SURVEY §8e testability)."""
    from ginfinity_amd import distance, parallel, synthetic
    shards = [synthetic.arbitrary_shard(s, nodes=1500, edges=6000, records=3)
              for s in (1, 2)]
    block, owned, counts = parallel.encode_owned_shards(gpu_encoder, shards)
    assert owned == [0, 1] and len(counts) == 2
    assert block.shape[0] == sum(sum(c) for c in counts)
    ref_values, ref_indices = distance.nearest(block, metric="cosine", exclude_self=True)
    # one chunk, and chunks that cut the block (the rows in front of / inside / behind the
    # window of every chunk are searched separately and merged on the device)
    for chunk_rows in (1 << 20, 1000, 257):
        values, indices, offsets = parallel.cross_shard_nearest(block, metric="cosine",
                                                                chunk_rows=chunk_rows)
        assert offsets == [0, block.shape[0]] and indices.dtype == torch.int64
        np.testing.assert_array_equal(indices.cpu().numpy(), ref_indices.cpu().numpy())
        np.testing.assert_array_equal(values.cpu().numpy(), ref_values.cpu().numpy())
    empty = parallel.cross_shard_nearest(block[:0], metric="l2")
    assert empty[0].shape == (0,) and empty[1].shape == (0,) and empty[2] == [0, 0]


@pytest.mark.parametrize("n,m", [(255, 129), (256, 128), (257, 385), (513, 640), (1, 513),
                                 (770, 1)])
def test_nearest_at_workgroup_and_tile_seams(n, m):
    """k_pairwise owns 256 a-rows per workgroup and sweeps b in 128-row tiles through a ring
    of four buffers: sizes one below / at / one above those seams, with and without an
    excluded (i, i + k) pair that straddles them."""
    from oracle import gine_numpy as G
    from ginfinity_amd import distance
    a, b = _rows(40 + n, n, unit=False), _rows(41 + m, m, unit=False)
    full = G.pairwise_l2(a, b)
    for offset in (None, 0, 127):
        values, indices = distance.nearest(a, b, metric="l2", exclude_offset=offset)
        values, indices = values.cpu().numpy().astype(np.float64), indices.cpu().numpy()
        masked = full.copy()
        if offset is not None:
            rows = np.arange(n)
            keep = rows + offset < m
            masked[rows[keep], rows[keep] + offset] = np.inf
        best = masked.min(axis=1)
        reachable = np.isfinite(best)          # a single b-row that is excluded: nothing left
        assert np.all(indices[~reachable] == -1)
        picked = masked[np.arange(n)[reachable], indices[reachable]]
        np.testing.assert_allclose(picked, best[reachable], rtol=0, atol=3e-3 * best.max())
        np.testing.assert_allclose(values[reachable], picked, rtol=2e-3, atol=2e-3)


def test_nearest_duplicates_across_tiles_and_sweep_chunks():
    """Ties go to the lowest index also when the equal rows sit in different b-tiles, in
    different ring buffers and — few a-rows, many b-rows — in different sweep chunks whose
    partial results are merged by k_nearest_finish."""
    from ginfinity_amd import distance
    base = _rows(11, 90)
    filler = _rows(12, 4000) * np.float16(0.5)           # never closer than an exact copy
    b = np.concatenate([filler[:700], base, filler[700:2900], base, filler[2900:], base])
    _, idx = distance.nearest(base, b, metric="l2")
    np.testing.assert_array_equal(idx.cpu().numpy(), 700 + np.arange(90))
    _, idx = distance.nearest(base, b, metric="cosine")
    np.testing.assert_array_equal(idx.cpu().numpy(), 700 + np.arange(90))


@pytest.mark.parametrize("metric", ["l2", "cosine"])
def test_config4_one_million_rows_sampled_against_the_float64_oracle(metric):
    """BASELINE configs[3] at full size: nearest other row of each of 1,000,000 unit rows
    (the 1M x 1M matrix is never materialised, 0.25 s).  4,096+ sampled a-rows — the rows on
    both sides of every 256-row workgroup seam sampled, the first and the last rows, the
    ragged last b-tile — are checked against the float64 definition (oracle.gine_numpy).
    Parity unpinned: the reference has no implementation of this step."""
    from oracle import gine_numpy as G
    from ginfinity_amd import distance, synthetic
    rows = 1_000_000
    points = synthetic.unit_rows(0, rows)
    device = torch.from_numpy(points).cuda()
    values, indices = distance.nearest(device, metric=metric, exclude_self=True)
    values, indices = values.cpu().numpy().astype(np.float64), indices.cpu().numpy()
    assert values.shape == (rows,) and indices.min() >= 0 and indices.max() < rows
    rng = np.random.default_rng(7)
    seams = np.concatenate([np.array([256 * k - 1, 256 * k]) for k in rng.integers(1, rows // 256, 96)])
    sample = np.unique(np.concatenate([
        np.arange(0, 260), np.arange(rows - 600, rows), seams,
        rng.integers(0, rows, 3_400)]))
    assert sample.size >= 4_096
    worst = 0.0
    for start in range(0, sample.size, 512):            # 512 x 1M float64 block = 4 GB
        block = sample[start:start + 512]
        full = (G.pairwise_l2(points[block], points) if metric == "l2"
                else G.pairwise_cosine(points[block], points))
        full[np.arange(block.size), block] = np.inf if metric == "l2" else -np.inf   # self
        best = full.min(axis=1) if metric == "l2" else full.max(axis=1)
        picked = full[np.arange(block.size), indices[block]]
        tolerance = 2e-6 if metric == "cosine" else 2e-5
        assert not np.any(indices[block] == block)
        np.testing.assert_allclose(picked, best, atol=tolerance)
        np.testing.assert_allclose(values[block], picked, atol=tolerance)
        worst = max(worst, float(np.abs(values[block] - best).max()))
    print(f"config 4 ({metric}): {sample.size} sampled rows, max |value - oracle| = {worst:.2e}")


def test_nearest_window_skips_every_row_s_own_copy():
    """``window_first``: b is rows [k, k + m) of a (one rank's own piece in the chunked
    cross-shard search): one call must equal the float64 definition with the pairs (k + j, j)
    removed — rows in front of the window, inside it, behind it; any alignment of the window
    against the kernel's 256-row blocks and 128-row tiles."""
    from ginfinity_amd import distance, synthetic
    from oracle import gine_numpy as G
    rows = synthetic.unit_rows(5, 3_000)
    a = torch.from_numpy(rows).cuda()
    workspace = distance.NearestWorkspace()
    for first, count in ((0, 3_000), (0, 700), (129, 1_000), (2_300, 700), (1_111, 1), (511, 513)):
        for metric in ("l2", "cosine"):
            values, indices = distance.nearest(a, a[first:first + count], metric=metric,
                                               window_first=first, workspace=workspace)
            values, indices = values.cpu().numpy(), indices.cpu().numpy()
            full = (G.pairwise_l2(rows, rows[first:first + count]) if metric == "l2"
                    else G.pairwise_cosine(rows, rows[first:first + count]))
            for j in range(count):
                full[first + j, j] = np.inf if metric == "l2" else -np.inf
            want = full.argmin(axis=1) if metric == "l2" else full.argmax(axis=1)
            best = full[np.arange(rows.shape[0]), want]
            chosen = full[np.arange(rows.shape[0]), indices]
            assert not np.any(indices == np.arange(rows.shape[0]) - first)   # never itself
            assert np.allclose(chosen, best, atol=2e-3), (first, count, metric)
            assert np.allclose(values if metric == "cosine" else values ** 2,
                               best if metric == "cosine" else best ** 2, atol=4e-3)
    with pytest.raises(ValueError, match="rows"):
        distance.nearest(a[:100], a[:200], window_first=0)
