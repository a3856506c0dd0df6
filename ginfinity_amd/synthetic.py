"""Seeded synthetic graph shards for benchmarks and parity tests.

``roofline_shard(seed)`` is the BASELINE.json config-3 workload:
exactly 60,000 nodes / 300,000 edges in 15 records of 4,000 nodes, built
directly as a ``GraphShard`` so the 4,096-nt / nesting rules of ``RNA`` do not
apply.  ``arbitrary_shard`` exercises what builder-made graphs never do:
edge types 6–9, high in-degree, isolated nodes, self loops.

The same functions generate the inputs on the golden-fixture side
(tests/golden/make_golden.py, run against the genuine reference) and on the
GPU side, so only sampled reference OUTPUT rows need to be committed.
"""
from __future__ import annotations

import numpy as np

from .graph import GraphShard
from .spec import NODE_ROLE_CORE, GraphSpec

_BASES = np.frombuffer(b"ACGU", dtype=np.uint8)


def _record_edges(rng: np.random.Generator, length: int, extra_pairs: int
                  ) -> tuple[np.ndarray, np.ndarray]:
    head = np.arange(length - 1, dtype=np.int32)
    matching = rng.permutation(length).astype(np.int32).reshape(-1, 2)
    low, high = matching.min(axis=1), matching.max(axis=1)
    near = np.arange(length - 2, dtype=np.int32)
    extra = rng.integers(0, length, size=(extra_pairs, 2)).astype(np.int32)
    source = np.concatenate([
        head, head + 1, low, high,
        np.stack((near, near + 2), axis=1).ravel(), extra[:, 0]])
    destination = np.concatenate([
        head + 1, head, high, low,
        np.stack((near + 2, near), axis=1).ravel(), extra[:, 1]])
    types = np.concatenate([
        np.full(length - 1, 0, np.uint8), np.full(length - 1, 1, np.uint8),
        np.full(low.size, 2, np.uint8), np.full(low.size, 3, np.uint8),
        np.tile(np.array([4, 5], np.uint8), length - 2),
        np.full(extra_pairs, 2, np.uint8)])
    return np.stack((source, destination)).astype(np.int32), types


def roofline_shard(seed: int = 0, *, records: int = 15, length: int = 4000,
                   extra_pairs: int = 6,
                   spec: GraphSpec | None = None) -> GraphShard:
    """Config-3 shard: ``records × length`` nodes, 5·length edges per record
    (backbone both ways, a random perfect matching both ways, skip-2 both
    ways, ``extra_pairs`` random type-2 edges); max in-degree ≈ 6."""
    if length % 2:
        raise ValueError("length must be even (perfect matching)")
    spec = spec or GraphSpec.bundled()
    rng = np.random.default_rng(seed)
    features = np.zeros((records * length, spec.node_feature_dim), np.float32)
    edge_blocks, type_blocks, sequences = [], [], []
    relative = np.arange(length, dtype=np.float32) / max(length - 1, 1)
    for r in range(records):
        bases = rng.integers(0, 4, size=length)
        rows = slice(r * length, (r + 1) * length)
        block = features[rows]
        block[np.arange(length), bases] = 1
        block[:, 4] = 1
        block[:, 5] = np.sin(np.pi * relative)
        block[:, 6] = np.cos(np.pi * relative)
        edges, types = _record_edges(rng, length, extra_pairs)
        edge_blocks.append(edges + np.int32(r * length))
        type_blocks.append(types)
        sequences.append(_BASES[bases].tobytes().decode("ascii"))
    per_record_edges = edge_blocks[0].shape[1]
    return GraphShard(
        identifiers=tuple(f"syn-{seed}-{r}" for r in range(records)),
        sequences=tuple(sequences),
        structures=tuple("." * length for _ in range(records)),
        node_features=features,
        edge_index=np.ascontiguousarray(np.concatenate(edge_blocks, axis=1)),
        edge_types=np.ascontiguousarray(np.concatenate(type_blocks)),
        node_ptr=np.arange(records + 1, dtype=np.int64) * length,
        edge_ptr=np.arange(records + 1, dtype=np.int64) * per_record_edges,
        spec=spec,
        residue_index=np.tile(np.arange(length, dtype=np.int32), records),
        node_roles=np.full(records * length, NODE_ROLE_CORE, np.uint8))


def arbitrary_shard(seed: int = 0, *, nodes: int = 10_000, edges: int = 50_000,
                    records: int = 4, hub_degree: int = 40,
                    spec: GraphSpec | None = None) -> GraphShard:
    """Interchange-format stress shard: uniformly random edges inside each
    record, all ten edge types, one hub node per record with ``hub_degree``
    extra in-edges, random (not one-hot) features, context roles on ~10 % of
    nodes.  Sequences are placeholders of the right length."""
    spec = spec or GraphSpec.bundled()
    rng = np.random.default_rng(seed)
    cuts = np.sort(rng.choice(np.arange(1, nodes), size=records - 1,
                              replace=False))
    node_ptr = np.concatenate(([0], cuts, [nodes])).astype(np.int64)
    share = np.diff(node_ptr) / nodes
    per_record = np.floor(share * (edges - records * hub_degree)).astype(int)
    per_record[-1] += (edges - records * hub_degree) - per_record.sum()
    edge_blocks, type_blocks = [], []
    for r in range(records):
        lo, hi = int(node_ptr[r]), int(node_ptr[r + 1])
        body = rng.integers(lo, hi, size=(2, per_record[r]))
        hub = np.stack((rng.integers(lo, hi, size=hub_degree),
                        np.full(hub_degree, lo + (hi - lo) // 2)))
        block = np.concatenate((body, hub), axis=1)
        block = block[:, rng.permutation(block.shape[1])]
        edge_blocks.append(block.astype(np.int32))
        type_blocks.append(rng.integers(0, spec.edge_dim, size=block.shape[1]
                                        ).astype(np.uint8))
    edge_ptr = np.zeros(records + 1, np.int64)
    np.cumsum([b.shape[1] for b in edge_blocks], out=edge_ptr[1:])
    features = rng.standard_normal(
        (nodes, spec.node_feature_dim)).astype(np.float32)
    roles = (rng.random(nodes) < 0.1).astype(np.uint8)
    roles[node_ptr[:-1]] = NODE_ROLE_CORE            # every record keeps a core
    sizes = np.diff(node_ptr)
    residue = (np.arange(nodes, dtype=np.int64)
               - np.repeat(node_ptr[:-1], sizes)).astype(np.int32)
    return GraphShard(
        identifiers=tuple(f"arb-{seed}-{r}" for r in range(records)),
        sequences=tuple("A" * int(s) for s in sizes),
        structures=tuple("." * int(s) for s in sizes),
        node_features=features,
        edge_index=np.ascontiguousarray(np.concatenate(edge_blocks, axis=1)),
        edge_types=np.ascontiguousarray(np.concatenate(type_blocks)),
        node_ptr=node_ptr, edge_ptr=edge_ptr, spec=spec,
        residue_index=residue, node_roles=roles)


def unit_rows(seed: int, rows: int, dim: int = 128,
              dtype=np.float16) -> np.ndarray:
    """Config-4 input: seeded unit-norm rows (normalised in float64, one
    rounding to ``dtype``)."""
    rng = np.random.default_rng(seed)
    data = rng.standard_normal((rows, dim))
    data /= np.linalg.norm(data, axis=1, keepdims=True)
    return data.astype(dtype)
