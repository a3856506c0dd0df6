"""Pin the oracle (oracle/gine_numpy.py) against fixtures generated from the
genuine reference (tests/golden/make_golden.py).  CPU only."""
from __future__ import annotations

import numpy as np
import pytest

from oracle import gine_numpy as G

F16_TOL = 1e-3      # north_star tolerance, fp16 model, on unit-norm rows
F32_TOL = 1e-6      # north_star tolerance, fp32 model

# stages whose oracle restatement came out bit-identical to the reference on
# every fixture (input Linear, message/aggregate, first Linear, BatchNorm).
# The K=256 Linear depends on the BLAS accumulation order (a 1-ulp flip on
# ~1e-3 of elements for 8-row inputs); LayerNorm carries the ≈7e-5 one-ulp
# residue documented in the oracle.
EXACT_STAGES_LAYER0 = ("h0", "l0.z", "l0.u", "l0.v")


def _first_records(shard, count):
    return shard.slice(0, count)


def test_example8_stages_and_outputs(golden, oracle_weights):
    g = golden("example8.npz")
    trace = {}
    out = G.encode(oracle_weights, g["node_features"], g["edge_index"],
                   g["edge_types"], trace=trace)
    for name in EXACT_STAGES_LAYER0:
        np.testing.assert_array_equal(trace[name], g[f"stage.{name}"], err_msg=name)
    assert np.mean(trace["l0.w"] != g["stage.l0.w"]) < 1e-2
    for name, value in trace.items():
        if name.endswith("table") or name.endswith("agg"):
            continue
        diff = np.abs(value.astype(np.float32) - g[f"stage.{name}"].astype(np.float32))
        assert diff.max() <= 2e-2, name          # a few fp16 ulps at most
    assert np.abs(out.astype(np.float64)
                  - g["out.m16.float16"].astype(np.float64)).max() <= F16_TOL
    for dtype in ("float32", "float64"):
        got = G.encode(oracle_weights, g["node_features"], g["edge_index"],
                       g["edge_types"], embedding_dtype=dtype)
        assert got.dtype == np.dtype(dtype)
        assert np.abs(got - g[f"out.m16.{dtype}"]).max() <= F16_TOL


def test_example8_full_precision(golden, oracle_weights):
    g = golden("example8.npz")
    for dtype in ("float16", "float32", "float64"):
        got = G.encode(oracle_weights, g["node_features"], g["edge_index"],
                       g["edge_types"], full_precision=True,
                       embedding_dtype=dtype)
        tol = F32_TOL if dtype != "float16" else 5e-4   # fp16 output rounding
        assert np.abs(got.astype(np.float64)
                      - g[f"out.m32.{dtype}"].astype(np.float64)).max() <= tol


def test_rouskin64_stage_agreement(golden, oracle_weights, rouskin_shard):
    g = golden("rouskin64.npz")
    part = _first_records(rouskin_shard, 4)
    trace = {}
    G.forward_f16(oracle_weights.half(), part.node_features, part.edge_index,
                  part.edge_types, trace)
    for name in EXACT_STAGES_LAYER0:
        np.testing.assert_array_equal(trace[name], g[f"stage.{name}"], err_msg=name)
    # BatchNorm restatement is exact given the reference's own input
    w16 = oracle_weights.half()
    for l in range(4):
        p = f"convs.{l}.mlp.1."
        v = np.maximum(G._batchnorm_f16(
            g[f"stage.l{l}.u"], w16[p + "weight"], w16[p + "bias"],
            w16[p + "running_mean"], w16[p + "running_var"]), np.float16(0))
        np.testing.assert_array_equal(v, g[f"stage.l{l}.v"])
        y = G._layernorm_f16(g[f"stage.l{l}.w"], w16[f"norms.{l}.weight"],
                             w16[f"norms.{l}.bias"])
        mismatch = np.mean(y != g[f"stage.l{l}.y"])
        assert mismatch < 5e-4, (l, mismatch)


def test_rouskin64_outputs_within_tolerance(golden, oracle_weights, rouskin_shard):
    g = golden("rouskin64.npz")
    part = _first_records(rouskin_shard, 64)
    out = G.encode(oracle_weights, part.node_features, part.edge_index,
                   part.edge_types)
    want = g["out.m16"]
    assert out.shape == want.shape
    diff = np.abs(out.astype(np.float64) - want.astype(np.float64))
    assert diff.max() <= F16_TOL
    assert np.mean(out == want) > 0.9          # most elements bit-identical
    part16 = _first_records(rouskin_shard, 16)
    out32 = G.encode(oracle_weights, part16.node_features, part16.edge_index,
                     part16.edge_types, full_precision=True,
                     embedding_dtype=np.float32)
    assert np.abs(out32.astype(np.float64)
                  - g["out.m32.float32"].astype(np.float64)).max() <= F32_TOL


def test_float64_truth_matches_reference_fp32_model(golden, oracle_weights, rouskin_shard):
    """The fp64 evaluation of the module (what the HIP fp32 path approximates)
    sits within the reference's own fp32 error of the fp32 golden."""
    g = golden("rouskin64.npz")
    part = _first_records(rouskin_shard, 16)
    raw = G.forward_f32(oracle_weights, part.node_features, part.edge_index,
                        part.edge_types, dtype=np.float64)
    e = raw / np.maximum(np.linalg.norm(raw, axis=1, keepdims=True), 1e-12)
    assert np.abs(e - g["out.m32.float32"]).max() <= F32_TOL


def test_degenerate_graphs(golden, oracle_weights):
    g = golden("degenerate.npz")
    for name in ("A", "AC", "GC"):
        out = G.encode(oracle_weights, g[f"{name}.node_features"],
                       g[f"{name}.edge_index"], g[f"{name}.edge_types"])
        assert np.abs(out.astype(np.float64)
                      - g[f"{name}.out.m16"].astype(np.float64)).max() <= F16_TOL


def test_sliced_graphs_drop_context_rows(golden, oracle_weights):
    g = golden("sliced.npz")
    for hops in (1, 2, 3):
        out = G.encode(oracle_weights, g[f"hops{hops}.node_features"],
                       g[f"hops{hops}.edge_index"], g[f"hops{hops}.edge_types"])
        core = g[f"hops{hops}.node_roles"] == 0
        want = g[f"hops{hops}.out.m16"]
        assert want.shape == (7, 128)
        assert np.abs(out[core].astype(np.float64)
                      - want.astype(np.float64)).max() <= F16_TOL


def test_arbitrary_interchange_shard(golden, oracle_weights):
    """Edge types 6-9, hubs (in-degree > 40), self loops, context roles."""
    from ginfinity_amd import synthetic
    shard = synthetic.arbitrary_shard(0)
    g = golden("arbitrary.npz")
    out = G.encode(oracle_weights, shard.node_features, shard.edge_index,
                   shard.edge_types)
    core = out[shard.node_roles == 0][::int(g["stride"])]
    assert np.abs(core.astype(np.float64)
                  - g["out.m16"].astype(np.float64)).max() <= F16_TOL


def test_synthetic_roofline_shard_rows(golden, oracle_weights):
    from ginfinity_amd import synthetic
    shard = synthetic.roofline_shard(0)
    assert (shard.node_count, shard.edge_count) == (60_000, 300_000)
    g = golden("synthetic.npz")
    out = G.encode(oracle_weights, shard.node_features, shard.edge_index,
                   shard.edge_types)
    rows = g["seed0.rows"]
    assert np.abs(out[rows].astype(np.float64)
                  - g["seed0.out.m16"].astype(np.float64)).max() <= F16_TOL


def test_microbatch_bounds_match_reference(golden, rouskin_shard):
    want = golden("integers.json")["rouskin"]["microbatches_60000_300000"]
    got = G.microbatch_bounds(rouskin_shard.lengths, rouskin_shard.edge_counts,
                              60_000, 300_000)
    assert [list(b) for b in got] == [w[:2] for w in want]
    assert len(got) == 15


def test_csr_oracle_is_a_stable_sort():
    rng = np.random.default_rng(5)
    n, e = 50, 400
    edge_index = rng.integers(0, n, size=(2, e)).astype(np.int32)
    types = rng.integers(0, 10, size=e).astype(np.uint8)
    row_ptr, col, typ = G.build_csr(edge_index, types, n)
    assert row_ptr[0] == 0 and row_ptr[-1] == e
    for node in range(n):
        mine = np.flatnonzero(edge_index[1] == node)      # COO order
        lo, hi = row_ptr[node], row_ptr[node + 1]
        np.testing.assert_array_equal(col[lo:hi], edge_index[0, mine])
        np.testing.assert_array_equal(typ[lo:hi], types[mine])


def test_distance_oracle_definitions():
    rng = np.random.default_rng(0)
    a = rng.standard_normal((5, 128))
    b = rng.standard_normal((7, 128))
    d = G.pairwise_l2(a, b)
    s = G.pairwise_cosine(a, b)
    for i in range(5):
        for j in range(7):
            assert d[i, j] == pytest.approx(np.linalg.norm(a[i] - b[j]), rel=1e-12)
            assert s[i, j] == pytest.approx(
                a[i] @ b[j] / np.linalg.norm(a[i]) / np.linalg.norm(b[j]), rel=1e-12)


def test_torch_port_matches_reference_goldens(golden, checkpoint, rouskin_shard):
    """oracle/gine_torch.py issues the reference's own aten op sequence: on the
    machine that recorded the goldens it is bit-identical; elsewhere (other CPU
    kernels) it must still be inside the fp16 tolerance."""
    from oracle import gine_torch as T
    g = golden("rouskin64.npz")
    part = rouskin_shard.slice(0, 64)
    params = T.prepare(checkpoint.state)
    out = T.encode(params, part.node_features, part.edge_index, part.edge_types)
    assert np.abs(out.astype(np.float64) - g["out.m16"].astype(np.float64)).max() <= F16_TOL
    assert np.mean(out == g["out.m16"]) > 0.9
    params32 = T.prepare(checkpoint.state, full_precision=True)
    part16 = rouskin_shard.slice(0, 16)
    out32 = T.encode(params32, part16.node_features, part16.edge_index,
                     part16.edge_types, embedding_dtype=np.float32)
    assert np.abs(out32.astype(np.float64)
                  - g["out.m32.float32"].astype(np.float64)).max() <= F32_TOL


def test_layernorm_inputs_are_far_from_the_cancellation_the_device_moments_guard(
        golden, oracle_weights, rouskin_shard):
    """The layer kernels take LayerNorm's variance as E[w^2] - mean^2 in fp32 (row sums on the
    matrix cores, ginfinity_amd/csrc/gine_layer.inc: layer_norm_residual), which loses
    log2(1 + mean^2 / var) bits, and fall back to centred values for a row beyond
    mean^2 > 15 var.  With the bundled weights the inputs stay two orders of magnitude inside
    that: the reference's OWN recorded w tensors (stage.l*.w of rouskin64.npz, the input of
    nn.LayerNorm, _model.py:69) and the oracle's trace on the arbitrary-graph shard."""
    from ginfinity_amd import synthetic
    g = golden("rouskin64.npz")
    rows = [g[f"stage.l{layer}.w"].astype(np.float64) for layer in range(4)]
    shard = synthetic.arbitrary_shard(0)
    trace = {}
    G.forward_f16(oracle_weights.half(), shard.node_features, shard.edge_index,
                  shard.edge_types, trace)
    rows += [trace[f"l{layer}.w"].astype(np.float64) for layer in range(4)]
    for w in rows:
        ratio = w.mean(axis=1) ** 2 / np.maximum(w.var(axis=1), 1e-30)
        assert ratio.max() < 0.1, ratio.max()
