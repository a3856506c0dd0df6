"""ORACLE (test infrastructure, not product code): numpy restatement of the
reference's GINE encode hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import this module; the product (``ginfinity_amd``) never does.

What it restates (all citations relative to /root/reference):
  * ``Ginfinity._run_graph_shard``       src/ginfinity/api.py:232-260
  * ``GINEEncoder.forward``              src/ginfinity/_model.py:65-72
  * ``GINEConv.forward``                 src/ginfinity/_model.py:39-46
  * ``model.half()`` parameter rounding  src/ginfinity/api.py:111-112

fp16 mode follows the rounding-point model of SURVEY.md §8-A: every torch-op
boundary of the reference rounds to fp16 (``R``), arithmetic inside one op is
fp32.  The one-hot ``edge_lin`` GEMM is restated as a 10-row table lookup
(bit-identical: a one-hot row selects exactly one weight column, the fp32
accumulation adds zeros).

Pinning: ``tests/test_oracle_golden.py`` checks this file against fixtures
generated from the genuine reference by ``tests/golden/make_golden.py``
(per-stage tensors of the 8-nt example and of 64 rouskin records, whole-shard
samples); see DESIGN.md §Oracle for the measured agreement.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable

import numpy as np

F16 = np.float16
F32 = np.float32
F64 = np.float64

BN_EPS = 1e-5       # nn.BatchNorm1d default   (_model.py:35)
LN_EPS = 1e-5       # nn.LayerNorm default     (_model.py:59-60)
NORM_FLOOR = 1e-12  # api.py:252

#: checkpoint tensor order of the flat weight pack (include/gfy.h documents the
#: same list).  Shapes for hidden=128, edge_dim=10, in_dim=7.
def pack_tensor_names(layers: int = 4) -> list[str]:
    names = ["input.weight", "input.bias"]
    for l in range(layers):
        p = f"convs.{l}."
        names += [p + "eps", p + "edge_lin.weight", p + "edge_lin.bias",
                  p + "mlp.0.weight", p + "mlp.0.bias",
                  p + "mlp.1.weight", p + "mlp.1.bias",
                  p + "mlp.1.running_mean", p + "mlp.1.running_var",
                  p + "mlp.4.weight", p + "mlp.4.bias",
                  f"norms.{l}.weight", f"norms.{l}.bias"]
    names += ["head.0.weight", "head.0.bias", "head.2.weight", "head.2.bias"]
    return names


def R(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even to fp16 (one reference op boundary)."""
    return np.asarray(x, dtype=F32).astype(F16)


def up(x: np.ndarray) -> np.ndarray:
    return np.asarray(x).astype(F32)


@dataclass
class Weights:
    """fp32 checkpoint tensors keyed by state-dict name; ``half()`` applies the
    reference's ``model.half()`` (api.py:111-112: parameters AND BatchNorm
    buffers are rounded to fp16)."""

    tensors: dict[str, np.ndarray]
    layers: int = 4
    residual: bool = True

    @classmethod
    def from_state_dict(cls, state: dict, *, layers: int = 4,
                        residual: bool = True) -> "Weights":
        wanted = pack_tensor_names(layers)
        return cls({k: np.ascontiguousarray(np.asarray(state[k], dtype=F32))
                    for k in wanted}, layers, residual)

    def half(self) -> "Weights":
        return Weights({k: v.astype(F16).astype(F32)
                        for k, v in self.tensors.items()},
                       self.layers, self.residual)

    def __getitem__(self, key: str) -> np.ndarray:
        return self.tensors[key]


def _segment_sum_f32(messages: np.ndarray, destination: np.ndarray,
                     nodes: int) -> np.ndarray:
    """fp32 accumulation in edge order per destination row (the reference's
    ``index_add_`` on CPU accumulates fp16 sources in fp32 and rounds once:
    SURVEY §8-A, a4)."""
    out = np.zeros((nodes, messages.shape[1]), dtype=F32)
    np.add.at(out, destination, messages.astype(F32))
    return out


def _linear_f16(x16: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """R(x·Wᵀ + b) with fp32 accumulation (aten::addmm on Half, CPU)."""
    return R(up(x16) @ w.T + b)


def fma32(a, b, c) -> np.ndarray:
    """Single-rounded fp32 fused multiply-add.  a·b is exact in float64
    (24+24 ≤ 53 bits); the one float64 add before the final rounding makes a
    double-rounding slip possible only with probability ~2⁻²⁹ per element."""
    return (np.asarray(a, F64) * np.asarray(b, F64)
            + np.asarray(c, F64)).astype(F32)


def bn_affine(gamma, beta, mean, var) -> tuple[np.ndarray, np.ndarray]:
    """Per-channel (alpha, shift) of eval-mode BatchNorm, each step rounded to
    fp32 as the torch CPU kernel does: invstd = 1/sqrt(var+eps);
    alpha = invstd·gamma; shift = beta − mean·alpha (two roundings)."""
    invstd = (F32(1) / np.sqrt(var + F32(BN_EPS))).astype(F32)
    alpha = (invstd * gamma).astype(F32)
    shift = (beta - (mean * alpha).astype(F32)).astype(F32)
    return alpha, shift


def _batchnorm_f16(u16: np.ndarray, gamma, beta, mean, var) -> np.ndarray:
    """BatchNorm1d eval on Half input, fp32 internals (_model.py:35):
    R(fma(u, alpha, shift)).  Bit-identical to the reference on all four
    layers of the 64-record fixture (tests/test_oracle_golden.py)."""
    alpha, shift = bn_affine(gamma, beta, mean, var)
    return R(fma32(up(u16), alpha[None, :], shift[None, :]))


def _layernorm_f16(w16: np.ndarray, gamma, beta) -> np.ndarray:
    """LayerNorm(128) on Half input, fp32 internals, biased variance
    (_model.py:59-60,69): y = fma(fma(x, rstd, −rstd·mean), γ, β).

    Mean/variance are taken exactly (float64) and rounded to fp32; the torch
    CPU kernel's vectorised Welford differs from that in the last fp32 bit on
    some rows, which shows as ≈7e-5 of elements one fp16 ulp off."""
    x = up(w16)
    mean64 = x.mean(axis=1, dtype=F64)
    var = ((x.astype(F64) - mean64[:, None]) ** 2).mean(axis=1).astype(F32)
    mean = mean64.astype(F32)
    rstd = (F32(1) / np.sqrt(var + F32(LN_EPS))).astype(F32)
    offset = (-rstd * mean).astype(F32)
    return R(fma32(fma32(x, rstd[:, None], offset[:, None]),
                   gamma[None, :], beta[None, :]))


def forward_f16(weights: Weights, node_features: np.ndarray,
                edge_index: np.ndarray, edge_types: np.ndarray,
                trace: dict | None = None) -> np.ndarray:
    """fp16-mode forward → raw head output ``o`` (N,128) float16.

    ``weights`` must already be ``half()``-rounded."""
    keep: Callable[[str, np.ndarray], None] = (
        (lambda k, v: trace.__setitem__(k, v)) if trace is not None
        else (lambda k, v: None))
    source = np.asarray(edge_index[0], dtype=np.int64)
    destination = np.asarray(edge_index[1], dtype=np.int64)
    types = np.asarray(edge_types, dtype=np.int64)
    nodes = node_features.shape[0]

    x16 = R(node_features)                                   # api.py:237-238
    h = _linear_f16(x16, weights["input.weight"], weights["input.bias"])
    keep("h0", h)
    for l in range(weights.layers):
        p = f"convs.{l}."
        table = R(weights[p + "edge_lin.weight"].T
                  + weights[p + "edge_lin.bias"][None, :])   # (10,128)
        message = np.maximum(R(up(h)[source] + up(table)[types]), F16(0))
        aggregate = R(_segment_sum_f32(message, destination, nodes))
        scale = R(F32(1) + up(R(weights[p + "eps"])))        # fp16 scalar
        z = R(up(R(up(scale) * up(h))) + up(aggregate))      # _model.py:46
        u = _linear_f16(z, weights[p + "mlp.0.weight"], weights[p + "mlp.0.bias"])
        v = np.maximum(_batchnorm_f16(
            u, weights[p + "mlp.1.weight"], weights[p + "mlp.1.bias"],
            weights[p + "mlp.1.running_mean"],
            weights[p + "mlp.1.running_var"]), F16(0))
        w = _linear_f16(v, weights[p + "mlp.4.weight"], weights[p + "mlp.4.bias"])
        y = _layernorm_f16(w, weights[f"norms.{l}.weight"],
                           weights[f"norms.{l}.bias"])
        h_next = R(up(h) + up(y)) if weights.residual else y  # _model.py:71
        for key, value in (("table", table), ("agg", aggregate), ("z", z),
                           ("u", u), ("v", v), ("w", w), ("y", y),
                           ("h", h_next)):
            keep(f"l{l}.{key}", value)
        h = h_next
    t = np.maximum(_linear_f16(h, weights["head.0.weight"],
                               weights["head.0.bias"]), F16(0))
    o = _linear_f16(t, weights["head.2.weight"], weights["head.2.bias"])
    keep("head.t", t)
    keep("o", o)
    return o


def forward_f32(weights: Weights, node_features: np.ndarray,
                edge_index: np.ndarray, edge_types: np.ndarray,
                *, dtype=F32) -> np.ndarray:
    """``full_precision`` forward (fp32 parameters, fp32 activations).

    ``dtype=float64`` evaluates the same module in double precision: the
    truth both the reference's fp32 run and the HIP fp32 path are compared
    against (reference fp32 vs fp64 max 4.96e-7)."""
    c = lambda a: np.asarray(a, dtype=dtype)
    source = np.asarray(edge_index[0], dtype=np.int64)
    destination = np.asarray(edge_index[1], dtype=np.int64)
    types = np.asarray(edge_types, dtype=np.int64)
    nodes = node_features.shape[0]
    lin = lambda x, wk, bk: x @ c(weights[wk]).T + c(weights[bk])
    h = lin(c(node_features), "input.weight", "input.bias")
    for l in range(weights.layers):
        p = f"convs.{l}."
        table = c(weights[p + "edge_lin.weight"]).T + c(weights[p + "edge_lin.bias"])
        message = np.maximum(h[source] + table[types], 0)
        aggregate = np.zeros((nodes, h.shape[1]), dtype=dtype)
        np.add.at(aggregate, destination, message)
        z = (dtype(1) + c(weights[p + "eps"])[0]) * h + aggregate
        u = lin(z, p + "mlp.0.weight", p + "mlp.0.bias")
        invstd = 1 / np.sqrt(c(weights[p + "mlp.1.running_var"]) + dtype(BN_EPS))
        v = np.maximum((u - c(weights[p + "mlp.1.running_mean"])) * invstd
                       * c(weights[p + "mlp.1.weight"])
                       + c(weights[p + "mlp.1.bias"]), 0)
        w = lin(v, p + "mlp.4.weight", p + "mlp.4.bias")
        mean = w.mean(axis=1, keepdims=True)
        var = ((w - mean) ** 2).mean(axis=1, keepdims=True)
        y = ((w - mean) / np.sqrt(var + dtype(LN_EPS))
             * c(weights[f"norms.{l}.weight"]) + c(weights[f"norms.{l}.bias"]))
        h = h + y if weights.residual else y
    t = np.maximum(lin(h, "head.0.weight", "head.0.bias"), 0)
    return lin(t, "head.2.weight", "head.2.bias")


def normalise(raw: np.ndarray, embedding_dtype=F16) -> np.ndarray:
    """float64 L2 normalise + single rounding to the output dtype
    (api.py:250-252,258-259)."""
    e = np.asarray(raw).astype(F32).astype(F64)
    norms = np.linalg.norm(e, axis=1, keepdims=True)
    return (e / np.maximum(norms, NORM_FLOOR)).astype(embedding_dtype)


def encode(weights32: Weights, node_features, edge_index, edge_types, *,
           full_precision: bool = False, embedding_dtype=F16,
           trace: dict | None = None) -> np.ndarray:
    """Whole ``_run_graph_shard`` numerics for one micro-batch: (N,128) rows in
    ``embedding_dtype`` (core-row filtering is the caller's, api.py:253-260)."""
    if full_precision:
        raw = forward_f32(weights32, node_features, edge_index, edge_types)
    else:
        raw = forward_f16(weights32.half(), node_features, edge_index,
                          edge_types, trace)
    return normalise(raw, embedding_dtype)


# --------------------------------------------------------------------------
# integer side of the path
# --------------------------------------------------------------------------

def build_csr(edge_index: np.ndarray, edge_types: np.ndarray, nodes: int):
    """Destination-major CSR with edges of a row kept in COO order (stable):
    row_ptr i32 (N+1), col i32 (E) = source ids, typ u8 (E).  The bit-exact
    target of ``gfy_build_csr``."""
    destination = np.asarray(edge_index[1], dtype=np.int64)
    order = np.argsort(destination, kind="stable")
    row_ptr = np.zeros(nodes + 1, dtype=np.int32)
    np.cumsum(np.bincount(destination, minlength=nodes), out=row_ptr[1:])
    return (row_ptr, np.asarray(edge_index[0])[order].astype(np.int32),
            np.asarray(edge_types)[order].astype(np.uint8))


def microbatch_bounds(lengths, edge_counts, max_batch_nodes: int,
                      max_batch_edges: int) -> list[tuple[int, int]]:
    """Greedy contiguous packing of ``encode_graphs`` (api.py:211-230) as
    (start, stop) record ranges."""
    bounds, start, count = [], 0, len(lengths)
    while start < count:
        stop, nodes, edges = start, 0, 0
        while stop < count:
            if stop > start and (nodes + lengths[stop] > max_batch_nodes
                                 or edges + edge_counts[stop] > max_batch_edges):
                break
            nodes += lengths[stop]
            edges += edge_counts[stop]
            stop += 1
        bounds.append((start, stop))
        start = stop
    return bounds


# --------------------------------------------------------------------------
# distance (a9): defined here, the reference has no implementation
# (parity unpinned — SURVEY §8c)
# --------------------------------------------------------------------------

def pairwise_l2(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a64, b64 = np.asarray(a, F64), np.asarray(b, F64)
    d2 = ((a64 ** 2).sum(1)[:, None] + (b64 ** 2).sum(1)[None, :]
          - 2.0 * a64 @ b64.T)
    return np.sqrt(np.maximum(d2, 0.0))


def pairwise_cosine(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a64, b64 = np.asarray(a, F64), np.asarray(b, F64)
    na = np.maximum(np.linalg.norm(a64, axis=1), NORM_FLOOR)
    nb = np.maximum(np.linalg.norm(b64, axis=1), NORM_FLOOR)
    return (a64 @ b64.T) / (na[:, None] * nb[None, :])
