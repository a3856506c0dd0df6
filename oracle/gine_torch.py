"""ORACLE (test infrastructure, not product code): the reference's encode path
restated with the same stock aten ops the reference executes on the CPU.

Used for two things only: (1) ``bench.py``'s ``cpu_baseline`` leg — it runs at
the speed of the reference's own CPU encode because it issues the same
operator sequence (addmm, index_select, add, relu, index_add_, batch_norm,
layer_norm); (2) cross-checking ``oracle/gine_numpy.py``.
The product (``ginfinity_amd``) never imports it.

Follows /root/reference/src/ginfinity/_model.py:39-46,65-72 (operator order)
and src/ginfinity/api.py:232-260 (casts, float64 normalise, core filter) as a
pure function over a flat state dict — there is no ``nn.Module``.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def prepare(state: dict[str, np.ndarray], *, full_precision: bool = False
            ) -> dict[str, torch.Tensor]:
    """fp32 checkpoint tensors → torch tensors in the model dtype
    (``model.half()`` rounds parameters and BatchNorm buffers: api.py:111-112)."""
    dtype = torch.float32 if full_precision else torch.float16
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
            for k, v in state.items()}


@torch.inference_mode()
def forward(params: dict[str, torch.Tensor], node_features: np.ndarray,
            edge_index: np.ndarray, edge_types: np.ndarray, *, layers: int = 4,
            residual: bool = True, edge_dim: int = 10) -> torch.Tensor:
    dtype = params["input.weight"].dtype
    x = torch.from_numpy(node_features).to(dtype)                 # api.py:237-238
    index = torch.from_numpy(edge_index).to(torch.long)            # api.py:239-240
    source, destination = index[0], index[1]
    attributes = F.one_hot(torch.from_numpy(edge_types).to(torch.long),
                           num_classes=edge_dim).to(dtype)         # api.py:243-245
    hidden = F.linear(x, params["input.weight"], params["input.bias"])
    for l in range(layers):
        c = f"convs.{l}."
        messages = F.relu(hidden.index_select(0, source) + F.linear(
            attributes, params[c + "edge_lin.weight"], params[c + "edge_lin.bias"]))
        aggregate = torch.zeros_like(hidden).index_add_(0, destination, messages)
        z = (1.0 + params[c + "eps"]) * hidden + aggregate
        u = F.linear(z, params[c + "mlp.0.weight"], params[c + "mlp.0.bias"])
        v = F.relu(F.batch_norm(
            u, params[c + "mlp.1.running_mean"], params[c + "mlp.1.running_var"],
            params[c + "mlp.1.weight"], params[c + "mlp.1.bias"],
            training=False, eps=1e-5))
        w = F.linear(v, params[c + "mlp.4.weight"], params[c + "mlp.4.bias"])
        update = F.layer_norm(w, (w.shape[1],), params[f"norms.{l}.weight"],
                              params[f"norms.{l}.bias"], eps=1e-5)
        hidden = hidden + update if residual else update
    t = F.relu(F.linear(hidden, params["head.0.weight"], params["head.0.bias"]))
    return F.linear(t, params["head.2.weight"], params["head.2.bias"])


def encode(params: dict[str, torch.Tensor], node_features, edge_index,
           edge_types, *, embedding_dtype=np.float16, **model) -> np.ndarray:
    raw = forward(params, node_features, edge_index, edge_types, **model)
    e = raw.to(torch.float32).numpy().astype(np.float64)           # api.py:250
    e = e / np.maximum(np.linalg.norm(e, axis=1, keepdims=True), 1e-12)
    return e.astype(embedding_dtype)
