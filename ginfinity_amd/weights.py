"""Checkpoint verification and the flat weight pack handed to libgfy.

Mirrors the loader/integrity half of ``Ginfinity.load``
(reference: src/ginfinity/api.py:77-109): SHA-256 of ``encoder.pt`` against
``model.json``, ``format_version == 1``, ``torch.load(weights_only=True)``,
architecture / graph-spec / parameter-count cross-checks.  Instead of building a
``torch.nn`` module, the verified state dict is flattened into the
little-endian pack documented in include/gfy.h; the library derives the
fp16-rounded tensors, the edge table, BatchNorm affine and MFMA fragment layout
from it.
"""
from __future__ import annotations

import hashlib
import json
import struct
from dataclasses import dataclass
from pathlib import Path

import numpy as np
import torch

from .spec import DATA_DIRECTORY, GraphSpec

PACK_MAGIC = 0x31594647          # 'GFY1'
PACK_VERSION = 1


class ModelIntegrityError(RuntimeError):
    """A packaged model artifact failed compatibility or integrity checks."""


@dataclass(frozen=True, slots=True)
class EncoderConfig:
    """Architecture record stored in the checkpoint (``_model.py:11-26``)."""

    hidden: int
    layers: int
    out_dim: int
    dropout: float
    struct_feature: str
    positional: bool
    residual: bool
    train_eps: bool
    edge_dim: int
    extra_edges: tuple[str, ...]

    @classmethod
    def from_dict(cls, value: dict) -> "EncoderConfig":
        return cls(**{**value, "extra_edges": tuple(value.get("extra_edges", ()))})

    @property
    def node_feature_dim(self) -> int:
        return (4 + (1 if self.struct_feature == "A" else 3)
                + (2 if self.positional else 0))


def tensor_order(layers: int) -> list[str]:
    """State-dict keys in weight-pack order (include/gfy.h)."""
    names = ["input.weight", "input.bias"]
    for l in range(layers):
        c = f"convs.{l}."
        names += [c + "eps", c + "edge_lin.weight", c + "edge_lin.bias",
                  c + "mlp.0.weight", c + "mlp.0.bias",
                  c + "mlp.1.weight", c + "mlp.1.bias",
                  c + "mlp.1.running_mean", c + "mlp.1.running_var",
                  c + "mlp.4.weight", c + "mlp.4.bias",
                  f"norms.{l}.weight", f"norms.{l}.bias"]
    return names + ["head.0.weight", "head.0.bias",
                    "head.2.weight", "head.2.bias"]


def expected_shapes(config: EncoderConfig) -> dict[str, tuple[int, ...]]:
    h, m = config.hidden, 2 * config.hidden
    shapes: dict[str, tuple[int, ...]] = {
        "input.weight": (h, config.node_feature_dim), "input.bias": (h,),
        "head.0.weight": (h, h), "head.0.bias": (h,),
        "head.2.weight": (config.out_dim, h), "head.2.bias": (config.out_dim,)}
    for l in range(config.layers):
        c = f"convs.{l}."
        shapes.update({
            c + "eps": (1,), c + "edge_lin.weight": (h, config.edge_dim),
            c + "edge_lin.bias": (h,), c + "mlp.0.weight": (m, h),
            c + "mlp.0.bias": (m,), c + "mlp.1.weight": (m,),
            c + "mlp.1.bias": (m,), c + "mlp.1.running_mean": (m,),
            c + "mlp.1.running_var": (m,), c + "mlp.4.weight": (h, m),
            c + "mlp.4.bias": (h,), f"norms.{l}.weight": (h,),
            f"norms.{l}.bias": (h,)})
    return shapes


_BUFFER_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked")


def parameter_count(state: dict[str, np.ndarray]) -> int:
    """Learnable parameters only (BatchNorm buffers excluded), as
    ``GINEEncoder.parameter_count`` counts them (_model.py:74-76)."""
    return sum(int(v.size) for k, v in state.items()
               if not k.endswith(_BUFFER_SUFFIXES))


def build_weight_pack(state: dict[str, np.ndarray],
                      config: EncoderConfig) -> bytes:
    shapes = expected_shapes(config)
    parts = [struct.pack(
        "<8I", PACK_MAGIC, PACK_VERSION, config.node_feature_dim,
        config.hidden, config.layers, config.edge_dim, config.out_dim,
        1 if config.residual else 0)]
    for name in tensor_order(config.layers):
        tensor = np.ascontiguousarray(state[name], dtype="<f4")
        if tensor.shape != shapes[name]:
            raise ModelIntegrityError(
                f"checkpoint tensor {name} has shape {tensor.shape}, "
                f"expected {shapes[name]}")
        parts.append(tensor.tobytes())
    return b"".join(parts)


@dataclass(frozen=True)
class LoadedCheckpoint:
    metadata: dict
    config: EncoderConfig
    graph_spec: GraphSpec
    state: dict[str, np.ndarray]      # fp32, exactly as stored
    weight_pack: bytes


def _read_json(path: Path) -> dict:
    try:
        return json.loads(path.read_text())
    except (OSError, json.JSONDecodeError) as error:
        raise ModelIntegrityError(
            f"cannot read model metadata {path}: {error}") from error


def load_checkpoint(model_dir: str | Path | None = None) -> LoadedCheckpoint:
    """Verify and read ``encoder.pt`` + ``model.json`` (api.py:77-109)."""
    root = Path(model_dir) if model_dir is not None else DATA_DIRECTORY
    metadata = _read_json(root / "model.json")
    checkpoint = root / "encoder.pt"
    if not checkpoint.is_file():
        raise ModelIntegrityError(f"missing checkpoint {checkpoint}")
    digest = hashlib.sha256(checkpoint.read_bytes()).hexdigest()
    if digest != metadata.get("checkpoint_sha256"):
        raise ModelIntegrityError("checkpoint SHA-256 mismatch")
    if metadata.get("format_version") != 1:
        raise ModelIntegrityError("unsupported model format")
    try:
        # weights_only=True: no arbitrary unpickling.  The tensors were saved
        # from a GPU process; they are read onto the host here and uploaded by
        # gfy_encoder_create.
        payload = torch.load(checkpoint, map_location="cpu", weights_only=True)
        config = EncoderConfig.from_dict(payload["cfg"])
        if config != EncoderConfig.from_dict(metadata["encoder_config"]):
            raise ModelIntegrityError(
                "checkpoint and metadata architecture mismatch")
        graph_spec = GraphSpec.from_dict(metadata["graph_spec"])
        derived = GraphSpec.from_encoder_config(config)
        if (graph_spec.sha256 != derived.sha256
                or metadata.get("graph_spec_sha256") != graph_spec.sha256):
            raise ModelIntegrityError("model and graph specification mismatch")
        raw = payload["state_dict"]
        wanted = set(tensor_order(config.layers))
        counters = {f"convs.{l}.mlp.1.num_batches_tracked"
                    for l in range(config.layers)}
        if set(raw) - counters != wanted:      # strict=True equivalent
            raise ModelIntegrityError(
                "checkpoint state dict does not match the architecture")
        state = {name: raw[name].detach().to(torch.float32).numpy()
                 for name in tensor_order(config.layers)}
        pack = build_weight_pack(state, config)
    except ModelIntegrityError:
        raise
    except Exception as error:
        raise ModelIntegrityError(
            f"checkpoint could not be loaded: {error}") from error
    if parameter_count(state) != metadata.get("parameter_count"):
        raise ModelIntegrityError("parameter-count mismatch")
    return LoadedCheckpoint(metadata, config, graph_spec, state, pack)


def random_state(config: EncoderConfig, seed: int = 0) -> dict[str, np.ndarray]:
    """Seeded random weights of the checkpoint's shapes (parity tests that must
    not depend on the bundled weights)."""
    rng = np.random.default_rng(seed)
    state: dict[str, np.ndarray] = {}
    for name, shape in expected_shapes(config).items():
        if name.endswith("running_var"):
            value = rng.uniform(0.05, 4.0, size=shape)
        elif name.endswith(("mlp.1.weight",)) or name.startswith("norms.") \
                and name.endswith("weight"):
            value = rng.uniform(0.5, 1.5, size=shape)
        elif name.endswith("eps"):
            value = rng.uniform(-0.7, 0.2, size=shape)
        elif name.endswith("weight"):
            value = rng.standard_normal(shape) / np.sqrt(shape[-1])
        else:
            value = 0.2 * rng.standard_normal(shape)
        state[name] = value.astype(np.float32)
    return state
