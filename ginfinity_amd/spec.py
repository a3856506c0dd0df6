"""Graph contract (feature/edge vocabulary) shared by builder, shards and encoder.

Mirror of the reference's ``GraphSpec`` (src/ginfinity/graph.py:18-161): the
canonical-JSON SHA-256 fingerprint must be byte-identical to the reference's so
that shards written by either implementation are interchangeable
(bundled model: da2e670e…fb9bd, data/model.json:42).
"""
from __future__ import annotations

import hashlib
import json
from dataclasses import dataclass, field
from pathlib import Path
from typing import Mapping

import numpy as np

GRAPH_SHARD_FORMAT = "ginfinity-graph-shard"
GRAPH_SHARD_FORMAT_VERSION = 1

# edge-type codes (reference: graph.py:20-27); order defines the code.
EDGE_TYPE_NAMES = (
    "backbone_forward", "backbone_reverse",
    "base_pair_forward", "base_pair_reverse",
    "skip2_forward", "skip2_reverse",
)
EDGE_TYPE_CODE = {name: code for code, name in enumerate(EDGE_TYPE_NAMES)}

# per-node provenance for sliced graphs (not model features)
NODE_ROLE_CORE = np.uint8(0)
NODE_ROLE_CONTEXT = np.uint8(1)

DATA_DIRECTORY = Path(__file__).resolve().parent / "data"


class GraphValidationError(ValueError):
    """A graph or graph shard violates the public interchange contract."""


class GraphCompatibilityError(GraphValidationError):
    """A graph was built with a specification the encoder does not accept."""


def canonical_json(value: Mapping) -> bytes:
    """Sorted keys, no whitespace, ASCII — the hashing form (graph.py:46-49)."""
    return json.dumps(value, sort_keys=True, separators=(",", ":"),
                      ensure_ascii=True).encode("utf-8")


@dataclass(frozen=True, slots=True)
class GraphSpec:
    """Model-versioned node-feature and edge-type contract."""

    format_version: int = 1
    struct_feature: str = "A"
    positional: bool = True
    edge_dim: int = 10
    extra_edges: tuple[str, ...] = ("skip2",)
    _fingerprint: str = field(init=False, repr=False, compare=False)

    def __post_init__(self) -> None:
        object.__setattr__(self, "extra_edges", tuple(self.extra_edges))
        if self.format_version != GRAPH_SHARD_FORMAT_VERSION:
            raise GraphValidationError(
                f"unsupported graph specification version {self.format_version}")
        if self.struct_feature not in ("A", "B"):
            raise GraphValidationError(
                f"unsupported structure feature {self.struct_feature!r}")
        unknown = sorted(set(self.extra_edges) - {"skip2"})
        if unknown:
            raise GraphValidationError(
                "unsupported extra edge type(s): " + ", ".join(unknown))
        if self.edge_dim < len(self.edge_types):
            raise GraphValidationError(
                f"edge_dim={self.edge_dim} cannot represent all configured edges")
        object.__setattr__(
            self, "_fingerprint",
            hashlib.sha256(canonical_json(self.to_dict())).hexdigest())

    # -- derived ----------------------------------------------------------
    @property
    def has_skip2(self) -> bool:
        return "skip2" in self.extra_edges

    @property
    def node_feature_dim(self) -> int:
        structural = 1 if self.struct_feature == "A" else 3
        return 4 + structural + (2 if self.positional else 0)

    @property
    def edge_types(self) -> dict[str, int]:
        count = 6 if self.has_skip2 else 4
        return {name: EDGE_TYPE_CODE[name] for name in EDGE_TYPE_NAMES[:count]}

    @property
    def sha256(self) -> str:
        """Contract identifier (not a per-graph content hash)."""
        return self._fingerprint

    # -- (de)serialisation --------------------------------------------------
    def to_dict(self) -> dict:
        return {
            "format_version": self.format_version,
            "struct_feature": self.struct_feature,
            "positional": self.positional,
            "node_feature_dimension": self.node_feature_dim,
            "edge_feature_dimension": self.edge_dim,
            "edge_types": self.edge_types,
            "extra_edges": list(self.extra_edges),
        }

    @classmethod
    def from_dict(cls, value: Mapping) -> "GraphSpec":
        edge_dim = value.get("edge_feature_dimension", value.get("edge_dim", 10))
        spec = cls(format_version=int(value.get("format_version", 1)),
                   struct_feature=str(value["struct_feature"]),
                   positional=bool(value["positional"]),
                   edge_dim=int(edge_dim),
                   extra_edges=tuple(value.get("extra_edges", ())))
        declared = value.get("node_feature_dimension")
        if declared is not None and int(declared) != spec.node_feature_dim:
            raise GraphValidationError("node feature dimension is inconsistent")
        if "edge_types" in value and dict(value["edge_types"]) != spec.edge_types:
            raise GraphValidationError("edge type mapping is inconsistent")
        return spec

    @classmethod
    def from_encoder_config(cls, config: Mapping | object) -> "GraphSpec":
        get = (config.__getitem__ if isinstance(config, Mapping)
               else lambda name: getattr(config, name))
        return cls(struct_feature=str(get("struct_feature")),
                   positional=bool(get("positional")),
                   edge_dim=int(get("edge_dim")),
                   extra_edges=tuple(get("extra_edges")))

    @classmethod
    def bundled(cls) -> "GraphSpec":
        """The packaged model's graph contract, without touching the weights."""
        path = DATA_DIRECTORY / "model.json"
        try:
            metadata = json.loads(path.read_text())
            spec = cls.from_dict(metadata["graph_spec"])
        except (OSError, KeyError, TypeError, ValueError) as error:
            raise GraphValidationError(
                f"cannot read bundled graph specification: {error}") from error
        if metadata.get("graph_spec_sha256") != spec.sha256:
            raise GraphValidationError(
                "bundled graph specification fingerprint mismatch")
        return spec
