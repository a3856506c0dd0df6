"""RNA graph containers and the vectorised graph builder.

Host-side mirror of the reference's graph layer (src/ginfinity/graph.py).
``GraphShard`` is the input format of the HIP encoder: the four arrays
``node_features f32 (N,7)``, ``edge_index i32 (2,E)``, ``edge_types u8 (E,)``
and ``node_roles u8 (N,)`` go to the device unchanged (graph.py:261-344 is the
contract).

The builder is written with whole-array numpy operations (pair table by a
stable sort on nesting level, edges by ``arange`` blocks, context expansion by
boolean frontiers) instead of the reference's per-nucleotide Python loops
(graph.py:494-561, 599-695); integer outputs — including edge ORDER — and the
float32 node features are bit-identical to the reference's (tests/golden).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, Iterator, Sequence

import numpy as np

from .records import RNA
from .spec import (EDGE_TYPE_CODE, GRAPH_SHARD_FORMAT,
                   GRAPH_SHARD_FORMAT_VERSION, NODE_ROLE_CONTEXT,
                   NODE_ROLE_CORE, GraphCompatibilityError, GraphSpec,
                   GraphValidationError)

_INT32_MAX = int(np.iinfo(np.int32).max)


def _check_roles(node_roles: np.ndarray) -> None:
    if node_roles.size and int(node_roles.max()) > int(NODE_ROLE_CONTEXT):
        raise GraphValidationError("unknown node role")


# --------------------------------------------------------------------------
# one graph (reference: graph.py:164-258)
# --------------------------------------------------------------------------

@dataclass(frozen=True, slots=True)
class Graph:
    """One validated RNA graph with local zero-based edge indices.

    ``sequence``/``structure`` always describe the full source molecule;
    ``residue_index`` maps nodes back to it (a sliced graph holds the core
    window plus any retained context).  ``node_roles`` is provenance, not a
    model feature.
    """

    identifier: str
    sequence: str
    structure: str
    node_features: np.ndarray
    edge_index: np.ndarray
    edge_types: np.ndarray
    spec: GraphSpec
    residue_index: np.ndarray
    node_roles: np.ndarray

    def __post_init__(self) -> None:
        texts = (self.identifier, self.sequence, self.structure)
        if (not all(isinstance(text, str) for text in texts)
                or not self.identifier or not self.sequence
                or len(self.structure) != len(self.sequence)):
            raise GraphValidationError("invalid graph record metadata")
        residue = self.residue_index
        if residue.dtype != np.int32 or residue.ndim != 1 or residue.size == 0:
            raise GraphValidationError(
                "residue_index must be a non-empty int32 vector")
        nodes = residue.shape[0]
        if self.node_roles.dtype != np.uint8 or self.node_roles.shape != (nodes,):
            raise GraphValidationError("node_roles must match residue_index")
        if (self.node_features.dtype != np.float32 or self.node_features.shape
                != (nodes, self.spec.node_feature_dim)):
            raise GraphValidationError("invalid node feature array")
        edges = self.edge_index
        if edges.dtype != np.int32 or edges.ndim != 2 or edges.shape[0] != 2:
            raise GraphValidationError(
                "edge_index must have shape (2, E) and int32 dtype")
        if (self.edge_types.dtype != np.uint8
                or self.edge_types.shape != (edges.shape[1],)):
            raise GraphValidationError(
                "edge_types must have shape (E,) and uint8 dtype")
        if edges.size and (int(edges.min()) < 0 or int(edges.max()) >= nodes):
            raise GraphValidationError("edge index outside graph node range")
        if self.edge_types.size and int(self.edge_types.max()) >= self.spec.edge_dim:
            raise GraphValidationError("edge type outside graph feature range")
        if int(residue[0]) < 0 or int(residue[-1]) >= len(self.sequence) \
                or int(residue.min()) < 0 or int(residue.max()) >= len(self.sequence):
            raise GraphValidationError("residue index outside source sequence")
        if nodes > 1 and not bool((residue[1:] > residue[:-1]).all()):
            raise GraphValidationError("residue_index must be strictly increasing")
        _check_roles(self.node_roles)
        if not bool((self.node_roles == NODE_ROLE_CORE).any()):
            raise GraphValidationError("graph has no core nodes")

    @property
    def length(self) -> int:
        """Length of the SOURCE molecule (not the selected node count)."""
        return len(self.sequence)

    @property
    def node_count(self) -> int:
        return int(self.node_features.shape[0])

    @property
    def edge_count(self) -> int:
        return int(self.edge_index.shape[1])

    @property
    def core_mask(self) -> np.ndarray:
        return self.node_roles == NODE_ROLE_CORE

    @property
    def core_count(self) -> int:
        return int(np.count_nonzero(self.core_mask))

    @property
    def core_positions(self) -> np.ndarray:
        """0-based source coordinates of the core nodes, 5'→3'."""
        return self.residue_index[self.core_mask]

    @property
    def core_span(self) -> tuple[int, int]:
        core = self.core_positions
        return int(core[0]), int(core[-1]) + 1


# --------------------------------------------------------------------------
# a shard = many graphs, CSR-like ptr arrays (reference: graph.py:261-457)
# --------------------------------------------------------------------------

@dataclass(frozen=True, slots=True)
class GraphShard:
    """A persistent scheduling unit of one or more RNA graphs."""

    identifiers: tuple[str, ...]
    sequences: tuple[str, ...]
    structures: tuple[str, ...]
    node_features: np.ndarray
    edge_index: np.ndarray
    edge_types: np.ndarray
    node_ptr: np.ndarray
    edge_ptr: np.ndarray
    spec: GraphSpec
    residue_index: np.ndarray
    node_roles: np.ndarray

    def __post_init__(self) -> None:
        records = len(self.identifiers)
        if records == 0:
            raise GraphValidationError("a graph shard cannot be empty")
        if not (all(isinstance(v, str) and v for v in self.identifiers)
                and all(isinstance(v, str) and v for v in self.sequences)
                and all(isinstance(v, str) for v in self.structures)):
            raise GraphValidationError("invalid graph shard record metadata")
        if len(set(self.identifiers)) != records:
            raise GraphValidationError("duplicate identifiers in graph shard")
        if len(self.sequences) != records or len(self.structures) != records:
            raise GraphValidationError("graph shard metadata count mismatch")
        for name in ("node_ptr", "edge_ptr"):
            ptr = getattr(self, name)
            if ptr.dtype != np.int64 or ptr.shape != (records + 1,):
                raise GraphValidationError(
                    f"{name} must have shape (B + 1,) and int64 dtype")
        node_sizes = np.diff(self.node_ptr)
        if (self.node_ptr[0] != 0 or self.edge_ptr[0] != 0
                or bool((node_sizes <= 0).any())
                or bool((np.diff(self.edge_ptr) < 0).any())):
            raise GraphValidationError("invalid graph shard offsets")
        nodes, edges = int(self.node_ptr[-1]), int(self.edge_ptr[-1])
        if (self.node_features.dtype != np.float32 or self.node_features.shape
                != (nodes, self.spec.node_feature_dim)):
            raise GraphValidationError("invalid shard node feature array")
        if self.edge_index.dtype != np.int32 or self.edge_index.shape != (2, edges):
            raise GraphValidationError("invalid shard edge index array")
        if self.edge_types.dtype != np.uint8 or self.edge_types.shape != (edges,):
            raise GraphValidationError("invalid shard edge type array")
        if self.residue_index.dtype != np.int32 or self.residue_index.shape != (nodes,):
            raise GraphValidationError("invalid shard residue_index array")
        if self.node_roles.dtype != np.uint8 or self.node_roles.shape != (nodes,):
            raise GraphValidationError("invalid shard node_roles array")
        if edges and (int(self.edge_index.min()) < 0
                      or int(self.edge_index.max()) >= nodes):
            raise GraphValidationError("edge index outside shard node range")
        if edges and int(self.edge_types.max()) >= self.spec.edge_dim:
            raise GraphValidationError("edge type outside shard feature range")
        source_lengths = np.fromiter(
            (len(s) for s in self.sequences), dtype=np.int64, count=records)
        structure_lengths = np.fromiter(
            (len(s) for s in self.structures), dtype=np.int64, count=records)
        if bool((source_lengths != structure_lengths).any()):
            raise GraphValidationError(
                "sequence/structure length mismatch in shard")
        _check_roles(self.node_roles)
        # per-record checks, all records at once (the reference loops per
        # record, graph.py:330-343)
        starts = self.node_ptr[:-1]
        residue = self.residue_index
        low = np.minimum.reduceat(residue, starts)
        high = np.maximum.reduceat(residue, starts)
        if bool((low < 0).any()) or bool((high >= source_lengths).any()):
            raise GraphValidationError("residue index outside source sequence")
        if nodes > 1:
            rising = residue[1:] > residue[:-1]
            rising[starts[1:] - 1] = True          # record boundaries are exempt
            if not bool(rising.all()):
                raise GraphValidationError(
                    "residue_index must be strictly increasing")
        core_per_record = np.add.reduceat(
            (self.node_roles == NODE_ROLE_CORE).astype(np.int64), starts)
        if bool((core_per_record == 0).any()):
            raise GraphValidationError("graph has no core nodes")

    # -- sizes ----------------------------------------------------------------
    @property
    def record_count(self) -> int:
        return len(self.identifiers)

    @property
    def node_count(self) -> int:
        return int(self.node_ptr[-1])

    @property
    def edge_count(self) -> int:
        return int(self.edge_ptr[-1])

    @property
    def lengths(self) -> tuple[int, ...]:
        """Selected node count per graph (context nodes included)."""
        return tuple(np.diff(self.node_ptr).tolist())

    @property
    def edge_counts(self) -> tuple[int, ...]:
        return tuple(np.diff(self.edge_ptr).tolist())

    @property
    def core_counts(self) -> tuple[int, ...]:
        return tuple(self.core_count_array().tolist())

    def core_count_array(self) -> np.ndarray:
        """int64 [records]: core nodes per record (== the lengths when no record has context
        nodes, the usual case — one pass over ``node_roles`` finds that out)."""
        if not self.node_roles.any():            # NODE_ROLE_CORE == 0
            return np.diff(self.node_ptr).astype(np.int64)
        return np.add.reduceat(self.node_roles == NODE_ROLE_CORE, self.node_ptr[:-1],
                               dtype=np.int64)

    # -- construction -----------------------------------------------------------
    @classmethod
    def from_graphs(cls, graphs: Sequence[Graph]) -> "GraphShard":
        """Concatenate graphs, rebasing edge indices (graph.py:376-412)."""
        graphs = list(graphs)
        if not graphs:
            raise GraphValidationError("cannot create a shard without graphs")
        spec = graphs[0].spec
        if any(g.spec.sha256 != spec.sha256 for g in graphs):
            raise GraphCompatibilityError(
                "all graphs in a shard must use the same graph specification")
        node_ptr = np.zeros(len(graphs) + 1, dtype=np.int64)
        edge_ptr = np.zeros(len(graphs) + 1, dtype=np.int64)
        np.cumsum([g.node_count for g in graphs], out=node_ptr[1:])
        np.cumsum([g.edge_count for g in graphs], out=edge_ptr[1:])
        if int(node_ptr[-1]) > _INT32_MAX:
            raise GraphValidationError(
                "graph shard exceeds the int32 node-index capacity; split it")
        edge_index = np.concatenate([g.edge_index for g in graphs], axis=1)
        edge_index += np.repeat(node_ptr[:-1], np.diff(edge_ptr)).astype(np.int32)

        def joined(name: str) -> np.ndarray:
            return np.ascontiguousarray(
                np.concatenate([getattr(g, name) for g in graphs], axis=0))

        return cls(
            identifiers=tuple(g.identifier for g in graphs),
            sequences=tuple(g.sequence for g in graphs),
            structures=tuple(g.structure for g in graphs),
            node_features=joined("node_features"),
            edge_index=np.ascontiguousarray(edge_index),
            edge_types=joined("edge_types"),
            node_ptr=node_ptr, edge_ptr=edge_ptr, spec=spec,
            residue_index=joined("residue_index"),
            node_roles=joined("node_roles"))

    def slice(self, start: int, stop: int) -> "GraphShard":
        """Records ``[start, stop)`` as their own shard, indices rebased
        (graph.py:414-444; SURVEY §8 a10 — integer work, bit-exact)."""
        if not 0 <= start < stop <= self.record_count:
            raise IndexError("invalid graph shard slice")
        n0, n1 = int(self.node_ptr[start]), int(self.node_ptr[stop])
        e0, e1 = int(self.edge_ptr[start]), int(self.edge_ptr[stop])
        return GraphShard(
            identifiers=self.identifiers[start:stop],
            sequences=self.sequences[start:stop],
            structures=self.structures[start:stop],
            node_features=np.ascontiguousarray(self.node_features[n0:n1]),
            edge_index=np.ascontiguousarray(
                self.edge_index[:, e0:e1] - np.int32(n0), dtype=np.int32),
            edge_types=np.ascontiguousarray(self.edge_types[e0:e1]),
            node_ptr=np.ascontiguousarray(
                self.node_ptr[start:stop + 1] - n0, dtype=np.int64),
            edge_ptr=np.ascontiguousarray(
                self.edge_ptr[start:stop + 1] - e0, dtype=np.int64),
            spec=self.spec,
            residue_index=np.ascontiguousarray(self.residue_index[n0:n1]),
            node_roles=np.ascontiguousarray(self.node_roles[n0:n1]))

    def validate_values(self) -> None:
        """Opt-in linear-time value checks: finite features, no edge leaving
        its own graph (graph.py:446-457)."""
        if not bool(np.isfinite(self.node_features).all()):
            raise GraphValidationError("non-finite node features in graph shard")
        if self.edge_count == 0:
            return
        owner = np.repeat(np.arange(self.record_count), np.diff(self.edge_ptr))
        low = self.node_ptr[:-1][owner]
        high = self.node_ptr[1:][owner]
        inside = (self.edge_index >= low) & (self.edge_index < high)
        if not bool(inside.all()):
            raise GraphValidationError("edge crosses graph boundaries")


# --------------------------------------------------------------------------
# builder (reference: graph.py:460-567, 599-747)
# --------------------------------------------------------------------------

_BASE_CODE = np.full(256, -1, dtype=np.int64)
for _code, _base in enumerate(b"ACGU"):
    _BASE_CODE[_base] = _code
_OPEN, _CLOSE, _DOT = ord("("), ord(")"), ord(".")


def pair_table(structure: str) -> np.ndarray:
    """partner[i] = index paired with i, or -1 (int32).

    Brackets at the same nesting level alternate open/close in text order, so
    a stable sort of bracket positions by level lines partners up pairwise.
    Equivalent to the reference's stack walk (graph.py:737-747) for balanced
    input, which ``RNA`` guarantees.
    """
    chars = np.frombuffer(structure.encode("ascii"), dtype=np.uint8)
    partners = np.full(chars.shape[0], -1, dtype=np.int32)
    opening = chars == _OPEN
    closing = chars == _CLOSE
    bracket_positions = np.flatnonzero(opening | closing)
    if bracket_positions.size == 0:
        return partners
    depth = np.cumsum(opening.astype(np.int32) - closing.astype(np.int32))
    level = depth + closing            # a ')' closes the level it came from
    by_level = bracket_positions[
        np.argsort(level[bracket_positions], kind="stable")]
    left, right = by_level[0::2], by_level[1::2]
    partners[left] = right
    partners[right] = left
    return partners


# the reference's private name, kept for callers that reached for it
_pair_table = pair_table


class GraphBuilder:
    """Deterministically convert validated RNA records into model-ready graphs.

    ``keep_paired_neighbours`` retains nucleotides outside a requested window
    when they are base-paired with a core nucleotide; ``context_hops`` is the
    depth of that neighbourhood (hop 1 = the crossing-pair partner, further
    hops follow every graph edge).
    """

    def __init__(self, spec: GraphSpec | None = None, *,
                 keep_paired_neighbours: bool = False,
                 context_hops: int = 1) -> None:
        if context_hops < 1:
            raise ValueError("context_hops must be >= 1")
        self.spec = spec if spec is not None else GraphSpec.bundled()
        self.keep_paired_neighbours = bool(keep_paired_neighbours)
        self.context_hops = int(context_hops)

    # -- node features (graph.py:496-514) ----------------------------------
    def _node_features(self, sequence: str, structure: str) -> np.ndarray:
        length = len(sequence)
        spec = self.spec
        out = np.zeros((length, spec.node_feature_dim), dtype=np.float32)
        rows = np.arange(length)
        out[rows, _BASE_CODE[np.frombuffer(sequence.encode("ascii"), np.uint8)]] = 1
        marks = np.frombuffer(structure.encode("ascii"), np.uint8)
        if spec.struct_feature == "A":
            out[:, 4] = marks != _DOT
            column = 5
        else:
            state = np.where(marks == _OPEN, 0, np.where(marks == _DOT, 1, 2))
            out[rows, 4 + state] = 1
            column = 7
        if spec.positional:
            # float32 throughout, exactly the reference's expression so the
            # sin/cos columns are bit-identical
            relative = np.arange(length, dtype=np.float32) / max(length - 1, 1)
            out[:, column] = np.sin(np.pi * relative)
            out[:, column + 1] = np.cos(np.pi * relative)
        return out

    # -- typed directed edges, in the reference's order (graph.py:516-546) --
    def _edges(self, structure: str) -> tuple[np.ndarray, np.ndarray]:
        length = len(structure)
        partners = pair_table(structure)
        opens = np.flatnonzero(partners > np.arange(length)).astype(np.int32)
        closes = partners[opens]
        head = np.arange(max(length - 1, 0), dtype=np.int32)
        blocks_src = [head, head + 1, opens, closes]
        blocks_dst = [head + 1, head, closes, opens]
        blocks_typ = [
            np.full(head.size, EDGE_TYPE_CODE["backbone_forward"], np.uint8),
            np.full(head.size, EDGE_TYPE_CODE["backbone_reverse"], np.uint8),
            np.full(opens.size, EDGE_TYPE_CODE["base_pair_forward"], np.uint8),
            np.full(opens.size, EDGE_TYPE_CODE["base_pair_reverse"], np.uint8),
        ]
        if self.spec.has_skip2 and length > 2:
            near = np.arange(length - 2, dtype=np.int32)
            far = near + 2
            # interleaved (i→i+2, i+2→i) per i
            blocks_src.append(np.stack((near, far), axis=1).ravel())
            blocks_dst.append(np.stack((far, near), axis=1).ravel())
            blocks_typ.append(np.tile(np.array(
                [EDGE_TYPE_CODE["skip2_forward"],
                 EDGE_TYPE_CODE["skip2_reverse"]], np.uint8), length - 2))
        edge_index = np.ascontiguousarray(np.stack(
            (np.concatenate(blocks_src), np.concatenate(blocks_dst))),
            dtype=np.int32)
        return edge_index, np.ascontiguousarray(np.concatenate(blocks_typ))

    def _build_full(self, record: RNA) -> Graph:
        edge_index, edge_types = self._edges(record.structure)
        return Graph(
            identifier=record.identifier, sequence=record.sequence,
            structure=record.structure,
            node_features=self._node_features(record.sequence, record.structure),
            edge_index=edge_index, edge_types=edge_types, spec=self.spec,
            residue_index=np.arange(record.length, dtype=np.int32),
            node_roles=np.full(record.length, NODE_ROLE_CORE, dtype=np.uint8))

    # -- windows (graph.py:608-695) -------------------------------------------
    def _window(self, full: Graph, start: int, end: int) -> Graph:
        nodes = full.node_count
        source, destination = full.edge_index
        chosen = np.zeros(nodes, dtype=bool)
        chosen[start:end] = True
        if self.keep_paired_neighbours:
            partners = pair_table(full.structure)
            mates = partners[start:end]
            mates = mates[mates >= 0]
            frontier = np.zeros(nodes, dtype=bool)
            frontier[mates] = True
            frontier &= ~chosen
            chosen |= frontier
            for _ in range(self.context_hops - 1):
                if not frontier.any():
                    break
                reached = np.zeros(nodes, dtype=bool)
                reached[destination[frontier[source]]] = True
                frontier = reached & ~chosen
                chosen |= frontier
        residue_index = np.flatnonzero(chosen).astype(np.int32)
        roles = np.where((residue_index >= start) & (residue_index < end),
                         NODE_ROLE_CORE, NODE_ROLE_CONTEXT).astype(np.uint8)
        kept = chosen[source] & chosen[destination]
        renumber = np.cumsum(chosen, dtype=np.int32) - np.int32(1)
        edge_index = np.ascontiguousarray(
            renumber[full.edge_index[:, kept]], dtype=np.int32)
        return Graph(
            identifier=full.identifier, sequence=full.sequence,
            structure=full.structure,
            node_features=np.ascontiguousarray(full.node_features[residue_index]),
            edge_index=edge_index.reshape(2, -1),
            edge_types=np.ascontiguousarray(full.edge_types[kept]),
            spec=full.spec, residue_index=residue_index, node_roles=roles)

    # -- public -------------------------------------------------------------------
    def build(self, record: RNA) -> Graph:
        full = self._build_full(record)
        if not record.sliced:
            return full
        return self._window(full, record.start, record.end)

    def build_many(self, records: Iterable[RNA]) -> list[Graph]:
        return [self.build(record) for record in records]

    def build_shard(self, records: Iterable[RNA]) -> GraphShard:
        """One shard from many records.  Unsliced records — the normal case — are
        built in ONE pass of whole-shard array operations (no per-record Python or
        numpy call overhead: for 136-nt median RNAs that overhead was the cost);
        the arrays are bit-identical to ``from_graphs(build_many(records))``."""
        records = list(records)
        if not records:
            return GraphShard.from_graphs(self.build_many(records))   # the reference's error
        whole, partners = self._build_shard_whole(records)
        if not any(record.sliced for record in records):
            return whole
        return self._window_shard(whole, partners, records)

    def _window_shard(self, whole: GraphShard, partners: np.ndarray,
                      records: Sequence[RNA]) -> GraphShard:
        """``_window`` (graph.py:608-695) for every record of a shard at once: the windows, the
        crossing-pair partners and the further context hops are masks over the whole shard's
        nodes (graphs never share an edge, so one sweep over all edges expands every record's
        frontier), then ONE compaction.  Bit-identical to
        ``from_graphs([_window(full, start, end) ...])``; the per-record loop cost 0.3 ms per
        record, most of it numpy call overhead."""
        count = len(records)
        node_ptr = whole.node_ptr
        lengths = np.diff(node_ptr)
        total = int(node_ptr[-1])
        record_of = np.repeat(np.arange(count, dtype=np.int64), lengths)
        position = whole.residue_index.astype(np.int64)            # 0 .. L-1 per record
        starts = np.fromiter((r.start if r.sliced else 0 for r in records), np.int64, count)
        ends = np.fromiter((r.end if r.sliced else r.length for r in records), np.int64, count)
        core = (position >= starts[record_of]) & (position < ends[record_of])
        chosen = core.copy()
        source, destination = whole.edge_index
        if self.keep_paired_neighbours:
            mates = partners[core]
            mates = mates[mates >= 0]
            frontier = np.zeros(total, dtype=bool)
            frontier[mates] = True
            frontier &= ~chosen
            chosen |= frontier
            for _ in range(self.context_hops - 1):
                if not frontier.any():
                    break
                reached = np.zeros(total, dtype=bool)
                reached[destination[frontier[source]]] = True
                frontier = reached & ~chosen
                chosen |= frontier
        kept = chosen[source] & chosen[destination]
        renumber = np.cumsum(chosen, dtype=np.int32) - np.int32(1)
        new_node_ptr = np.zeros(count + 1, dtype=np.int64)
        np.cumsum(np.add.reduceat(chosen.astype(np.int64), node_ptr[:-1]), out=new_node_ptr[1:])
        edge_owner = np.repeat(np.arange(count, dtype=np.int64), np.diff(whole.edge_ptr))
        new_edge_ptr = np.zeros(count + 1, dtype=np.int64)
        np.cumsum(np.bincount(edge_owner[kept], minlength=count), out=new_edge_ptr[1:])
        return GraphShard(
            identifiers=whole.identifiers, sequences=whole.sequences,
            structures=whole.structures,
            node_features=np.ascontiguousarray(whole.node_features[chosen]),
            edge_index=np.ascontiguousarray(renumber[whole.edge_index[:, kept]],
                                            dtype=np.int32).reshape(2, -1),
            edge_types=np.ascontiguousarray(whole.edge_types[kept]),
            node_ptr=new_node_ptr, edge_ptr=new_edge_ptr, spec=whole.spec,
            residue_index=position[chosen].astype(np.int32),
            node_roles=np.where(core[chosen], NODE_ROLE_CORE, NODE_ROLE_CONTEXT).astype(np.uint8))

    def _build_shard_whole(self, records: Sequence[RNA]) -> tuple[GraphShard, np.ndarray]:
        """The shard of the records' WHOLE molecules (windows ignored) and its pair table."""
        spec = self.spec
        count = len(records)
        lengths = np.fromiter((r.length for r in records), dtype=np.int64, count=count)
        node_ptr = np.zeros(count + 1, dtype=np.int64)
        np.cumsum(lengths, out=node_ptr[1:])
        total = int(node_ptr[-1])
        if total > _INT32_MAX:
            raise GraphValidationError(
                "graph shard exceeds the int32 node-index capacity; split it")
        sequences = tuple(r.sequence for r in records)
        structures = tuple(r.structure for r in records)
        bases = np.frombuffer("".join(sequences).encode("ascii"), dtype=np.uint8)
        marks = np.frombuffer("".join(structures).encode("ascii"), dtype=np.uint8)
        record_of = np.repeat(np.arange(count, dtype=np.int64), lengths)
        position = np.arange(total, dtype=np.int64) - node_ptr[record_of]

        # ---- node features (same expressions as _node_features, per element) --------
        features = np.zeros((total, spec.node_feature_dim), dtype=np.float32)
        rows = np.arange(total)
        features[rows, _BASE_CODE[bases]] = 1
        if spec.struct_feature == "A":
            features[:, 4] = marks != _DOT
            column = 5
        else:
            state = np.where(marks == _OPEN, 0, np.where(marks == _DOT, 1, 2))
            features[rows, 4 + state] = 1
            column = 7
        if spec.positional:
            denominator = np.maximum(lengths - 1, 1).astype(np.float32)[record_of]
            relative = position.astype(np.float32) / denominator
            features[:, column] = np.sin(np.float32(np.pi) * relative)
            features[:, column + 1] = np.cos(np.float32(np.pi) * relative)

        # ---- pair table of the whole text: every record is balanced, so the running
        # depth returns to zero at record ends and (record, level) identifies a nest
        opening = marks == _OPEN
        closing = marks == _CLOSE
        brackets = np.flatnonzero(opening | closing)
        partners = np.full(total, -1, dtype=np.int64)
        if brackets.size:
            depth = np.cumsum(opening.astype(np.int32) - closing.astype(np.int32))
            level = (depth + closing)[brackets].astype(np.int64)
            key = record_of[brackets] * np.int64(level.max() + 1) + level
            ordered = brackets[np.argsort(key, kind="stable")]
            left, right = ordered[0::2], ordered[1::2]
            partners[left] = right
            partners[right] = left
        opens = np.flatnonzero(partners > np.arange(total))     # ascending, per record too
        closes = partners[opens]
        pairs = np.bincount(record_of[opens], minlength=count).astype(np.int64)

        # ---- edges: per record [backbone fwd, backbone rev, pair fwd, pair rev, skip-2]
        head = np.maximum(lengths - 1, 0)
        skip = np.maximum(lengths - 2, 0) if spec.has_skip2 else np.zeros(count, np.int64)
        per_record = 2 * head + 2 * pairs + 2 * skip
        edge_ptr = np.zeros(count + 1, dtype=np.int64)
        np.cumsum(per_record, out=edge_ptr[1:])
        edges = int(edge_ptr[-1])
        source = np.empty(edges, dtype=np.int32)
        destination = np.empty(edges, dtype=np.int32)
        types = np.empty(edges, dtype=np.uint8)

        def place(block_sizes, block_offset, src, dst, code, stride=1, lane=0):
            """Scatter one block kind of every record: element k of record r goes to
            edge_ptr[r] + block_offset[r] + stride * k + lane."""
            owner = np.repeat(np.arange(count, dtype=np.int64), block_sizes)
            first = np.zeros(count, dtype=np.int64)
            np.cumsum(block_sizes[:-1], out=first[1:])
            local = np.arange(owner.size, dtype=np.int64) - first[owner]
            at = edge_ptr[owner] + block_offset[owner] + stride * local + lane
            source[at] = src(owner, local)
            destination[at] = dst(owner, local)
            types[at] = code

        base = node_ptr[:-1]
        zero = np.zeros(count, dtype=np.int64)
        place(head, zero, lambda r, i: base[r] + i, lambda r, i: base[r] + i + 1,
              EDGE_TYPE_CODE["backbone_forward"])
        place(head, head, lambda r, i: base[r] + i + 1, lambda r, i: base[r] + i,
              EDGE_TYPE_CODE["backbone_reverse"])
        # pair blocks: opens are already grouped by record in ascending order
        pair_first = np.zeros(count, dtype=np.int64)
        np.cumsum(pairs[:-1], out=pair_first[1:])
        pair_owner = record_of[opens]
        pair_local = np.arange(opens.size, dtype=np.int64) - pair_first[pair_owner]
        at = edge_ptr[pair_owner] + 2 * head[pair_owner] + pair_local
        source[at], destination[at] = opens, closes
        types[at] = EDGE_TYPE_CODE["base_pair_forward"]
        at = at + pairs[pair_owner]
        source[at], destination[at] = closes, opens
        types[at] = EDGE_TYPE_CODE["base_pair_reverse"]
        if spec.has_skip2:
            after_pairs = 2 * head + 2 * pairs
            place(skip, after_pairs, lambda r, i: base[r] + i, lambda r, i: base[r] + i + 2,
                  EDGE_TYPE_CODE["skip2_forward"], stride=2, lane=0)
            place(skip, after_pairs, lambda r, i: base[r] + i + 2, lambda r, i: base[r] + i,
                  EDGE_TYPE_CODE["skip2_reverse"], stride=2, lane=1)

        return GraphShard(
            identifiers=tuple(r.identifier for r in records),
            sequences=sequences, structures=structures,
            node_features=features,
            edge_index=np.ascontiguousarray(np.stack((source, destination))),
            edge_types=types, node_ptr=node_ptr, edge_ptr=edge_ptr, spec=spec,
            residue_index=position.astype(np.int32),
            node_roles=np.full(total, NODE_ROLE_CORE, dtype=np.uint8)), partners


@dataclass(frozen=True)
class ShardText:
    """Unsliced records in the form the device builder takes (``gfy_build_graphs``,
    include/gfy.h): the concatenated sequence and dot-bracket bytes plus the node / edge
    offsets of every record — everything the host can know without building a graph."""

    bases: np.ndarray       # uint8 [N]
    marks: np.ndarray       # uint8 [N]
    node_ptr: np.ndarray    # int64 [R+1]
    edge_ptr: np.ndarray    # int64 [R+1]
    spec: GraphSpec

    @property
    def lengths(self) -> np.ndarray:
        return np.diff(self.node_ptr)

    @property
    def edge_counts(self) -> np.ndarray:
        return np.diff(self.edge_ptr)

    def positional(self, start: int, stop: int) -> np.ndarray | None:
        """float32 [nodes of records start..stop, 2]: sin / cos of the relative position,
        the reference's numpy float32 expression (graph.py:508-513), or None."""
        if not self.spec.positional:
            return None
        lengths = self.lengths[start:stop]
        first = self.node_ptr[start]
        nodes = int(self.node_ptr[stop] - first)
        owner = np.repeat(np.arange(stop - start, dtype=np.int64), lengths)
        position = np.arange(nodes, dtype=np.int64) - (self.node_ptr[start:stop] - first)[owner]
        denominator = np.maximum(lengths - 1, 1).astype(np.float32)[owner]
        angle = np.float32(np.pi) * (position.astype(np.float32) / denominator)
        columns = np.empty((nodes, 2), dtype=np.float32)
        np.sin(angle, out=columns[:, 0])
        np.cos(angle, out=columns[:, 1])
        return columns


def shard_text(records: Sequence[RNA], spec: GraphSpec) -> ShardText:
    """``ShardText`` of unsliced records.  The edge count of a record follows from its
    length and its number of '(' (graph.py:516-546): 2(L-1) backbone + 2 pairs +
    2 max(L-2, 0) skip-2 edges."""
    if any(record.sliced for record in records):
        raise ValueError("shard_text takes unsliced records only")
    count = len(records)
    if count == 0:
        raise GraphValidationError("a graph shard cannot be empty")
    if len({record.identifier for record in records}) != count:
        raise GraphValidationError("duplicate identifiers in graph shard")
    sequences = [record.sequence for record in records]
    lengths = np.fromiter(map(len, sequences), dtype=np.int64, count=count)
    node_ptr = np.zeros(count + 1, dtype=np.int64)
    np.cumsum(lengths, out=node_ptr[1:])
    if int(node_ptr[-1]) > _INT32_MAX:
        raise GraphValidationError(
            "graph shard exceeds the int32 node-index capacity; split it")
    bases = np.frombuffer("".join(sequences).encode("ascii"), np.uint8)
    marks = np.frombuffer("".join(record.structure for record in records).encode("ascii"),
                          np.uint8)
    # '(' per record, counted on the concatenated text (records are never empty)
    pairs = np.add.reduceat((marks == _OPEN).astype(np.int64), node_ptr[:-1])
    per_record = 2 * np.maximum(lengths - 1, 0) + 2 * pairs
    if spec.has_skip2:
        per_record += 2 * np.maximum(lengths - 2, 0)
    edge_ptr = np.zeros(count + 1, dtype=np.int64)
    np.cumsum(per_record, out=edge_ptr[1:])
    return ShardText(bases.copy(), marks.copy(), node_ptr, edge_ptr, spec)


def partition_records(records: Iterable[RNA], *, max_records: int,
                      max_nodes: int | None = None
                      ) -> Iterator[tuple[RNA, ...]]:
    """Deterministic scheduling units without building graphs
    (graph.py:570-596)."""
    if max_records <= 0:
        raise ValueError("max_records must be positive")
    if max_nodes is not None and max_nodes <= 0:
        raise ValueError("max_nodes must be positive")
    group: list[RNA] = []
    load = 0
    for record in records:
        if max_nodes is not None and record.length > max_nodes:
            raise ValueError(
                f"record {record.identifier!r} exceeds max_nodes={max_nodes}")
        full = len(group) >= max_records or (
            max_nodes is not None and load + record.length > max_nodes)
        if group and full:
            yield tuple(group)
            group, load = [], 0
        group.append(record)
        load += record.length
    if group:
        yield tuple(group)


from .shard_io import (_file_sha256 as _sha256,  # noqa: E402,F401  (what shard I/O hashes with)
                       graph_metadata_path, load_graph_shard, save_graph_shard)

__all__ = [
    "GRAPH_SHARD_FORMAT", "GRAPH_SHARD_FORMAT_VERSION", "NODE_ROLE_CONTEXT",
    "NODE_ROLE_CORE", "Graph", "GraphBuilder", "GraphCompatibilityError",
    "GraphShard", "GraphSpec", "GraphValidationError", "ShardText",
    "graph_metadata_path", "load_graph_shard", "pair_table", "partition_records",
    "save_graph_shard", "shard_text",
]
