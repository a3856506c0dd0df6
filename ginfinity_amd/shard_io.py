"""Graph-shard persistence: safetensors tensors + JSON sidecar.

Byte-compatible with the reference's artifact contract
(src/ginfinity/graph.py:750-923, docs/GRAPH_PIPELINE.md:26-50): a shard written
here loads in the reference and vice versa.  Writes are atomic
(temp file + ``os.replace``); content hashing is opt-in.
"""
from __future__ import annotations

import hashlib
import json
import os
import tempfile
from contextlib import contextmanager
from pathlib import Path
from typing import Iterator, Literal

import numpy as np
from safetensors.numpy import save_file

from .spec import (GRAPH_SHARD_FORMAT, GRAPH_SHARD_FORMAT_VERSION,
                   NODE_ROLE_CORE, GraphCompatibilityError, GraphSpec,
                   GraphValidationError)

_REQUIRED = frozenset(
    ("node_features", "edge_index", "edge_types", "node_ptr", "edge_ptr"))
_OPTIONAL = frozenset(("residue_index", "node_roles"))


def _file_sha256(path: Path) -> str:
    digest = hashlib.sha256()
    with path.open("rb") as handle:
        while chunk := handle.read(1 << 20):
            digest.update(chunk)
    return digest.hexdigest()


def _sha256(path: Path) -> str:
    """The tensor file's hash, looked up through ``ginfinity_amd.graph._sha256`` at call time:
    the reference keeps shard I/O and this helper in graph.py (graph.py:756-923) and its tests
    replace ``graph._sha256`` to prove that hashing stays opt-in."""
    from . import graph
    return graph._sha256(path)


_SAFETENSORS_DTYPES = {"F32": np.float32, "I32": np.int32, "I64": np.int64, "U8": np.uint8,
                       "F16": np.float16, "F64": np.float64, "I8": np.int8, "I16": np.int16,
                       "BOOL": np.bool_}


def _map_tensors(path: Path) -> tuple[dict[str, str], dict[str, np.ndarray]]:
    """(header metadata, {name: read-only array}) of a safetensors file WITHOUT reading it:
    the arrays are views of one ``mmap`` of the file (8-byte little-endian header length, JSON
    header with dtype / shape / data_offsets per tensor, then the byte buffer).  Pages come in
    from the page cache when a micro-batch's slice is copied into the pinned upload buffer —
    the only host copy between the file and the device (``safetensors.numpy.load_file`` reads
    the whole payload into fresh arrays first).

    Lifetime: the arrays are READ-ONLY views of the mapping, which lives as long as any of them
    (numpy keeps the memmap referenced).  Replacing the file through ``save_graph_shard`` —
    or any writer that renames a new file over the path, as the reference's does
    (graph.py:770-791) — leaves a live mapping on the old inode untouched; truncating or
    rewriting the file IN PLACE while a shard loaded from it is alive is undefined (SIGBUS),
    as for any memory-mapped file."""
    size = path.stat().st_size
    with path.open("rb") as handle:
        prefix = handle.read(8)
        if len(prefix) != 8:
            raise ValueError("file too short for a safetensors header")
        header_bytes = int.from_bytes(prefix, "little")
        if header_bytes <= 0 or 8 + header_bytes > size:
            raise ValueError("safetensors header length out of range")
        header = json.loads(handle.read(header_bytes))
    if not isinstance(header, dict):
        raise ValueError("safetensors header is not an object")
    metadata = header.pop("__metadata__", None) or {}
    payload = np.memmap(path, mode="r", dtype=np.uint8, offset=8 + header_bytes,
                        shape=(size - 8 - header_bytes,)) if size > 8 + header_bytes else \
        np.zeros(0, np.uint8)
    arrays = {}
    spans: list[tuple[int, int, str]] = []
    for name, entry in header.items():
        dtype = np.dtype(_SAFETENSORS_DTYPES[entry["dtype"]])
        shape = tuple(int(d) for d in entry["shape"])
        begin, end = (int(v) for v in entry["data_offsets"])
        count = int(np.prod(shape, dtype=np.int64))
        if not 0 <= begin <= end <= payload.shape[0] or end - begin != count * dtype.itemsize:
            raise ValueError(f"tensor {name!r}: data offsets do not match dtype and shape")
        arrays[name] = np.frombuffer(payload, dtype=dtype, count=count,
                                     offset=begin).reshape(shape)
        spans.append((begin, end, name))
    # what safetensors itself checks on load: the tensors tile the byte buffer — no overlap, no
    # hole, nothing behind the last one
    position = 0
    for begin, end, name in sorted(spans):
        if begin != position:
            raise ValueError(f"tensor {name!r}: byte ranges overlap or leave a gap")
        position = end
    if position != payload.shape[0]:
        raise ValueError("safetensors file has trailing bytes behind its last tensor")
    return metadata, arrays


def graph_metadata_path(tensor_path: str | Path) -> Path:
    """Conventional JSON sidecar path of a tensor shard."""
    return Path(tensor_path).with_suffix(".json")


@contextmanager
def _atomic_target(final: Path) -> Iterator[Path]:
    """Yield a temp path beside ``final``; move it into place on success."""
    final.parent.mkdir(parents=True, exist_ok=True)
    handle, name = tempfile.mkstemp(
        dir=final.parent, prefix=f".{final.name}.", suffix=".tmp")
    os.close(handle)
    scratch = Path(name)
    try:
        yield scratch
        os.replace(scratch, final)
    finally:
        scratch.unlink(missing_ok=True)


def _full_molecule_layout(sequences, node_ptr) -> tuple[np.ndarray, np.ndarray]:
    """residue_index/node_roles implied by whole-molecule records."""
    sizes = np.fromiter((len(s) for s in sequences), np.int64, len(sequences))
    if node_ptr.dtype != np.int64 or node_ptr.shape != (len(sequences) + 1,):
        raise GraphValidationError("invalid graph shard offsets")
    if not np.array_equal(np.diff(node_ptr), sizes):
        raise GraphValidationError("sequence lengths do not match node offsets")
    total = int(node_ptr[-1])
    residue = (np.arange(total, dtype=np.int64)
               - np.repeat(node_ptr[:-1], sizes)).astype(np.int32)
    return residue, np.full(total, NODE_ROLE_CORE, dtype=np.uint8)


def _carries_window_metadata(shard) -> bool:
    """True when roles / residue indices cannot be re-derived from sequences."""
    if bool((shard.node_roles != NODE_ROLE_CORE).any()):
        return True
    try:
        residue, _ = _full_molecule_layout(shard.sequences, shard.node_ptr)
    except GraphValidationError:
        return True
    return not np.array_equal(residue, shard.residue_index)


def save_graph_shard(shard, tensor_path: str | Path, *,
                     metadata_path: str | Path | None = None,
                     checksum: bool = False) -> tuple[Path, Path]:
    """Atomically persist ``shard``; returns (tensor_path, metadata_path)."""
    tensor_path = Path(tensor_path)
    metadata_path = (Path(metadata_path) if metadata_path is not None
                     else graph_metadata_path(tensor_path))
    if tensor_path.resolve() == metadata_path.resolve():
        raise ValueError("tensor and metadata paths must be different")
    tensors = {name: getattr(shard, name) for name in sorted(_REQUIRED)}
    if _carries_window_metadata(shard):
        tensors.update({name: getattr(shard, name) for name in _OPTIONAL})
    header = {"format": GRAPH_SHARD_FORMAT,
              "format_version": str(GRAPH_SHARD_FORMAT_VERSION),
              "graph_spec_sha256": shard.spec.sha256}
    with _atomic_target(tensor_path) as scratch:
        save_file(tensors, str(scratch), metadata=header)
    sidecar = {
        "format": GRAPH_SHARD_FORMAT,
        "format_version": GRAPH_SHARD_FORMAT_VERSION,
        "graph_spec": shard.spec.to_dict(),
        "graph_spec_sha256": shard.spec.sha256,
        "tensor_file": tensor_path.name,
        "record_count": shard.record_count,
        "node_count": shard.node_count,
        "edge_count": shard.edge_count,
        "identifiers": list(shard.identifiers),
        "sequences": list(shard.sequences),
        "structures": list(shard.structures),
    }
    if checksum:
        sidecar["tensor_sha256"] = _sha256(tensor_path)
    metadata_path.parent.mkdir(parents=True, exist_ok=True)
    with _atomic_target(metadata_path) as scratch:
        scratch.write_text(json.dumps(sidecar, indent=2) + "\n")
    return tensor_path, metadata_path


def load_graph_shard(tensor_path: str | Path, *,
                     metadata_path: str | Path | None = None,
                     expected_spec: GraphSpec | None = None,
                     verify_checksum: bool = False,
                     validation: Literal["metadata", "full"] = "metadata"):
    """Load and validate a shard without deserialising Python objects."""
    from .graph import GraphShard

    tensor_path = Path(tensor_path)
    metadata_path = (Path(metadata_path) if metadata_path is not None
                     else graph_metadata_path(tensor_path))
    if validation not in ("metadata", "full"):
        raise ValueError("validation must be 'metadata' or 'full'")
    try:
        sidecar = json.loads(metadata_path.read_text())
    except (OSError, json.JSONDecodeError) as error:
        raise GraphValidationError(
            f"cannot read graph shard metadata: {error}") from error
    if (sidecar.get("format") != GRAPH_SHARD_FORMAT
            or sidecar.get("format_version") != GRAPH_SHARD_FORMAT_VERSION):
        raise GraphValidationError("unsupported graph shard format")
    try:
        spec = GraphSpec.from_dict(sidecar["graph_spec"])
    except GraphValidationError:
        raise
    except (KeyError, TypeError, ValueError) as error:
        raise GraphValidationError(
            f"invalid graph specification metadata: {error}") from error
    if sidecar.get("graph_spec_sha256") != spec.sha256:
        raise GraphValidationError("graph specification fingerprint mismatch")
    if expected_spec is not None and expected_spec.sha256 != spec.sha256:
        raise GraphCompatibilityError(
            "graph shard specification is incompatible with the encoder")
    if verify_checksum:
        stored = sidecar.get("tensor_sha256")
        if not stored:
            raise GraphValidationError("graph shard has no stored checksum")
        if _sha256(tensor_path) != stored:
            raise GraphValidationError("graph shard checksum mismatch")
    try:
        header, arrays = _map_tensors(tensor_path)
        if (header.get("format") != GRAPH_SHARD_FORMAT
                or header.get("format_version") != str(GRAPH_SHARD_FORMAT_VERSION)
                or header.get("graph_spec_sha256") != spec.sha256):
            raise GraphValidationError("tensor header metadata mismatch")
    except GraphValidationError:
        raise
    except Exception as error:
        raise GraphValidationError(
            f"cannot load graph shard tensors: {error}") from error
    names = set(arrays)
    if not _REQUIRED <= names:
        raise GraphValidationError("graph shard tensor set mismatch")
    extra = names - _REQUIRED - _OPTIONAL
    if extra:
        raise GraphValidationError(
            "unexpected graph shard tensor(s): " + ", ".join(sorted(extra)))
    if len(names & _OPTIONAL) == 1:
        raise GraphValidationError(
            "residue_index and node_roles must be stored together")
    try:
        sequences = tuple(sidecar["sequences"])
        if "residue_index" in arrays:
            residue_index, node_roles = (arrays["residue_index"],
                                         arrays["node_roles"])
        else:                      # shard written before windows existed
            residue_index, node_roles = _full_molecule_layout(
                sequences, arrays["node_ptr"])
        shard = GraphShard(
            identifiers=tuple(sidecar["identifiers"]), sequences=sequences,
            structures=tuple(sidecar["structures"]),
            node_features=arrays["node_features"],
            edge_index=arrays["edge_index"], edge_types=arrays["edge_types"],
            node_ptr=arrays["node_ptr"], edge_ptr=arrays["edge_ptr"],
            spec=spec, residue_index=residue_index, node_roles=node_roles)
    except GraphValidationError:
        raise
    except (KeyError, TypeError, ValueError) as error:
        raise GraphValidationError(
            f"invalid graph shard metadata: {error}") from error
    counts = (sidecar.get("record_count"), sidecar.get("node_count"),
              sidecar.get("edge_count"))
    if counts != (shard.record_count, shard.node_count, shard.edge_count):
        raise GraphValidationError("graph shard count metadata mismatch")
    if validation == "full":
        shard.validate_values()
    return shard
