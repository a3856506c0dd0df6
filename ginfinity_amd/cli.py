"""Command-line interface — drop-in for ``ginfinity`` (reference:
src/ginfinity/cli.py:226-305): ``info``, ``alignment-config``, ``embed``,
``build-graphs``, ``embed-graphs`` with the same flags, NPZ + manifest outputs
and exit code 2 on any error.  ``--device`` defaults to ``cpu`` as in the reference
(cli.py:231,239,265); ``--device cuda --allow-nondeterministic-cuda`` selects the MI355X.
``build-graphs`` and ``alignment-config`` are host-only and need no GPU.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import platform
import sys
import time
from pathlib import Path

import numpy as np
import torch

from .npz import write_npz
from .api import Ginfinity, default_alignment_parameters
from .graph import GraphBuilder
from .shard_io import graph_metadata_path, load_graph_shard, save_graph_shard
from .table import read_rna_table

PACKAGE_VERSION = "1.2.1+mi355x.1"


def _file_sha256(path: Path) -> str:
    return hashlib.sha256(Path(path).read_bytes()).hexdigest()


def _builder_from(args: argparse.Namespace) -> GraphBuilder:
    # --context-hops implies --keep-paired-neighbours (cli.py:46-55)
    if args.context_hops is not None:
        return GraphBuilder(keep_paired_neighbours=True,
                            context_hops=int(args.context_hops))
    return GraphBuilder(
        keep_paired_neighbours=bool(args.keep_paired_neighbours))


def _records_from(args: argparse.Namespace):
    start, end = ((None, None) if args.no_slices
                  else (args.start_column, args.end_column))
    return read_rna_table(
        args.input, identifier_column=args.id_column,
        sequence_column=args.sequence_column,
        structure_column=args.structure_column,
        start_column=start, end_column=end, delimiter=args.delimiter)


def _load_encoder(args: argparse.Namespace) -> Ginfinity:
    return Ginfinity.load(
        device=args.device,
        allow_nondeterministic_cuda=args.allow_nondeterministic_cuda,
        full_precision=args.full_precision)


def _write_npz(path: Path, names, arrays) -> None:
    """The reference's ``np.savez_compressed`` archive (cli.py:85-88,158-163), members
    deflated on all cores (ginfinity_amd/npz.py)."""
    path.parent.mkdir(parents=True, exist_ok=True)
    write_npz(path, names, arrays)


def _provenance(encoder: Ginfinity) -> dict:
    info = encoder.info()
    return {"status": "complete", "ginfinity_version": PACKAGE_VERSION,
            "model_version": info["model_version"],
            "checkpoint_sha256": info["checkpoint_sha256"]}


def _cmd_embed(args: argparse.Namespace) -> int:
    began = time.time()
    records = _records_from(args)
    encoder = _load_encoder(args)
    builder = _builder_from(args)
    outputs = encoder.encode_many(
        records, max_batch_nodes=args.max_batch_nodes,
        max_batch_edges=args.max_batch_edges,
        keep_paired_neighbours=builder.keep_paired_neighbours,
        context_hops=builder.context_hops,
        embedding_dtype=args.embedding_dtype)
    _write_npz(args.output, (r.identifier for r in records), outputs)
    manifest_path = args.manifest or args.output.with_suffix(".manifest.json")
    rows = []
    for record, value in zip(records, outputs):
        row = {"identifier": record.identifier, "length": record.length,
               "core_length": int(value.shape[0]), "shape": list(value.shape)}
        if record.sliced:
            row.update(start=record.start, end=record.end)
        rows.append(row)
    manifest = {
        **_provenance(encoder),
        "input": str(args.input), "input_sha256": _file_sha256(args.input),
        "output": str(args.output), "output_sha256": _file_sha256(args.output),
        "device": args.device, "python": platform.python_version(),
        "numpy": np.__version__, "torch": torch.__version__,
        "records": rows, "elapsed_seconds": time.time() - began}
    manifest_path.write_text(json.dumps(manifest, indent=2) + "\n")
    print(json.dumps({"output": str(args.output),
                      "manifest": str(manifest_path),
                      "records": len(records)}))
    return 0


def _cmd_build_graphs(args: argparse.Namespace) -> int:
    began = time.time()
    shard = _builder_from(args).build_shard(_records_from(args))
    metadata_path = args.metadata or graph_metadata_path(args.output)
    save_graph_shard(shard, args.output, metadata_path=metadata_path,
                     checksum=args.checksum)
    print(json.dumps({
        "output": str(args.output), "metadata": str(metadata_path),
        "records": shard.record_count, "nodes": shard.node_count,
        "edges": shard.edge_count, "graph_spec_sha256": shard.spec.sha256,
        "checksum": args.checksum, "elapsed_seconds": time.time() - began}))
    return 0


def _cmd_embed_graphs(args: argparse.Namespace) -> int:
    began = time.time()
    encoder = _load_encoder(args)
    shard = load_graph_shard(
        args.input, metadata_path=args.metadata,
        expected_spec=encoder.graph_spec,
        verify_checksum=args.verify_checksum,
        validation="full" if args.full_validation else "metadata")
    outputs = encoder.encode_graphs(
        shard, max_batch_nodes=args.max_batch_nodes,
        max_batch_edges=args.max_batch_edges,
        embedding_dtype=args.embedding_dtype)
    _write_npz(args.output, shard.identifiers, outputs)
    manifest_path = args.manifest or args.output.with_suffix(".manifest.json")
    manifest = {
        **_provenance(encoder),
        "graph_spec_sha256": shard.spec.sha256,
        "input": str(args.input),
        "input_metadata": str(args.metadata or graph_metadata_path(args.input)),
        "output": str(args.output), "device": args.device,
        "records": [
            {"identifier": name, "length": len(sequence), "node_count": nodes,
             "core_length": core, "shape": list(value.shape)}
            for name, sequence, nodes, core, value in zip(
                shard.identifiers, shard.sequences, shard.lengths,
                shard.core_counts, outputs)],
        "elapsed_seconds": time.time() - began}
    if args.checksum:
        manifest["output_sha256"] = _file_sha256(args.output)
    manifest_path.parent.mkdir(parents=True, exist_ok=True)
    manifest_path.write_text(json.dumps(manifest, indent=2) + "\n")
    print(json.dumps({"output": str(args.output),
                      "manifest": str(manifest_path),
                      "records": shard.record_count}))
    return 0


def _table_options(parser: argparse.ArgumentParser) -> None:
    parser.add_argument("--id-column", default="transcript_id")
    parser.add_argument("--sequence-column", default="sequence")
    parser.add_argument("--structure-column", default="secondary_structure")
    parser.add_argument("--start-column", default="start",
                        help="optional 0-based half-open window start column")
    parser.add_argument("--end-column", default="end",
                        help="optional 0-based half-open window end column")
    parser.add_argument("--no-slices", action="store_true",
                        help="ignore start/end columns and encode full molecules")
    parser.add_argument("--delimiter", default="\t")
    parser.add_argument("--keep-paired-neighbours", action="store_true",
                        help="keep crossing pair partners outside each window")
    parser.add_argument("--context-hops", type=int, default=None, metavar="N",
                        help="neighbourhood depth around crossing pair partners "
                             "(implies --keep-paired-neighbours; hop 1 is the partner)")


def _encoder_options(parser: argparse.ArgumentParser) -> None:
    parser.add_argument("--device", default="cpu")
    parser.add_argument("--allow-nondeterministic-cuda", action="store_true")
    parser.add_argument("--full-precision", action="store_true",
                        help="run model inference in float32 instead of float16")
    parser.add_argument("--max-batch-nodes", type=int, default=60_000)
    parser.add_argument("--max-batch-edges", type=int, default=300_000)
    parser.add_argument("--embedding-dtype",
                        choices=("float16", "float32", "float64"),
                        default="float16",
                        help="dtype for returned node embeddings")


def _parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(prog="ginfinity")
    parser.add_argument("--version", action="version", version=PACKAGE_VERSION)
    commands = parser.add_subparsers(dest="command", required=True)
    info = commands.add_parser("info", help="show verified model metadata")
    info.add_argument("--device", default="cpu")
    alignment = commands.add_parser(
        "alignment-config", help="export parameters for ginfinity-sw")
    alignment.add_argument("--output", type=Path)

    embed = commands.add_parser("embed", help="encode a TSV of RNA records")
    embed.add_argument("--input", type=Path, required=True)
    embed.add_argument("--output", type=Path, required=True)
    embed.add_argument("--manifest", type=Path)
    _encoder_options(embed)
    _table_options(embed)

    build = commands.add_parser(
        "build-graphs", help="build a persistent graph shard from an RNA TSV")
    build.add_argument("--input", type=Path, required=True)
    build.add_argument("--output", type=Path, required=True)
    build.add_argument("--metadata", type=Path)
    build.add_argument("--checksum", action="store_true")
    _table_options(build)

    graphs = commands.add_parser(
        "embed-graphs", help="encode a previously built graph shard")
    graphs.add_argument("--input", type=Path, required=True)
    graphs.add_argument("--metadata", type=Path)
    graphs.add_argument("--output", type=Path, required=True)
    graphs.add_argument("--manifest", type=Path)
    _encoder_options(graphs)
    graphs.add_argument("--verify-checksum", action="store_true")
    graphs.add_argument("--full-validation", action="store_true")
    graphs.add_argument("--checksum", action="store_true")
    return parser


def main(argv: list[str] | None = None) -> int:
    args = _parser().parse_args(argv)
    try:
        if args.command == "info":
            print(json.dumps(Ginfinity.load(device=args.device).info(), indent=2))
            return 0
        if args.command == "alignment-config":
            text = json.dumps(
                {"scoring_parameters": default_alignment_parameters()},
                indent=2) + "\n"
            if args.output:
                args.output.parent.mkdir(parents=True, exist_ok=True)
                args.output.write_text(text)
            else:
                print(text, end="")
            return 0
        handler = {"embed": _cmd_embed, "build-graphs": _cmd_build_graphs,
                   "embed-graphs": _cmd_embed_graphs}[args.command]
        return handler(args)
    except Exception as error:           # reference contract: message + exit 2
        print(f"ginfinity: {error}", file=sys.stderr)
        return 2


if __name__ == "__main__":
    raise SystemExit(main())
