"""Generate the REFERENCE-WRITTEN file fixtures of the formats either side of the path.

Run only in the build container, where the reference is mounted read-only:

    python tests/golden/make_io_golden.py          # writes tests/golden/io/*

It imports ``ginfinity`` from /root/reference/src (never copied into this repo, never
shipped to the GPU box) and lets the reference itself write, for the first 24 records of
``rouskin_sample_6k.tsv`` (3.4 k nucleotides) and for a small windowed table:

  io/small.tsv                  the 24 records (transcript_id, sequence, secondary_structure)
  io/windowed.tsv               6 records with start / end columns (sliced graphs)
  io/ref_shard.safetensors      ``ginfinity build-graphs --checksum`` (graph.py:756-823,
  io/ref_shard.json              cli.py:112-136): tensor file + JSON sidecar with tensor_sha256
  io/ref_windowed.safetensors   the same for the windowed table with --keep-paired-neighbours
  io/ref_windowed.json           --context-hops 2: residue_index / node_roles tensors present
  io/ref_embed.npz              ``ginfinity embed`` (cli.py:69-111): np.savez_compressed archive,
  io/ref_embed.manifest.json     one member per record, + manifest
  io/ref_embed_graphs.npz       ``ginfinity embed-graphs --checksum`` on ref_shard
  io/ref_embed_graphs.manifest.json
  io/ref_windowed_embed.npz     ``ginfinity embed`` on the windowed table (core rows only)
  io/ref_windowed_embed.manifest.json

These are data files written by the reference's writers — what a user switching packages
already has on disk.  tests/test_io_fixtures.py loads them with this repo's readers
(``load_graph_shard(verify_checksum=True, validation="full")``), re-encodes, and compares
archives and manifests member by member.
"""
from __future__ import annotations

import json
import shutil
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
OUT = HERE / "io"
sys.path.insert(0, "/root/reference/src")

from ginfinity import cli as ref_cli            # noqa: E402  (the genuine reference)
from ginfinity import read_rna_table            # noqa: E402


def run(*argv: str) -> None:
    status = ref_cli.main([str(a) for a in argv])
    assert status == 0, (argv, status)


def relative_paths(manifest: Path) -> None:
    """The manifests record the paths they were called with: keep them relative to the
    fixture directory so the committed files do not depend on where they were generated."""
    data = json.loads(manifest.read_text())
    for key in ("input", "input_metadata", "output"):
        if key in data:
            data[key] = Path(data[key]).name
    manifest.write_text(json.dumps(data, indent=2) + "\n")


def main() -> None:
    if OUT.exists():
        shutil.rmtree(OUT)
    OUT.mkdir(parents=True)
    records = read_rna_table(HERE / "rouskin_sample_6k.tsv")[:24]
    with open(OUT / "small.tsv", "w") as handle:
        handle.write("transcript_id\tsequence\tsecondary_structure\n")
        for record in records:
            handle.write(f"{record.identifier}\t{record.sequence}\t{record.structure}\n")
    # windows: 0-based half-open [start, end) (cli.py:204-209), cut through stems on purpose
    with open(OUT / "windowed.tsv", "w") as handle:
        handle.write("transcript_id\tsequence\tsecondary_structure\tstart\tend\n")
        for index, record in enumerate(records[:6]):
            length = len(record.sequence)
            start = (index * 7) % max(length // 3, 1)
            end = min(length, start + length // 2)
            handle.write(f"{record.identifier}\t{record.sequence}\t{record.structure}"
                         f"\t{start}\t{end}\n")

    run("build-graphs", "--input", OUT / "small.tsv", "--output", OUT / "ref_shard.safetensors",
        "--checksum")
    run("build-graphs", "--input", OUT / "windowed.tsv",
        "--output", OUT / "ref_windowed.safetensors", "--checksum",
        "--keep-paired-neighbours", "--context-hops", "2")
    run("embed", "--input", OUT / "small.tsv", "--output", OUT / "ref_embed.npz")
    run("embed-graphs", "--input", OUT / "ref_shard.safetensors",
        "--output", OUT / "ref_embed_graphs.npz", "--checksum", "--verify-checksum",
        "--full-validation")
    run("embed", "--input", OUT / "windowed.tsv", "--output", OUT / "ref_windowed_embed.npz",
        "--keep-paired-neighbours", "--context-hops", "2")
    for manifest in OUT.glob("*.manifest.json"):
        relative_paths(manifest)
    for path in sorted(OUT.iterdir()):
        print(f"{path.stat().st_size:9d}  {path.name}")


if __name__ == "__main__":
    main()
