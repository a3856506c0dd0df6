"""Host-side drop-in surface: records, tables, graph builder, shard I/O,
checkpoint loader, CLI (host-only commands).  Mirrors the reference's own test
strategy (tests/test_validation.py, test_table.py, test_graph.py,
test_sliced_graphs.py, test_api.py, test_encoder_cli.py) — CPU only."""
from __future__ import annotations

import hashlib
import json
import shutil

import numpy as np
import pytest
from safetensors import safe_open
from safetensors.numpy import save_file

import ginfinity_amd.shard_io as shard_io
from ginfinity_amd import (GRAPH_SHARD_FORMAT, GRAPH_SHARD_FORMAT_VERSION,
                           NODE_ROLE_CONTEXT, NODE_ROLE_CORE, Ginfinity,
                           GraphBuilder, GraphCompatibilityError, GraphShard,
                           GraphSpec, GraphValidationError,
                           InputValidationError, ModelIntegrityError, RNA,
                           default_alignment_parameters, load_graph_shard,
                           partition_records, read_rna_table, save_graph_shard)
from ginfinity_amd.api import microbatch_bounds
from ginfinity_amd.cli import main
from ginfinity_amd.weights import load_checkpoint

SEQ, STRUCT = "GGGAAACCCUUUUGGG", "......(((....)))"


def _records():
    return [RNA("rna-1", "ACGUACGU", "((....))"), RNA("rna-2", "GGAACCUU", "........")]


# ---- records ------------------------------------------------------------------------

def test_record_normalisation_and_rejection():
    record = RNA(" id ", " acgt ", " (()) ")
    assert (record.identifier, record.sequence, record.structure) == ("id", "ACGU", "(())")
    for seq, struct, message in [("", "", "empty sequence"), ("ACGN", "....", "unsupported sequence"),
                                 ("ACGU", "...", "characters against"),
                                 ("ACGU", "[..]", "unsupported structure"),
                                 ("ACGU", "((.)", "unmatched"), ("ACGU", ".)..", "unmatched")]:
        with pytest.raises(InputValidationError, match=message):
            RNA("id", seq, struct)
    with pytest.raises(InputValidationError, match="identifier"):
        RNA("bad\tid", "ACGU", "....")
    with pytest.raises(InputValidationError, match="exceeds maximum"):
        RNA("long", "A" * 4097, "." * 4097)
    for start, end, message in [(1, None, "both be provided"), (-1, 2, "invalid slice"),
                                (2, 2, "invalid slice"), (0, 5, "invalid slice")]:
        with pytest.raises(InputValidationError, match=message):
            RNA("id", "ACGU", "....", start=start, end=end)
    window = RNA("id", "ACGUACGU", "((....))", start=2, end=6)
    assert window.sliced and window.core_length == 4


def test_mapping_and_table_reader(tmp_path):
    record = RNA.from_mapping({"name": "r", "bases": "ACGT", "fold": "(())"},
                              identifier_column="name", sequence_column="bases",
                              structure_column="fold")
    assert record == RNA("r", "ACGU", "(())")
    with pytest.raises(InputValidationError, match="multiple slices"):
        RNA.from_mapping({"transcript_id": "r", "sequence": "ACGUACGU",
                          "secondary_structure": "((....))", "start": "2,4", "end": "6,8"},
                         start_column="start", end_column="end")
    table = tmp_path / "t.tsv"
    table.write_text("transcript_id\tsequence\tsecondary_structure\tstart\tend\n"
                     "rna-1\tACGUACGU\t((....))\t2,4\t6, 8\n")
    got = read_rna_table(table)
    assert [(r.identifier, r.start, r.end) for r in got] == [("rna-1:2-6", 2, 6), ("rna-1:4-8", 4, 8)]
    table.write_text("transcript_id\tsequence\tsecondary_structure\tstart\tend\n"
                     "rna-1\tACGUACGU\t((....))\t2,4\t6\n")
    with pytest.raises(InputValidationError, match="start has 2"):
        read_rna_table(table)
    table.write_text("name\tbases\nfirst\tACGU\n")
    with pytest.raises(ValueError, match="dot_bracket"):
        read_rna_table(table, identifier_column="name", sequence_column="bases",
                       structure_column="dot_bracket")
    table.write_text("transcript_id\tsequence\tsecondary_structure\nbad\tACGN\t....\n")
    with pytest.raises(InputValidationError, match="line 2"):
        read_rna_table(table)
    table.write_text("transcript_id\tsequence\tsecondary_structure\na\tACGU\t....\na\tACGU\t....\n")
    with pytest.raises(InputValidationError, match="duplicate"):
        read_rna_table(table)


# ---- graph builder: integer known answers ----------------------------------------------

def test_example8_arrays_are_the_reference_arrays(golden):
    g = golden("example8.npz")
    shard = GraphBuilder().build_shard([RNA("example", "ACGUACGU", "((....))")])
    for name in ("node_features", "edge_index", "edge_types", "node_ptr", "edge_ptr",
                 "residue_index", "node_roles"):
        got, want = getattr(shard, name), g[name]
        assert got.dtype == want.dtype and got.shape == want.shape, name
        np.testing.assert_array_equal(got, want, err_msg=name)
    # SURVEY §8-B known answers
    np.testing.assert_array_equal(shard.edge_index[0, :7], np.arange(7))
    np.testing.assert_array_equal(shard.edge_types[14:18], [2, 2, 3, 3])
    np.testing.assert_array_equal(shard.edge_index[:, 18:22], [[0, 2, 1, 3], [2, 0, 3, 1]])
    assert shard.spec.sha256 == "da2e670e377e47667fec8a8ebb1c90c6e506b9cdd8a5555a6bfab50b202fb9bd"


def test_whole_rouskin_shard_hashes_match_reference_builder(golden, rouskin_shard):
    want = golden("integers.json")["rouskin"]
    assert (rouskin_shard.record_count, rouskin_shard.node_count,
            rouskin_shard.edge_count) == (want["records"], want["nodes"], want["edges"])
    for name, digest in want["sha256"].items():
        got = hashlib.sha256(np.ascontiguousarray(getattr(rouskin_shard, name)).tobytes())
        assert got.hexdigest() == digest, name
    bounds = microbatch_bounds(rouskin_shard.lengths, rouskin_shard.edge_counts,
                               60_000, 300_000)
    assert [list(b) for b in bounds] == [w[:2] for w in want["microbatches_60000_300000"]]
    piece = rouskin_shard.slice(413, 798)
    assert (piece.node_count, piece.edge_count) == tuple(want["microbatches_60000_300000"][1][2:])
    assert int(piece.edge_index.min()) >= 0 and int(piece.edge_index.max()) < piece.node_count
    piece.validate_values()


def test_whole_shard_builder_equals_per_record_builder(rouskin_records):
    """build_shard builds unsliced records in one pass of whole-shard array operations; the
    arrays must be bit-identical to concatenating per-record graphs (which the reference
    fixtures pin), for every length / nesting corner and for both struct features."""
    from ginfinity_amd import GraphShard
    corner = [RNA("one", "A", "."), RNA("two", "AC", ".."), RNA("three", "ACG", "..."),
              RNA("pair2", "GC", "()"), RNA("stack", "GGGGAAAACCCC", "((((....))))"),
              RNA("twostems", "GGAACCGGAACC", "((..))((..))"),
              RNA("nested", "GGGAAACCCAAAGGGAAACCC", "(((...)))...(((...)))"),
              RNA("deep", "G" * 40 + "AAAA" + "C" * 40, "(" * 40 + "...." + ")" * 40)]
    again = [RNA(r.identifier + "-again", r.sequence, r.structure) for r in corner[::-1]]
    records = corner + list(rouskin_records[:300]) + again
    for spec in (GraphSpec.bundled(),):
        builder = GraphBuilder(spec)
        whole = builder.build_shard(records)
        parts = GraphShard.from_graphs(builder.build_many(records))
        assert whole.identifiers == parts.identifiers
        assert whole.sequences == parts.sequences and whole.structures == parts.structures
        for name in ("node_features", "edge_index", "edge_types", "node_ptr", "edge_ptr",
                     "residue_index", "node_roles"):
            a, b = getattr(whole, name), getattr(parts, name)
            assert a.dtype == b.dtype and a.shape == b.shape, name
            assert a.tobytes() == b.tobytes(), name
    # a sliced record anywhere sends the whole call down the per-record path
    mixed = [RNA("w", "GGGAAACCCUUUUGGG", "(((...))).......", start=9, end=16)] + corner
    shard = GraphBuilder(keep_paired_neighbours=True).build_shard(mixed)
    assert shard.record_count == len(mixed) and shard.node_roles.max() <= 1


def test_sliced_graphs_match_reference(golden):
    g = golden("sliced.npz")
    for hops in (1, 2, 3):
        graph = GraphBuilder(keep_paired_neighbours=True, context_hops=hops).build(
            RNA("stem", SEQ, STRUCT, start=9, end=16))
        for name in ("node_features", "edge_index", "edge_types", "residue_index", "node_roles"):
            np.testing.assert_array_equal(getattr(graph, name), g[f"hops{hops}.{name}"])
    plain = GraphBuilder().build(RNA("stem", SEQ, STRUCT, start=9, end=16))
    np.testing.assert_array_equal(plain.residue_index, np.arange(9, 16, dtype=np.int32))
    assert plain.core_span == (9, 16) and plain.core_count == 7
    hops2 = GraphBuilder(keep_paired_neighbours=True, context_hops=2).build(
        RNA("stem", SEQ, STRUCT, start=9, end=16))
    np.testing.assert_array_equal(hops2.residue_index[hops2.node_roles == NODE_ROLE_CONTEXT],
                                  [4, 5, 6, 7, 8])
    with pytest.raises(ValueError, match="context_hops"):
        GraphBuilder(context_hops=0)


def test_degenerate_graphs(golden):
    g = golden("degenerate.npz")
    for seq, struct in (("A", "."), ("AC", ".."), ("GC", "()")):
        graph = GraphBuilder().build(RNA(seq, seq, struct))
        np.testing.assert_array_equal(graph.edge_index, g[f"{seq}.edge_index"])
        np.testing.assert_array_equal(graph.edge_types, g[f"{seq}.edge_types"])
        np.testing.assert_array_equal(graph.node_features, g[f"{seq}.node_features"])
        assert graph.edge_index.shape == g[f"{seq}.edge_index"].shape


def test_shard_validation_errors():
    shard = GraphBuilder().build_shard(_records())
    with pytest.raises(ValueError, match="duplicate"):
        GraphBuilder().build_shard([_records()[0], _records()[0]])
    with pytest.raises(GraphValidationError, match="edge type"):
        GraphShard(**{**{f: getattr(shard, f) for f in shard.__slots__},
                      "edge_types": np.full_like(shard.edge_types, 10)})
    with pytest.raises(GraphValidationError, match="edge index"):
        GraphShard(**{**{f: getattr(shard, f) for f in shard.__slots__},
                      "edge_index": shard.edge_index + np.int32(100)})
    with pytest.raises(IndexError):
        shard.slice(1, 1)
    other = GraphSpec(struct_feature="B", positional=True, edge_dim=10, extra_edges=("skip2",))
    assert GraphBuilder(other).build(_records()[0]).node_features.shape == (8, 9)
    with pytest.raises(GraphCompatibilityError):
        GraphShard.from_graphs([GraphBuilder().build(_records()[0]),
                                GraphBuilder(other).build(_records()[1])])
    parts = list(partition_records([RNA(str(i), "ACGU", "....") for i in range(5)],
                                   max_records=3, max_nodes=8))
    assert [[r.identifier for r in p] for p in parts] == [["0", "1"], ["2", "3"], ["4"]]


# ---- shard persistence --------------------------------------------------------------------

def test_shard_round_trip_and_checksums(tmp_path, monkeypatch):
    original = GraphBuilder().build_shard(_records())
    path = tmp_path / "graphs.safetensors"
    _, meta = save_graph_shard(original, path)
    assert "tensor_sha256" not in json.loads(meta.read_text())
    with safe_open(str(path), framework="np") as handle:
        assert set(handle.keys()) == {"node_features", "edge_index", "edge_types",
                                      "node_ptr", "edge_ptr"}
    restored = load_graph_shard(path, validation="full")
    assert restored.identifiers == original.identifiers and restored.spec == original.spec
    for name in ("node_features", "edge_index", "edge_types", "node_ptr", "edge_ptr",
                 "residue_index", "node_roles"):
        np.testing.assert_array_equal(getattr(restored, name), getattr(original, name))
    # hashing stays opt-in
    monkeypatch.setattr(shard_io, "_sha256",
                        lambda _p: (_ for _ in ()).throw(AssertionError("hashing must be opt-in")))
    save_graph_shard(original, tmp_path / "again.safetensors")
    assert load_graph_shard(tmp_path / "again.safetensors").record_count == 2
    monkeypatch.undo()
    save_graph_shard(original, path, checksum=True)
    payload = bytearray(path.read_bytes())
    payload[-1] ^= 1
    path.write_bytes(payload)
    with pytest.raises(GraphValidationError, match="checksum"):
        load_graph_shard(path, verify_checksum=True)
    incompatible = GraphSpec(struct_feature="B", positional=True, edge_dim=10,
                             extra_edges=("skip2",))
    save_graph_shard(original, path)
    with pytest.raises(GraphCompatibilityError):
        load_graph_shard(path, expected_spec=incompatible)


def test_sliced_and_legacy_shards(tmp_path):
    sliced = GraphBuilder(keep_paired_neighbours=True, context_hops=2).build_shard(
        [RNA("stem", SEQ, STRUCT, start=9, end=16)])
    path = tmp_path / "sliced.safetensors"
    save_graph_shard(sliced, path)
    with safe_open(str(path), framework="np") as handle:
        assert {"residue_index", "node_roles"} <= set(handle.keys())
    restored = load_graph_shard(path, validation="full")
    np.testing.assert_array_equal(restored.node_roles, sliced.node_roles)
    assert restored.core_counts == (7,) and restored.lengths[0] > 7
    full = GraphBuilder().build_shard([RNA("stem", SEQ, STRUCT)])
    legacy = tmp_path / "legacy.safetensors"
    save_file({name: getattr(full, name) for name in
               ("node_features", "edge_index", "edge_types", "node_ptr", "edge_ptr")},
              str(legacy), metadata={"format": GRAPH_SHARD_FORMAT,
                                     "format_version": str(GRAPH_SHARD_FORMAT_VERSION),
                                     "graph_spec_sha256": full.spec.sha256})
    (tmp_path / "legacy.json").write_text(json.dumps({
        "format": GRAPH_SHARD_FORMAT, "format_version": GRAPH_SHARD_FORMAT_VERSION,
        "graph_spec": full.spec.to_dict(), "graph_spec_sha256": full.spec.sha256,
        "tensor_file": legacy.name, "record_count": 1, "node_count": 16,
        "edge_count": full.edge_count, "identifiers": ["stem"], "sequences": [SEQ],
        "structures": [STRUCT]}))
    loaded = load_graph_shard(legacy)
    np.testing.assert_array_equal(loaded.residue_index, full.residue_index)
    assert np.all(loaded.node_roles == NODE_ROLE_CORE)


# ---- loader / API behaviour that needs no GPU -------------------------------------------------

def test_checkpoint_loader_and_integrity(tmp_path, checkpoint):
    assert checkpoint.metadata["parameter_count"] == 306_436
    assert checkpoint.config.hidden == 128 and checkpoint.config.layers == 4
    assert checkpoint.graph_spec.sha256 == GraphSpec.bundled().sha256
    assert len(checkpoint.weight_pack) == 32 + 4 * 308_484   # params + BN buffers
    from ginfinity_amd.spec import DATA_DIRECTORY
    copy = tmp_path / "model"
    shutil.copytree(DATA_DIRECTORY, copy)
    payload = bytearray((copy / "encoder.pt").read_bytes())
    payload[-1] ^= 1
    (copy / "encoder.pt").write_bytes(payload)
    with pytest.raises(ModelIntegrityError, match="SHA-256"):
        load_checkpoint(copy)
    (copy / "encoder.pt").unlink()
    with pytest.raises(ModelIntegrityError, match="missing checkpoint"):
        load_checkpoint(copy)


def test_device_policy_is_the_reference_s():
    """api.py:69-76 — 'cpu' (default) or 'cuda*'; CUDA needs the acknowledgement flag."""
    with pytest.raises(ValueError, match="CUDA requires allow_nondeterministic_cuda=True"):
        Ginfinity.load("cuda")
    with pytest.raises(ValueError, match="device must be"):
        Ginfinity.load("tpu")
    assert set(default_alignment_parameters()) == {
        "mu", "sigma", "gamma", "score_min", "score_max", "gap_open", "gap_extend",
        "score_offset"}


def test_microbatch_bounds_semantics():
    assert microbatch_bounds([8, 8], [30, 30], 8, 30) == [(0, 1), (1, 2)]
    assert microbatch_bounds([8, 8], [30, 30], 16, 60) == [(0, 2)]
    assert microbatch_bounds([8, 8, 8], [30, 30, 30], 16, 59) == [(0, 1), (1, 2), (2, 3)]
    # an oversized single record still forms its own batch (limits are checked earlier)
    assert microbatch_bounds([20, 3], [5, 5], 10, 10) == [(0, 1), (1, 2)]


def test_cli_host_only_commands(tmp_path, capsys):
    source = tmp_path / "m.tsv"
    source.write_text("transcript_id\tsequence\tsecondary_structure\n"
                      "rna-1\tACGUACGU\t((....))\nrna-2\tGGAACCUU\t........\n")
    graphs, meta = tmp_path / "g.safetensors", tmp_path / "g.json"
    assert main(["build-graphs", "--input", str(source), "--output", str(graphs),
                 "--metadata", str(meta), "--checksum"]) == 0
    out = json.loads(capsys.readouterr().out)
    assert out["records"] == 2 and out["nodes"] == 16 and out["checksum"] is True
    assert load_graph_shard(graphs, metadata_path=meta, verify_checksum=True).record_count == 2
    target = tmp_path / "a.json"
    assert main(["alignment-config", "--output", str(target)]) == 0
    assert json.loads(target.read_text())["scoring_parameters"]["sigma"] == 1.0
    # any failure → message on stderr, exit code 2 (reference cli.py:299-301)
    assert main(["build-graphs", "--input", str(tmp_path / "missing.tsv"),
                 "--output", str(graphs)]) == 2
    assert "ginfinity:" in capsys.readouterr().err


def test_shard_text_offsets_and_positional_columns_equal_the_builder(rouskin_records):
    """What the host hands to the device builder (gfy_build_graphs): record offsets from
    lengths and '(' counts, and the numpy float32 sin / cos columns — the same numbers the
    host builder puts into a GraphShard, for whole shards and for record ranges."""
    from ginfinity_amd.graph import shard_text
    records = rouskin_records[:400] + [RNA("one", "A", "."), RNA("two", "AU", "()"),
                                       RNA("three", "ACG", "...")]
    for spec in (GraphSpec.bundled(),
                 GraphSpec(struct_feature="B", positional=True, edge_dim=10, extra_edges=())):
        want = GraphBuilder(spec).build_shard(records)
        text = shard_text(records, spec)
        np.testing.assert_array_equal(text.node_ptr, want.node_ptr)
        np.testing.assert_array_equal(text.edge_ptr, want.edge_ptr)
        assert text.bases.tobytes() == "".join(want.sequences).encode()
        assert text.marks.tobytes() == "".join(want.structures).encode()
        for start, stop in ((0, len(records)), (17, 230), (400, 403)):
            piece = want if (start, stop) == (0, len(records)) else want.slice(start, stop)
            columns = text.positional(start, stop)
            assert columns.tobytes() == np.ascontiguousarray(
                piece.node_features[:, -2:]).tobytes()
    plain = GraphSpec(struct_feature="A", positional=False, edge_dim=10, extra_edges=("skip2",))
    assert shard_text(records, plain).positional(0, 5) is None
    with pytest.raises(GraphValidationError, match="duplicate"):
        shard_text([records[0], records[0]], GraphSpec.bundled())
    with pytest.raises(ValueError, match="unsliced"):
        shard_text([RNA("w", "ACGUACGU", "((....))", start=2, end=5)], GraphSpec.bundled())


def test_parallel_npz_writer_is_read_back_like_savez_compressed(tmp_path):
    """ginfinity_amd.npz.write_npz: the reference CLI's np.savez_compressed archive
    (cli.py:85-88), members deflated on all cores — np.load and zipfile read it back."""
    import zipfile
    from ginfinity_amd.npz import write_npz
    rng = np.random.default_rng(4)
    names = [f"rec/{i}|x y" if i % 7 == 0 else f"r{i}" for i in range(300)]
    arrays = [rng.standard_normal((int(rng.integers(1, 90)), 128)).astype(
        (np.float16, np.float32, np.float64)[i % 3]) for i in range(300)]
    arrays[5] = np.zeros((0, 128), np.float16)
    written = write_npz(tmp_path / "emb", names, arrays, threads=3)
    assert written.name == "emb.npz"
    with zipfile.ZipFile(written) as archive:
        assert archive.testzip() is None
        assert archive.namelist() == [n + ".npy" for n in names]
        assert all(info.compress_type == zipfile.ZIP_DEFLATED for info in archive.infolist())
    with np.load(written) as back:
        assert list(back.keys()) == names
        for name, array in zip(names, arrays):
            got = back[name]
            assert got.dtype == array.dtype and got.shape == array.shape
            assert got.tobytes() == array.tobytes()
    # same payload as numpy's own writer
    plain = [n for n in names if "/" not in n]
    np.savez_compressed(tmp_path / "ref.npz", **{n: arrays[names.index(n)] for n in plain})
    with np.load(tmp_path / "ref.npz") as ref, np.load(written) as back:
        for n in plain:
            assert ref[n].tobytes() == back[n].tobytes()
    with pytest.raises(ValueError, match="duplicate"):
        write_npz(tmp_path / "d.npz", ["a", "a"], arrays[:2])
    # more than 65,535 members: the ZIP64 end-of-directory records
    many = [np.float32([i]) for i in range(66000)]
    big = write_npz(tmp_path / "many.npz", (f"m{i}" for i in range(66000)), many)
    with np.load(big) as back:
        assert len(back.files) == 66000 and float(back["m65999"][0]) == 65999.0


def test_loaded_shard_arrays_are_views_of_the_file_mapping(tmp_path):
    """f3: load_graph_shard maps the safetensors payload instead of reading it — the arrays are
    read-only views of one mmap (no host copy before the pinned upload buffer) and equal what
    safetensors itself loads; a truncated or inconsistent header is refused."""
    from safetensors.numpy import load_file
    from ginfinity_amd import GraphBuilder, load_graph_shard, save_graph_shard
    from ginfinity_amd.shard_io import _map_tensors
    shard = GraphBuilder().build_shard(_records())
    tensor_path, _ = save_graph_shard(shard, tmp_path / "views.safetensors")
    loaded = load_graph_shard(tensor_path)
    reference = load_file(str(tensor_path))
    for name in sorted(reference):     # whole-molecule shards store no residue_index / node_roles
        array = getattr(loaded, name)
        np.testing.assert_array_equal(array, reference[name])
        assert array.dtype == reference[name].dtype and not array.flags.writeable
        base = array
        while getattr(base, "base", None) is not None:
            base = base.base
        assert isinstance(base, (np.memmap, memoryview)) or type(base).__name__ == "mmap", type(base)
    data = tensor_path.read_bytes()
    broken = tmp_path / "broken.safetensors"
    broken.write_bytes(data[:-5])                      # payload shorter than the offsets say
    with pytest.raises(ValueError):
        _map_tensors(broken)
    broken.write_bytes((10 ** 9).to_bytes(8, "little") + data[8:])
    with pytest.raises(ValueError):
        _map_tensors(broken)


def test_whole_shard_window_extraction_equals_the_per_record_builder(rouskin_records):
    """f2: sliced records (windows + crossing-pair partners + context hops) are cut out of the
    whole shard in one pass; the arrays must equal from_graphs(build_many(records)) — the
    reference's per-record construction (graph.py:608-695) — bit for bit, for every option."""
    from ginfinity_amd import RNA, GraphBuilder, GraphShard
    rng = np.random.default_rng(11)
    records = []
    for index, record in enumerate(rouskin_records[:120]):
        length = record.length
        if index % 3 == 0 or length < 12:
            records.append(record)                                   # unsliced among the sliced
            continue
        start = int(rng.integers(0, length - 6))
        end = int(rng.integers(start + 1, min(length, start + 1 + int(rng.integers(3, 90))) + 1))
        records.append(RNA(record.identifier, record.sequence, record.structure,
                           start=start, end=end))
    records.append(RNA("stem", "GGGAAACCCUUUUGGG", "......(((....)))", start=9, end=16))
    assert sum(r.sliced for r in records) > 60
    for keep, hops in ((False, 1), (True, 1), (True, 2), (True, 3)):
        builder = GraphBuilder(keep_paired_neighbours=keep, context_hops=hops)
        want = GraphShard.from_graphs(builder.build_many(records))
        got = builder.build_shard(records)
        for name in ("node_features", "edge_index", "edge_types", "node_ptr", "edge_ptr",
                     "residue_index", "node_roles"):
            a, b = getattr(got, name), getattr(want, name)
            assert a.dtype == b.dtype and a.shape == b.shape, (name, keep, hops)
            np.testing.assert_array_equal(a, b, err_msg=f"{name} keep={keep} hops={hops}")
        assert (got.identifiers, got.sequences, got.structures) == (
            want.identifiers, want.sequences, want.structures)
        if keep:
            assert (got.node_roles != 0).any()


def test_record_boundaries_ride_on_the_edge_index_tensor():
    """``gfy_shard.node_ptr / edge_ptr`` (ABI 4): the Python layer passes a micro-batch's record
    boundaries by attaching them to its edge_index tensor, and only where a record's edge list is
    short against what every 256-row workgroup of the record would have to scan."""
    import torch
    from ginfinity_amd import _native as native
    from ginfinity_amd.engine import MAX_RECORD_EDGES, attach_records, records_of, records_pay
    node_ptr = np.array([0, 4000, 8000, 8064], dtype=np.int64)
    edge_ptr = np.array([0, 20000, 40000, 40300], dtype=np.int64)
    assert records_pay(node_ptr, edge_ptr)
    assert not records_pay(node_ptr, np.array([0, MAX_RECORD_EDGES + 1, MAX_RECORD_EDGES + 2,
                                               MAX_RECORD_EDGES + 3], dtype=np.int64))
    assert not records_pay(node_ptr[:1], edge_ptr[:1])            # no record at all
    assert not records_pay(node_ptr, edge_ptr[:-1])               # inconsistent lengths
    edge_index = torch.zeros((2, 5), dtype=torch.int32)
    assert records_of(edge_index) is None
    attach_records(edge_index, torch.from_numpy(node_ptr), torch.from_numpy(edge_ptr))
    attached = records_of(edge_index)
    assert attached is not None and attached[0].dtype == torch.int64
    assert int(attached[0].numel()) - 1 == 3
    # the descriptor the C ABI takes has the three optional fields at its end (ABI 4)
    names = [name for name, _kind in native.GfyShard._fields_]
    assert names[-3:] == ["node_ptr", "edge_ptr", "n_records"] and native.ABI_VERSION == 4


@pytest.mark.parametrize("stream", ["1", "0"])
def test_native_packer_writes_the_bytes_of_the_numpy_packer(stream, monkeypatch):
    """``gfy_pack_microbatch`` (csrc/gfy_base.cpp: one call without the interpreter lock) against
    ``_pack_microbatch_at`` + ``_Uploader.pack_at`` in numpy: the same offsets, counts and staging
    bytes — whole shards and record ranges in the middle (edge_index rebased), context rows
    (``out_rows``), a record too long for record boundaries, one-record and one-node ranges — and
    the reference's error for an edge that leaves the records' node range on either side
    (graph.py:318-321).  Both forms of its copies: streaming stores (the default) and memcpy
    (``GFY_PACK_STREAM=0``)."""
    import torch
    from ginfinity_amd import api, synthetic
    from ginfinity_amd.spec import GraphValidationError
    monkeypatch.setenv("GFY_PACK_STREAM", stream)

    class Slots(api._Uploader):
        def __init__(self):
            self._staging = [torch.zeros(12 << 20, dtype=torch.uint8) for _ in range(2)]

    long_record = synthetic.roofline_shard(3, records=1, length=16_000)   # 80,000 edges in one record
    cases = [(synthetic.roofline_shard(1, records=5, length=600), [(0, 5), (1, 4), (2, 3), (4, 5)]),
             (synthetic.arbitrary_shard(2, nodes=5000, edges=20000, records=6), [(0, 6), (1, 5), (3, 4)]),
             (synthetic.arbitrary_shard(7, nodes=40, edges=90, records=4, hub_degree=3), [(0, 4), (2, 3)]),
             (long_record, [(0, 1)])]
    slots = Slots()
    for shard, ranges in cases:
        assert api._packable(shard)
        for start, stop in ranges:
            for base in (0, 4096):
                got, want = [], []
                for native_packer, slot, sink in ((True, 0, got), (False, 1, want)):
                    api.NATIVE_PACKER = native_packer
                    slots._staging[slot].zero_()
                    try:
                        sink.append(api.Ginfinity._pack_microbatch_at(slots, slot, base, shard, start, stop))
                    finally:
                        api.NATIVE_PACKER = True
                (offsets_a, *counts_a), (offsets_b, *counts_b) = got[0], want[0]
                assert list(offsets_a) == list(offsets_b) and counts_a == counts_b, (start, stop)
                assert api.Ginfinity._microbatch_bytes(shard, start, stop) >= max(offsets_a) - base
                assert slots._staging[0].numpy().tobytes() == slots._staging[1].numpy().tobytes()
    assert got[0][3] == 0                              # the 80,000-edge record travels without boundaries
    broken = synthetic.roofline_shard(1, records=3, length=100)
    edges = broken.edge_index.copy()
    edges[0, 5] = 250                                   # record 0's edge points into record 2
    edges[1, int(broken.edge_ptr[1]) + 7] = 3           # record 1's edge points into record 0
    object.__setattr__(broken, "edge_index", edges)
    for native_packer in (True, False):
        api.NATIVE_PACKER = native_packer
        try:
            with pytest.raises(GraphValidationError, match="edge index outside"):
                api.Ginfinity._pack_microbatch_at(slots, 0, 0, broken, 0, 1)
            with pytest.raises(GraphValidationError, match="edge index outside"):
                api.Ginfinity._pack_microbatch_at(slots, 0, 0, broken, 1, 2)   # ... and record 1's below it
        finally:
            api.NATIVE_PACKER = True
