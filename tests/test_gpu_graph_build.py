"""gfy_build_graphs (device GraphBuilder for unsliced records) against the host builder —
itself pinned to the reference's arrays by SHA-256 (tests/test_host_layer.py) — bit for bit,
and the encode_many path that uses it against encode_graphs on host-built shards."""
from __future__ import annotations

import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def gpu_encoder():
    from ginfinity_amd import Ginfinity
    return Ginfinity.load("cuda:0", allow_nondeterministic_cuda=True)


def _random_structure(rng, length, pair_bias=0.45):
    """A balanced dot-bracket string of exactly ``length`` characters."""
    out, depth = [], 0
    for position in range(length):
        left = length - position
        if depth == left:
            out.append(")"); depth -= 1
        elif depth + 2 <= left and rng.random() < pair_bias:
            out.append("("); depth += 1
        elif depth and rng.random() < 0.4:
            out.append(")"); depth -= 1
        else:
            out.append(".")
    assert depth == 0
    return "".join(out)


def _records(rng):
    from ginfinity_amd import RNA
    texts = ["A|.", "AC|..", "ACG|...", "AU|()", "ACGU|(..)", "GCAU|()()",
             "G" * 130 + "|" + "()" * 65,                           # one level, every step
             "A" * 4096 + "|" + "(" * 2048 + ")" * 2048,       # the deepest nest a legal record has
             "G" * 3000 + "|" + ("(" * 1100 + "." + ")" * 50 + "(" * 50 + ")" * 1100) + "." * 699,
             "C" * 400 + "|" + ("(" * 100 + ")" * 100) * 2,
             "U" * 127 + "|" + "(" * 63 + "." + ")" * 63,            # pairs across a 64-step seam
             "U" * 129 + "|" + "(" * 64 + "." + ")" * 64]
    records = []
    for index, text in enumerate(texts):
        sequence, structure = text.split("|")
        records.append(RNA(f"hand{index}", sequence, structure))
    for index in range(300):
        length = int(rng.integers(1, 700))
        sequence = "".join(rng.choice(list("ACGU"), size=length))
        records.append(RNA(f"rand{index}", sequence, _random_structure(rng, length)))
    order = rng.permutation(len(records))
    return [records[i] for i in order]


def _device_build(engine, text, start, stop):
    device = engine.device
    n0, n1 = int(text.node_ptr[start]), int(text.node_ptr[stop])
    e0, e1 = int(text.edge_ptr[start]), int(text.edge_ptr[stop])
    columns = text.positional(start, stop)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)   # noqa: E731
    x, ei, et, bad = engine.build_graphs(
        up(text.bases[n0:n1]), up(text.marks[n0:n1]), up(text.node_ptr[start:stop + 1]),
        up(text.edge_ptr[start:stop + 1]), None if columns is None else up(columns),
        n1 - n0, e1 - e0, struct_states=1 if text.spec.struct_feature == "A" else 3,
        skip2=text.spec.has_skip2)
    torch.cuda.synchronize()
    return x.cpu().numpy(), ei.cpu().numpy(), et.cpu().numpy(), int(bad.item())


@pytest.mark.parametrize("variant", ["bundled", "three_state_no_skip", "flag_only"])
def test_device_builder_equals_host_builder(gpu_encoder, variant):
    from ginfinity_amd import GraphBuilder, GraphSpec
    from ginfinity_amd.graph import shard_text
    spec = {"bundled": GraphSpec.bundled(),
            "three_state_no_skip": GraphSpec(struct_feature="B", positional=True,
                                             edge_dim=10, extra_edges=()),
            "flag_only": GraphSpec(struct_feature="A", positional=False, edge_dim=10,
                                   extra_edges=("skip2",))}[variant]
    records = _records(np.random.default_rng(11))
    want = GraphBuilder(spec).build_shard(records)
    text = shard_text(records, spec)
    np.testing.assert_array_equal(text.node_ptr, want.node_ptr)
    np.testing.assert_array_equal(text.edge_ptr, want.edge_ptr)
    x, ei, et, bad = _device_build(gpu_encoder._engine, text, 0, len(records))
    assert bad == -1
    assert x.dtype == want.node_features.dtype and x.shape == want.node_features.shape
    assert x.tobytes() == want.node_features.tobytes()
    np.testing.assert_array_equal(ei, want.edge_index)
    np.testing.assert_array_equal(et, want.edge_types)
    # a slice in the middle: offsets are rebased to the slice like GraphShard.slice
    piece = want.slice(40, 97)
    x, ei, et, bad = _device_build(gpu_encoder._engine, text, 40, 97)
    assert bad == -1 and x.tobytes() == piece.node_features.tobytes()
    np.testing.assert_array_equal(ei, piece.edge_index)
    np.testing.assert_array_equal(et, piece.edge_types)


def test_device_builder_reproduces_the_reference_hashes(gpu_encoder, golden, rouskin_records):
    """The SHA-256 digests recorded from the genuine reference builder on the 5,840-record
    rouskin sample (tests/golden/integers.json)."""
    from ginfinity_amd.graph import shard_text
    want = golden("integers.json")["rouskin"]
    text = shard_text(rouskin_records, gpu_encoder.graph_spec)
    x, ei, et, bad = _device_build(gpu_encoder._engine, text, 0, len(rouskin_records))
    assert bad == -1 and (x.shape[0], et.shape[0]) == (want["nodes"], want["edges"])
    got = {"node_features": x, "edge_index": ei, "edge_types": et,
           "node_ptr": text.node_ptr, "edge_ptr": text.edge_ptr}
    checked = 0
    for name, digest in want["sha256"].items():
        if name in got:
            assert hashlib.sha256(np.ascontiguousarray(got[name]).tobytes()).hexdigest() \
                == digest, name
            checked += 1
    assert checked >= 3


def test_device_builder_flags_the_first_invalid_record(gpu_encoder):
    """Text that RNA() would have refused (fed past it on purpose): nothing is written out
    of bounds and the first offending record is reported."""
    from ginfinity_amd import RNA, GraphSpec
    from ginfinity_amd.graph import shard_text
    records = [RNA("a", "ACGU", "(..)"), RNA("b", "ACGUAC", "((..))"),
               RNA("c", "GGGAAACCC", "(((...)))"), RNA("d", "AC", "..")]
    spec = GraphSpec.bundled()
    for damage, expect in (((1, ")"), 0), ((5, ")"), 1), ((8, "("), 1), ((10, "x"), 2)):
        text = shard_text(records, spec)
        text.marks[damage[0]] = ord(damage[1])
        assert _device_build(gpu_encoder._engine, text, 0, 4)[3] == expect
    text = shard_text(records, spec)
    text.bases[17] = ord("N")
    assert _device_build(gpu_encoder._engine, text, 0, 4)[3] == 2
    text = shard_text(records, spec)
    text.edge_ptr[2:] += 2                     # record 1 claims one pair too many
    assert _device_build(gpu_encoder._engine, text, 0, 4)[3] == 1
    # a nest deeper than 2,048 levels cannot come from a legal record (4,096 nt at most);
    # text assembled past RNA() is reported, not mis-built
    from ginfinity_amd.graph import ShardText
    depth = 2049
    marks = np.frombuffer(("(..)" + "(" * depth + "." * 64 + ")" * depth).encode(),
                          np.uint8).copy()
    length = 2 * depth + 64
    deep = ShardText(np.full(marks.size, ord("G"), np.uint8), marks,
                     np.array([0, 4, marks.size], np.int64),
                     np.array([0, 12, 12 + 2 * (length - 1) + 2 * depth
                               + 2 * (length - 2)], np.int64), spec)
    assert _device_build(gpu_encoder._engine, deep, 0, 2)[3] == 1


@pytest.mark.parametrize("dtype", [np.float16, np.float32])
def test_encode_many_on_device_built_graphs_equals_encode_graphs(gpu_encoder, dtype,
                                                                 rouskin_records):
    from ginfinity_amd import GraphBuilder
    records = rouskin_records[:700] + _records(np.random.default_rng(5))[:40]
    shard = GraphBuilder().build_shard(records)
    want = gpu_encoder.encode_graphs(shard, embedding_dtype=dtype)
    got = gpu_encoder.encode_many(records, embedding_dtype=dtype)
    assert len(got) == len(want) == len(records)
    for a, b, record in zip(got, want, records):
        assert a.dtype == b.dtype and a.shape == (record.length, 128)
        assert a.tobytes() == b.tobytes()
    # limits and errors as encode_graphs has them
    with pytest.raises(ValueError, match="max_batch_nodes is smaller"):
        gpu_encoder.encode_many(records, max_batch_nodes=100)
    small = gpu_encoder.encode_many(records[:50], max_batch_nodes=7000, max_batch_edges=31000,
                                    embedding_dtype=dtype)
    differing = [i for i, (a, b) in enumerate(zip(small, want[:50]))
                 if a.dtype != b.dtype or a.tobytes() != b.tobytes()]
    assert not differing, differing


def test_cli_embed_and_embed_graphs_write_the_reference_archives(tmp_path, capsys,
                                                                 gpu_encoder, rouskin_records):
    """`embed` (text in, device-built graphs) and `build-graphs` + `embed-graphs` produce the
    same .npz — member per record, as np.savez_compressed wrote them in the reference
    (cli.py:85-88,158-163) — plus the manifests."""
    import json
    from ginfinity_amd.cli import main
    records = rouskin_records[:120]
    table = tmp_path / "rna.tsv"
    table.write_text("transcript_id\tsequence\tsecondary_structure\n" + "".join(
        f"{r.identifier}\t{r.sequence}\t{r.structure}\n" for r in records))
    direct = tmp_path / "direct.npz"
    gpu = ["--device", "cuda", "--allow-nondeterministic-cuda"]
    assert main(["embed", "--input", str(table), "--output", str(direct), *gpu]) == 0
    printed = json.loads(capsys.readouterr().out)
    assert printed["records"] == 120
    manifest = json.loads((tmp_path / "direct.manifest.json").read_text())
    assert manifest["status"] == "complete" and len(manifest["records"]) == 120
    assert manifest["output_sha256"] and manifest["records"][3]["shape"][1] == 128

    graphs = tmp_path / "g.safetensors"
    assert main(["build-graphs", "--input", str(table), "--output", str(graphs)]) == 0
    capsys.readouterr()
    staged = tmp_path / "staged.npz"
    assert main(["embed-graphs", "--input", str(graphs), "--output", str(staged),
                 "--embedding-dtype", "float16", *gpu]) == 0
    want = gpu_encoder.encode_many(records)
    with np.load(direct) as a, np.load(staged) as b:
        assert list(a.keys()) == [r.identifier for r in records] == list(b.keys())
        for record, expected in zip(records, want):
            assert a[record.identifier].tobytes() == expected.tobytes()
            assert b[record.identifier].tobytes() == expected.tobytes()


# ---- f3 on the GPU against the files the REFERENCE wrote (tests/golden/io, make_io_golden.py) ----
IO = Path(__file__).resolve().parent / "golden" / "io"


def _archives_agree(ours_path, reference_path, tolerance=1e-3):
    with np.load(ours_path) as ours, np.load(reference_path) as theirs:
        assert list(ours.files) == list(theirs.files)
        for key in theirs.files:
            assert ours[key].shape == theirs[key].shape and ours[key].dtype == theirs[key].dtype
            worst = np.abs(ours[key].astype(np.float64) - theirs[key].astype(np.float64)).max()
            assert worst <= tolerance, (key, worst)


def _manifests_agree(ours_path, reference_path, *, device):
    mine = json.loads(Path(ours_path).read_text())
    reference = json.loads(Path(reference_path).read_text())
    assert set(mine) == set(reference)
    assert mine["device"] == device and mine["status"] == "complete"
    assert [set(r) for r in mine["records"]] == [set(r) for r in reference["records"]]
    for a, b in zip(mine["records"], reference["records"]):
        assert ({k: a[k] for k in a if "sha256" not in k}
                == {k: b[k] for k in b if "sha256" not in k})
    return mine, reference


def test_cli_embed_graphs_on_the_gpu_from_the_reference_s_shard(tmp_path, capsys):
    """``ginfinity embed-graphs --device cuda`` on the shard file the REFERENCE's
    ``build-graphs`` wrote (graph.py:756-823; loaded with checksum and full validation,
    graph.py:826-923) against the archive and manifest the reference's ``embed-graphs`` wrote
    (cli.py:138-197): members within 1e-3, same manifest keys and per-record entries."""
    from ginfinity_amd import cli
    out = tmp_path / "graphs.npz"
    assert cli.main(["embed-graphs", "--input", str(IO / "ref_shard.safetensors"),
                     "--output", str(out), "--verify-checksum", "--full-validation",
                     "--checksum", "--device", "cuda", "--allow-nondeterministic-cuda"]) == 0
    capsys.readouterr()
    _archives_agree(out, IO / "ref_embed_graphs.npz")
    mine, reference = _manifests_agree(out.with_suffix(".manifest.json"),
                                       IO / "ref_embed_graphs.manifest.json", device="cuda")
    assert mine["graph_spec_sha256"] == reference["graph_spec_sha256"]


def test_cli_embed_on_the_gpu_against_the_reference_s_archives(tmp_path, capsys):
    """``ginfinity embed --device cuda`` (cli.py:69-111) on the tables the reference embedded:
    whole molecules (graphs built on the device) and windowed records with paired neighbours
    and two context hops (context rows dropped at the head's store)."""
    from ginfinity_amd import cli
    gpu = ["--device", "cuda", "--allow-nondeterministic-cuda"]
    out = tmp_path / "embed.npz"
    assert cli.main(["embed", "--input", str(IO / "small.tsv"), "--output", str(out), *gpu]) == 0
    _archives_agree(out, IO / "ref_embed.npz")
    _manifests_agree(out.with_suffix(".manifest.json"), IO / "ref_embed.manifest.json",
                     device="cuda")
    windowed = tmp_path / "windowed.npz"     # the flags make_io_golden.py gave the reference
    arguments = ["embed", "--input", str(IO / "windowed.tsv"), "--output", str(windowed), *gpu,
                 "--keep-paired-neighbours", "--context-hops", "2"]
    assert cli.main(arguments) == 0
    capsys.readouterr()
    _archives_agree(windowed, IO / "ref_windowed_embed.npz")
