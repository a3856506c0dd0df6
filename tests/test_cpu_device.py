"""``device="cpu"``: the reference's default device, served by the host implementation inside
libgfy (csrc/gine_host.cpp).  Runs without a GPU.

The expected values are the reference's own (tests/golden/*.npz, recorded by importing
/root/reference/src; tests/golden/io/* written by the reference's CLI).  Mirrors the
reference's tests/test_api.py:14-45 (load, encode, dtype handling, device policy).
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import pytest

GOLDEN = Path(__file__).resolve().parent / "golden"
IO = GOLDEN / "io"
F16_TOL, F32_TOL = 1e-3, 1e-6          # BASELINE.json north_star tolerances


def _maxabs(a, b) -> float:
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


@pytest.fixture(scope="module")
def cpu_encoder():
    from ginfinity_amd import Ginfinity
    return Ginfinity.load()                       # the reference's default: device="cpu"


@pytest.fixture(scope="module")
def cpu_encoder_fp32():
    from ginfinity_amd import Ginfinity
    return Ginfinity.load("cpu", full_precision=True)


def test_default_device_is_cpu_like_the_reference(cpu_encoder):
    assert cpu_encoder.device == "cpu" and cpu_encoder.full_precision is False
    assert cpu_encoder.embedding_dimension == 128
    assert cpu_encoder.info()["model_version"]


def test_device_policy_is_the_reference_s():
    """api.py:69-76: anything but 'cpu' / 'cuda*' is refused; CUDA needs the explicit
    acknowledgement and an available device."""
    from ginfinity_amd import Ginfinity
    with pytest.raises(ValueError, match="device must be 'cpu' or a CUDA device"):
        Ginfinity.load("tpu")
    with pytest.raises(ValueError, match="CUDA requires allow_nondeterministic_cuda=True"):
        Ginfinity.load("cuda")
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(ValueError, match="CUDA was requested but is unavailable"):
            Ginfinity.load("cuda", allow_nondeterministic_cuda=True)


def test_readme_example_on_cpu_matches_the_reference(cpu_encoder, cpu_encoder_fp32, golden):
    """BASELINE configs[0]: encode() on the 8-nt README example via Ginfinity.load() on CPU."""
    from ginfinity_amd import RNA
    g = golden("example8.npz")
    record = RNA("example", "ACGUACGU", "((....))")
    out = cpu_encoder.encode(record)
    assert out.shape == (8, 128) and out.dtype == np.float16 and out.flags.c_contiguous
    assert _maxabs(out, g["out.m16.float16"]) <= F16_TOL
    np.testing.assert_allclose(np.linalg.norm(out.astype(np.float64), axis=1), 1.0, atol=2e-3)
    for dtype in ("float32", "float64"):
        got = cpu_encoder.encode(record, embedding_dtype=dtype)
        assert got.dtype == np.dtype(dtype)
        assert _maxabs(got, g[f"out.m16.{dtype}"]) <= F16_TOL
    full = cpu_encoder_fp32.encode(record, embedding_dtype="float64")
    assert _maxabs(full, g["out.m32.float64"]) <= F32_TOL
    assert cpu_encoder.encode(record).tobytes() == out.tobytes()      # deterministic


def test_cpu_encode_many_matches_the_reference_archive(cpu_encoder):
    """24 rouskin records against the archive the reference's ``ginfinity embed`` wrote."""
    from ginfinity_amd import read_rna_table
    records = read_rna_table(IO / "small.tsv")
    reference = np.load(IO / "ref_embed.npz")
    outputs = cpu_encoder.encode_many(records, max_batch_nodes=1000, max_batch_edges=5000)
    assert [r.identifier for r in records] == list(reference.files)
    worst = max(_maxabs(out, reference[r.identifier]) for r, out in zip(records, outputs))
    assert worst <= F16_TOL, worst
    assert all(out.shape == reference[r.identifier].shape for r, out in zip(records, outputs))


def test_cpu_sliced_records_drop_context_rows_like_the_reference(cpu_encoder):
    from ginfinity_amd import read_rna_table
    records = read_rna_table(IO / "windowed.tsv")
    reference = np.load(IO / "ref_windowed_embed.npz")
    outputs = cpu_encoder.encode_many(records, keep_paired_neighbours=True, context_hops=2)
    for record, out in zip(records, outputs):
        want = reference[record.identifier]
        assert out.shape == want.shape and _maxabs(out, want) <= F16_TOL


def test_cpu_encode_graphs_on_a_reference_written_shard(cpu_encoder):
    from ginfinity_amd import load_graph_shard
    shard = load_graph_shard(IO / "ref_shard.safetensors", verify_checksum=True,
                             validation="full", expected_spec=cpu_encoder.graph_spec)
    reference = np.load(IO / "ref_embed_graphs.npz")
    outputs = cpu_encoder.encode_graphs(shard)
    for identifier, out in zip(shard.identifiers, outputs):
        assert _maxabs(out, reference[identifier]) <= F16_TOL
    assert cpu_encoder.encode_graphs([]) == []


def test_cli_embed_on_cpu_writes_the_reference_s_archive_and_manifest(tmp_path):
    """``ginfinity embed`` with the reference's defaults (device cpu): the archive's members
    within tolerance of the reference-written one, the manifest with the same keys."""
    from ginfinity_amd import cli
    out = tmp_path / "embed.npz"
    assert cli.main(["embed", "--input", str(IO / "small.tsv"), "--output", str(out)]) == 0
    ours, theirs = np.load(out), np.load(IO / "ref_embed.npz")
    assert list(ours.files) == list(theirs.files)
    assert max(_maxabs(ours[k], theirs[k]) for k in theirs.files) <= F16_TOL
    mine = json.loads(out.with_suffix(".manifest.json").read_text())
    reference = json.loads((IO / "ref_embed.manifest.json").read_text())
    assert set(mine) == set(reference)
    assert mine["device"] == "cpu" and mine["status"] == "complete"
    assert [set(r) for r in mine["records"]] == [set(r) for r in reference["records"]]
    for a, b in zip(mine["records"], reference["records"]):
        assert {k: a[k] for k in a if "sha256" not in k} == {k: b[k] for k in b if "sha256" not in k}


def test_cli_embed_graphs_on_cpu_from_the_reference_s_shard(tmp_path):
    from ginfinity_amd import cli
    out = tmp_path / "graphs.npz"
    assert cli.main(["embed-graphs", "--input", str(IO / "ref_shard.safetensors"),
                     "--output", str(out), "--verify-checksum", "--full-validation",
                     "--checksum"]) == 0
    ours, theirs = np.load(out), np.load(IO / "ref_embed_graphs.npz")
    assert list(ours.files) == list(theirs.files)
    assert max(_maxabs(ours[k], theirs[k]) for k in theirs.files) <= F16_TOL
    mine = json.loads(out.with_suffix(".manifest.json").read_text())
    reference = json.loads((IO / "ref_embed_graphs.manifest.json").read_text())
    assert set(mine) == set(reference)
    assert mine["records"] == reference["records"]
    assert mine["graph_spec_sha256"] == reference["graph_spec_sha256"]


def test_loaded_parameters_have_the_compute_dtype_and_hashing_stays_opt_in(tmp_path, monkeypatch):
    """Two things the reference's own tests reach for below the public surface
    (tests/test_api.py:24,29: ``next(encoder._model.parameters()).dtype``;
    tests/test_graph.py:75-86: ``graph._sha256`` replaced to prove that loading a shard with
    the default options does not hash its tensors).  `tools/run_reference_tests.sh` runs those
    test files themselves against this package where the reference tree exists."""
    import torch
    from ginfinity_amd import (Ginfinity, GraphBuilder, RNA, graph, load_graph_shard,
                               save_graph_shard)
    half, full = Ginfinity.load(), Ginfinity.load(full_precision=True)
    assert next(half._model.parameters()).dtype == torch.float16
    assert next(full._model.parameters()).dtype == torch.float32
    assert sum(p.numel() for p in half._model.parameters()) == half.info()["parameter_count"]
    shard = GraphBuilder().build_shard([RNA("a", "ACGUACGU", "((....))")])
    tensor_path, _ = save_graph_shard(shard, tmp_path / "one.safetensors", checksum=True)

    def unexpected(_path):
        raise AssertionError("content hashing must remain opt-in")

    monkeypatch.setattr(graph, "_sha256", unexpected)
    loaded = load_graph_shard(tensor_path)                  # default: no hashing
    assert loaded.identifiers == ("a",)
    with pytest.raises(AssertionError, match="opt-in"):
        load_graph_shard(tensor_path, verify_checksum=True)


def test_cpu_device_loads_only_the_host_library():
    """device="cpu" (the reference's default, api.py:64-76) must work where no ROCm runtime
    exists: it is served by libgfy_host.so — gine_host.cpp + gfy_base.cpp built with the host
    compiler, no HIP dependency — and never maps libgfy.so.  In a fresh interpreter."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    script = (
        "import sys; sys.path.insert(0, %r)\n"
        "from ginfinity_amd import Ginfinity, RNA\n"
        "out = Ginfinity.load().encode(RNA('example', 'ACGUACGU', '((....))'))\n"
        "maps = open('/proc/self/maps').read()\n"
        "assert out.shape == (8, 128), out.shape\n"
        "assert 'libgfy_host.so' in maps\n"
        "assert 'csrc/libgfy.so' not in maps, 'the CPU device mapped the HIP library'\n"
        "print('ok')\n" % str(root))
    done = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True,
                          timeout=300)
    assert done.returncode == 0 and done.stdout.strip() == "ok", done.stderr[-2000:]
    needed = subprocess.run(["readelf", "-d", str(root / "ginfinity_amd" / "csrc" /
                                                  "libgfy_host.so")],
                            capture_output=True, text=True).stdout
    assert "NEEDED" in needed and "hip" not in needed.lower() and "hsa" not in needed.lower()


def test_host_library_builds_without_hipcc(tmp_path, monkeypatch):
    """``python -m ginfinity_amd.build --host-only`` needs the host compiler and nothing else: with
    HIPCC pointing nowhere (and no hipcc on PATH) the host library still builds, and the full
    build says which library it could not make instead of dying in a compile job."""
    import importlib
    import shutil
    build = importlib.import_module("ginfinity_amd.build")
    monkeypatch.setenv("HIPCC", str(tmp_path / "no-such-hipcc"))
    monkeypatch.setattr(build, "_hipcc", lambda: None)
    monkeypatch.setattr(build, "HOST_LIBRARY", tmp_path / "libgfy_host.so")
    assert shutil.which(build.os.environ.get("CXX", "g++")), "no host compiler in this container"
    built = build.build(host_only=True)
    assert built == tmp_path / "libgfy_host.so" and built.stat().st_size > 10_000
    with pytest.raises(RuntimeError, match="--host-only"):
        build.build()          # the GPU library: a clear message, after the host library was built


def test_cpu_device_error_text_survives_the_gpu_library():
    """Both libraries define gfy::set_error; libgfy.so is loaded RTLD_GLOBAL.  The host library
    binds its own copy (-Bsymbolic), so a CPU-device failure keeps its message when the GPU
    library is already in the process.  In a fresh interpreter (load order matters)."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    script = (
        "import sys, ctypes; sys.path.insert(0, %r)\n"
        "from ginfinity_amd import _native as native\n"
        "gpu = native.library()\n"                       # maps libgfy.so first (no GPU needed)
        "host = native.host_library()\n"
        "handle = ctypes.c_void_p()\n"
        "rc = host.gfy_host_encoder_create(b'xxxx', 4, native.GFY_F16, ctypes.byref(handle))\n"
        "assert rc != 0\n"
        "text = host.gfy_last_error().decode()\n"
        "assert 'weight pack' in text, repr(text)\n"
        "try:\n"
        "    native.check(rc, 'gfy_host_encoder_create', host)\n"
        "except Exception as error:\n"
        "    assert 'weight pack' in str(error), str(error)\n"
        "print('ok')\n" % str(root))
    done = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True,
                          timeout=300)
    assert done.returncode == 0 and done.stdout.strip() == "ok", done.stderr[-2000:]


def test_gfy_library_names_the_build_that_is_opened(tmp_path):
    """The GPU library is opened by its path (LD_LIBRARY_PATH does not redirect it); GFY_LIBRARY
    names another build for a side-by-side measurement, and a path that is not there fails loudly
    instead of falling back to the in-tree build."""
    import os
    import shutil
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    in_tree = root / "ginfinity_amd" / "csrc" / "libgfy.so"
    copy = tmp_path / "libgfy_copy.so"
    shutil.copy(in_tree, copy)
    script = (
        "import sys; sys.path.insert(0, %r)\n"
        "from ginfinity_amd import _native as native\n"
        "lib = native.library()\n"
        "maps = open('/proc/self/maps').read()\n"
        "print(native.LIBRARY_PATH, 'libgfy_copy.so' in maps, 'csrc/libgfy.so' in maps)\n" % str(root))
    done = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True,
                          timeout=300, env={**os.environ, "GFY_LIBRARY": str(copy)})
    assert done.returncode == 0, done.stderr[-2000:]
    assert done.stdout.split() == [str(copy), "True", "False"]
    missing = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True,
                             timeout=300, env={**os.environ, "GFY_LIBRARY": str(tmp_path / "nope.so")})
    assert missing.returncode != 0 and "nope.so is missing" in missing.stderr
