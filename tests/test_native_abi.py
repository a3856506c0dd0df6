"""The C-ABI library loads and exports every symbol include/gfy.h declares;
argument validation that needs no GPU returns the documented status codes."""
from __future__ import annotations

import ctypes
import re
from pathlib import Path

import pytest

from ginfinity_amd import _native as native
from ginfinity_amd.build import CSRC, LIBRARY, SOURCES

HEADER = Path(__file__).resolve().parents[1] / "include" / "gfy.h"


def _declared_symbols() -> set[str]:
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return set(re.findall(r"\b(gfy_[a-z0-9_]+)\s*\(", text))


def test_library_is_built_in_tree():
    assert LIBRARY.is_file(), "run `python -m ginfinity_amd.build`"
    for name in SOURCES:
        assert (CSRC / name).is_file()


def test_every_declared_symbol_is_exported_and_bound():
    lib = native.library()
    declared = _declared_symbols()
    assert declared == set(native.SIGNATURES), declared ^ set(native.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.gfy_abi_version() == native.ABI_VERSION


def test_weight_pack_size_matches_python_builder(checkpoint):
    lib = native.library()
    assert lib.gfy_weight_pack_bytes(7, 128, 4, 10, 128) == len(checkpoint.weight_pack)


def test_argument_errors_without_a_gpu(checkpoint):
    lib = native.library()
    handle = ctypes.c_void_p()
    status = lib.gfy_encoder_create(b"short", 5, native.GFY_F16, 0, ctypes.byref(handle))
    assert status == native.GFY_ERR_INVALID and not handle.value
    assert b"truncated" in lib.gfy_last_error()
    bad = bytearray(checkpoint.weight_pack)
    bad[0] ^= 0xFF
    status = lib.gfy_encoder_create(bytes(bad), len(bad), native.GFY_F16, 0, ctypes.byref(handle))
    assert status == native.GFY_ERR_INVALID
    pack = bytearray(checkpoint.weight_pack)
    pack[12:16] = (64).to_bytes(4, "little")          # hidden = 64: not compiled in
    status = lib.gfy_encoder_create(bytes(pack), len(pack), native.GFY_F16, 0, ctypes.byref(handle))
    assert status == native.GFY_ERR_UNSUPPORTED
    with pytest.raises(ValueError, match="bad arguments"):
        native.check(lib.gfy_pairwise_nearest(None, 0, None, 0, 0, -1, None, None, None, 0, None),
                     "gfy_pairwise_nearest")
    assert lib.gfy_encode(None, None, None, None, None, 1, 0, None, None, 0, 1, None, 0,
                          None) == native.GFY_ERR_INVALID
    assert lib.gfy_csr_workspace_bytes(60_000, 300_000) > 2 * 4 * 300_000
    # the upload ring refuses what it can without a device
    ring = ctypes.c_void_p()
    for slots in (0, 65):
        assert lib.gfy_upload_ring_create(slots, ctypes.byref(ring)) == native.GFY_ERR_INVALID
        assert b"1..64" in lib.gfy_last_error() and not ring.value
    assert lib.gfy_upload_ring_create(8, None) == native.GFY_ERR_INVALID
    assert lib.gfy_upload_async(None, 0, None, None, 0, None, None) == native.GFY_ERR_INVALID
    assert lib.gfy_upload_wait(None, 0) == native.GFY_ERR_INVALID
    lib.gfy_upload_ring_destroy(None)


def test_layer_kernel_keeps_its_register_and_scratch_budget():
    """The fused layer kernel must not spill: a scratch reload is a `vmcnt` wait that drains the
    LDS-DMA look-ahead, and more than 256 VGPRs would halve the waves per SIMD (two per SIMD
    is what one 512-thread workgroup per CU needs).  hipcc cross-compiles, no GPU needed."""
    import subprocess
    script = Path(__file__).resolve().parents[1] / "tools" / "kernel_resources.sh"
    done = subprocess.run(["bash", str(script)], capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    layers = [line for line in done.stdout.splitlines() if "k_gine_layer_f16" in line]
    assert len(layers) == 4, done.stdout            # <kResidual, kHead> x 2 x 2
    for line in layers:
        fields = line.split()
        vgprs = int(fields[fields.index("vgpr") + 1])
        spilled = int(fields[fields.index("spilled") + 1])
        scratch = int(fields[fields.index("scratch") + 1])
        assert vgprs <= 256 and spilled == 0 and scratch == 0, line
