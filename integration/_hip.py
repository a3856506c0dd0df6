"""The file a maintainer of the reference would add as ``src/ginfinity/_hip.py`` (see
INTEGRATION.md): the whole reference-side binding of libgfy — ``ctypes``, numpy and torch
(for device buffers) only.  ``tests/test_gpu_integration_stub.py`` executes exactly this file
against this repository's ``GraphShard`` and compares with ``Ginfinity.encode_graphs`` byte
for byte.

Replaces the body of ``Ginfinity._run_graph_shard`` (reference src/ginfinity/api.py:232-260).
"""
import ctypes
import os
import struct

import numpy as np
import torch

_lib = ctypes.CDLL(os.environ.get("GFY_LIBRARY", "libgfy.so"))
_lib.gfy_last_error.restype = ctypes.c_char_p
_lib.gfy_encode_coo_workspace_bytes.restype = ctypes.c_size_t
_P, _I64, _SZ, _I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_size_t, ctypes.c_int
_lib.gfy_encoder_create.argtypes = [_P, _SZ, _I, _I, ctypes.POINTER(_P)]
_lib.gfy_encoder_destroy.argtypes = [_P]
_lib.gfy_encode_coo_workspace_bytes.argtypes = [_P, _I64, _I64]
_lib.gfy_encode_coo.argtypes = [_P, _P, _P, _P, _I64, _I64, _P, _P, _I, _I, _P, _SZ, _P]
_DT = {np.dtype("float16"): (0, torch.float16), np.dtype("float32"): (1, torch.float32),
       np.dtype("float64"): (2, torch.float64)}


def _ok(status):
    if status:
        raise RuntimeError(_lib.gfy_last_error().decode())


def weight_pack(state_dict, cfg):               # order documented in include/gfy.h
    names = ["input.weight", "input.bias"]
    for layer in range(cfg.layers):
        conv = f"convs.{layer}."
        names += [conv + k for k in ("eps", "edge_lin.weight", "edge_lin.bias", "mlp.0.weight",
                                     "mlp.0.bias", "mlp.1.weight", "mlp.1.bias",
                                     "mlp.1.running_mean", "mlp.1.running_var", "mlp.4.weight",
                                     "mlp.4.bias")]
        names += [f"norms.{layer}.weight", f"norms.{layer}.bias"]
    names += ["head.0.weight", "head.0.bias", "head.2.weight", "head.2.bias"]
    head = struct.pack("<8I", 0x31594647, 1, 7, cfg.hidden, cfg.layers, cfg.edge_dim,
                       cfg.out_dim, int(cfg.residual))
    return head + b"".join(state_dict[n].detach().cpu().float().numpy().tobytes() for n in names)


def create(pack: bytes, full_precision: bool, device_index: int):
    handle = _P()
    _ok(_lib.gfy_encoder_create(pack, len(pack), 1 if full_precision else 0, device_index,
                                ctypes.byref(handle)))
    return handle


def destroy(handle):
    _lib.gfy_encoder_destroy(handle)


def run_graph_shard(handle, shard, embedding_dtype, device):   # body of api.py:236-260
    n, e = shard.node_count, shard.edge_count
    put = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)   # noqa: E731
    x, ei, et = put(shard.node_features), put(shard.edge_index), put(shard.edge_types)
    # a fresh, ZEROED workspace per call keeps the stub short (gfy.h: the leading counters
    # must be zero); a real binding keeps one per shape, as ginfinity_amd/engine.py does
    ws = torch.zeros(_lib.gfy_encode_coo_workspace_bytes(handle, n, e), dtype=torch.uint8,
                     device=device)
    core = shard.node_roles == 0
    rows = put(np.where(core, np.cumsum(core) - 1, -1).astype(np.int32))
    code, tdtype = _DT[np.dtype(embedding_dtype)]
    out = torch.empty((int(core.sum()), 128), dtype=tdtype, device=device)
    _ok(_lib.gfy_encode_coo(handle, x.data_ptr(), ei.data_ptr() if e else None,
                            et.data_ptr() if e else None, n, e, rows.data_ptr(),
                            out.data_ptr(), code, 1, ws.data_ptr(), ws.numel(),
                            torch.cuda.current_stream(device).cuda_stream))
    return np.split(out.cpu().numpy(), np.cumsum(shard.core_counts)[:-1])
