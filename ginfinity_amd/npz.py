"""Embedding archives: the ``.npz`` the reference's CLI writes with
``np.savez_compressed`` (src/ginfinity/cli.py:85-88,158-163), deflated on all cores.

At GPU speed the archive is the end-to-end cost of ``ginfinity embed``: zlib on one core
does ≈23 MB/s on fp16 embeddings (they barely compress: 0.93), i.e. ≈10 s for the 230 MB of
a 900k-nucleotide shard that the encoder produces in 17 ms.  An ``.npz`` is a ZIP of ``.npy``
members; the members are independent, so they are serialised, checksummed and deflated by a
thread pool (zlib releases the GIL) and this module writes the ZIP records itself.  The file
is read back by ``np.load`` / ``zipfile`` like any other ``.npz`` — same member names, same
``.npy`` payloads, same deflate level; only the compressed bytes are not promised to be
identical to numpy's.
"""
from __future__ import annotations

import io
import os
import struct
import time
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Iterable

import numpy as np

_LOCAL, _CENTRAL, _END = 0x04034B50, 0x02014B50, 0x06054B50
_END64, _LOCATOR64 = 0x06064B50, 0x07064B50
_LIMIT32, _LIMIT16 = 0xFFFFFFFF, 0xFFFF
_DEFLATED = 8


def _member(name: str, array: np.ndarray) -> tuple[bytes, int, int, bytes]:
    """(member name, crc32, uncompressed size, raw-deflate stream) of one array."""
    buffer = io.BytesIO()
    np.lib.format.write_array(buffer, np.asanyarray(array), allow_pickle=False)
    payload = buffer.getbuffer()
    deflater = zlib.compressobj(zlib.Z_DEFAULT_COMPRESSION, zlib.DEFLATED, -15)
    packed = deflater.compress(payload) + deflater.flush()
    return (name + ".npy").encode("utf-8"), zlib.crc32(payload), len(payload), packed


def _dos_time(moment: float) -> tuple[int, int]:
    t = time.localtime(moment)
    year = min(max(t.tm_year, 1980), 2107)
    return ((t.tm_hour << 11) | (t.tm_min << 5) | (t.tm_sec // 2),
            ((year - 1980) << 9) | (t.tm_mon << 5) | t.tm_mday)


def write_npz(path: str | os.PathLike, names: Iterable[str], arrays: Iterable[np.ndarray],
              *, threads: int | None = None) -> Path:
    """``np.savez_compressed(path, **dict(zip(names, arrays)))`` with the members deflated in
    parallel.  Like numpy, appends ``.npz`` to a path that lacks it; unlike numpy, member
    names need not be valid keyword arguments.  Returns the path written."""
    path = Path(path)
    if path.suffix != ".npz":
        path = path.with_name(path.name + ".npz")
    names, arrays = list(names), list(arrays)
    if len(names) != len(arrays):
        raise ValueError("names and arrays differ in length")
    if len(set(names)) != len(names):
        raise ValueError("duplicate member names")
    workers = threads or min(32, os.cpu_count() or 1)
    clock, date = _dos_time(time.time())
    directory = []
    with open(path, "wb") as out:
        with ThreadPoolExecutor(max_workers=workers) as pool:
            for name, crc, size, packed in pool.map(_member, names, arrays, chunksize=8):
                offset = out.tell()
                big = size >= _LIMIT32 or len(packed) >= _LIMIT32
                extra = struct.pack("<HHQQ", 1, 16, size, len(packed)) if big else b""
                out.write(struct.pack(
                    "<IHHHHHIIIHH", _LOCAL, 45 if big else 20, 0x800, _DEFLATED, clock, date,
                    crc, _LIMIT32 if big else len(packed), _LIMIT32 if big else size,
                    len(name), len(extra)))
                out.write(name)
                out.write(extra)
                out.write(packed)
                directory.append((name, crc, size, len(packed), offset))
        start = out.tell()
        for name, crc, size, packed_size, offset in directory:
            fields = []
            if size >= _LIMIT32 or packed_size >= _LIMIT32:
                fields += [size, packed_size]
            if offset >= _LIMIT32:
                fields.append(offset)
            extra = (struct.pack("<HH" + "Q" * len(fields), 1, 8 * len(fields), *fields)
                     if fields else b"")
            wide = size >= _LIMIT32 or packed_size >= _LIMIT32
            out.write(struct.pack(
                "<IHHHHHHIIIHHHHHII", _CENTRAL, 45, 45 if fields else 20, 0x800, _DEFLATED,
                clock, date, crc, _LIMIT32 if wide else packed_size,
                _LIMIT32 if wide else size, len(name), len(extra), 0, 0, 0, 0o600 << 16,
                _LIMIT32 if offset >= _LIMIT32 else offset))
            out.write(name)
            out.write(extra)
        length = out.tell() - start
        count = len(directory)
        if count >= _LIMIT16 or start >= _LIMIT32 or length >= _LIMIT32:
            end64 = out.tell()
            out.write(struct.pack("<IQHHIIQQQQ", _END64, 44, 45, 45, 0, 0, count, count,
                                  length, start))
            out.write(struct.pack("<IIQI", _LOCATOR64, 0, end64, 1))
        out.write(struct.pack("<IHHHHIIH", _END, 0, 0, min(count, _LIMIT16),
                              min(count, _LIMIT16), min(length, _LIMIT32),
                              min(start, _LIMIT32), 0))
    return path


__all__ = ["write_npz"]
