"""Reader/writer for the reference's ``.fsim`` database files.

Format (all integers big-endian, QDataStream ``Qt_5_2``), as read by
``GPUSimServer::extractData`` (reference gpusim.cpp:173-253) and written by
``python/gpusim_createdb.py:86-98,135-143``::

    i32 version (=3) | cstr dbkey | i32 fp_bitcount | i32 fp_count
    3 x ( i32 nblocks | nblocks x QByteArray(qCompress(payload)) )   # FP, SMILES, ID

``cstr`` = u32 length including the NUL, bytes, NUL.  ``QByteArray`` = u32 length,
bytes.  ``qCompress`` = u32 uncompressed length + a zlib stream.  The FP payload is
the raw concatenation of ``fp_bitcount/8``-byte fingerprints which the server
reinterprets as little-endian int32 words (gpusim.cpp:58-61 ->
fingerprintdb_cuda.cu:117-126); every FP block becomes one storage slice.  The
SMILES / ID payloads are concatenated ``cstr``s.

Pure Python + zlib + numpy; no Qt.
"""
from __future__ import annotations

import struct
import zlib
from dataclasses import dataclass, field
from typing import List

import numpy as np

DATABASE_VERSION = 3  # gpusim.cpp:43


@dataclass
class FsimData:
    version: int
    dbkey: str
    fp_bitcount: int
    fp_count: int
    fp_blocks: List[np.ndarray] = field(default_factory=list)  # each uint32 [rows, W]
    smiles: List[bytes] = field(default_factory=list)
    ids: List[bytes] = field(default_factory=list)

    @property
    def words_per_fp(self) -> int:
        return self.fp_bitcount // 32

    def rows(self) -> np.ndarray:
        """All fingerprints as one uint32 [fp_count, W] array (slices concatenated)."""
        if not self.fp_blocks:
            return np.zeros((0, self.words_per_fp), dtype=np.uint32)
        return np.ascontiguousarray(np.concatenate(self.fp_blocks, axis=0))


class _Cursor:
    def __init__(self, buf: bytes):
        self.buf = buf
        self.off = 0

    def i32(self) -> int:
        (v,) = struct.unpack_from(">i", self.buf, self.off)
        self.off += 4
        return v

    def u32(self) -> int:
        (v,) = struct.unpack_from(">I", self.buf, self.off)
        self.off += 4
        return v

    def take(self, n: int) -> bytes:
        if self.off + n > len(self.buf):
            raise ValueError("truncated .fsim stream")
        b = self.buf[self.off:self.off + n]
        self.off += n
        return b

    def cstr(self) -> bytes:
        n = self.u32()
        raw = self.take(n)
        return raw[:-1] if n else b""

    def qbytearray(self) -> bytes:
        n = self.u32()
        if n == 0xFFFFFFFF:
            return b""
        return self.take(n)

    def at_end(self) -> bool:
        return self.off >= len(self.buf)


def q_uncompress(blob: bytes) -> bytes:
    """Qt ``qUncompress``: u32 big-endian expected length + zlib stream."""
    if len(blob) < 4:
        return b""
    (expect,) = struct.unpack_from(">I", blob, 0)
    out = zlib.decompress(blob[4:])
    if len(out) != expect:
        raise ValueError("qUncompress length mismatch")
    return out


def q_compress(payload: bytes, level: int = -1) -> bytes:
    return struct.pack(">I", len(payload)) + zlib.compress(payload, level)


def _cstr_list(payload: bytes) -> List[bytes]:
    cur = _Cursor(payload)
    out = []
    while not cur.at_end():
        out.append(cur.cstr())
    return out


def read_fsim(path: str) -> FsimData:
    with open(path, "rb") as f:
        cur = _Cursor(f.read())
    version = cur.i32()
    if version != DATABASE_VERSION:
        # gpusim.cpp:186-189
        raise RuntimeError("Database version incompatible with this GPUSim version")
    dbkey = cur.cstr().decode("utf-8")
    fp_bitcount = cur.i32()
    fp_count = cur.i32()
    data = FsimData(version, dbkey, fp_bitcount, fp_count)
    W = fp_bitcount // 32
    for _ in range(cur.i32()):
        raw = q_uncompress(cur.qbytearray())
        data.fp_blocks.append(np.frombuffer(raw, dtype="<u4").reshape(-1, W).copy())
    for _ in range(cur.i32()):
        data.smiles.extend(_cstr_list(q_uncompress(cur.qbytearray())))
    for _ in range(cur.i32()):
        data.ids.extend(_cstr_list(q_uncompress(cur.qbytearray())))
    total = sum(b.shape[0] for b in data.fp_blocks)
    if total != fp_count:
        # fingerprintdb_cuda.cu:153-156
        raise RuntimeError("Mismatch between FP count and data, potential database corruption.")
    return data


def _cstr(b: bytes) -> bytes:
    return struct.pack(">I", len(b) + 1) + b + b"\0"


def write_fsim(path: str, dbkey: str, fp_bitcount: int, fp_blocks, smiles, ids, smiles_blocks: int = 1,
               id_blocks: int = 1) -> None:
    """Writer for tests and benchmark fixtures: one FP block per entry of ``fp_blocks`` (each becomes a
    storage in the server, like the ~1 GiB blocks gpusim_createdb.py:56-69 rolls over), the SMILES / ID
    strings split into ``smiles_blocks`` / ``id_blocks`` blocks of about equal length."""
    fp_blocks = [np.ascontiguousarray(b, dtype="<u4") for b in fp_blocks]
    fp_count = sum(b.shape[0] for b in fp_blocks)
    out = [struct.pack(">i", DATABASE_VERSION), _cstr(dbkey.encode("utf-8")),
           struct.pack(">i", fp_bitcount), struct.pack(">i", fp_count)]

    def qba_list(payloads):
        parts = [struct.pack(">i", len(payloads))]
        for p in payloads:
            blob = q_compress(p)
            parts.append(struct.pack(">I", len(blob)) + blob)
        return b"".join(parts)

    def split(strings, nblocks):
        strings = list(strings)
        per = (len(strings) + nblocks - 1) // max(1, nblocks)
        return [b"".join(_cstr(s) for s in strings[i * per:(i + 1) * per]) for i in range(max(1, nblocks))]

    out.append(qba_list([b.tobytes() for b in fp_blocks]))
    out.append(qba_list(split(smiles, smiles_blocks)))
    out.append(qba_list(split(ids, id_blocks)))
    with open(path, "wb") as f:
        f.write(b"".join(out))
