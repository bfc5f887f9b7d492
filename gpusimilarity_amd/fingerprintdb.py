"""Python twin of the reference's ``gpusim::FingerprintDB`` (fingerprintdb_cuda.h:53-140)
over the C ABI: same method names, argument meaning and error behaviour, so the
parity tests read like test/test_gpusim.cpp.  The C++ twin used by the server is
gpusimilarity_amd/csrc/host/fingerprintdb.h.

Strings (SMILES / IDs) live here, above the ABI; the ABI speaks row indices.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from . import capi


def get_gpu_count() -> int:
    """fingerprintdb_cuda.cu:40-52"""
    return capi.device_count()


def get_next_gpu(required_memory: int) -> int:
    """fingerprintdb_cuda.cu:54-68 (throws when no GPU has room)"""
    return capi.next_device(required_memory)


def get_available_gpu_memory() -> int:
    """fingerprintdb_cuda.cu:401-413"""
    return capi.available_device_bytes()


class FingerprintDB:
    def __init__(self, fp_bitcount: int, fp_count: int, dbkey: str, data: Sequence[np.ndarray],
                 smiles_vector: List[bytes], ids_vector: List[bytes]):
        """fingerprintdb_cuda.cu:133-166.  ``data``: one uint32 [rows, W] array (or raw
        bytes) per storage block."""
        if fp_bitcount % 32 != 0:
            raise ValueError("fingerprint size must be a multiple of 32 bits")
        self.m_fp_intsize = fp_bitcount // 32
        self.m_total_count = fp_count
        self.m_dbkey = dbkey
        self.m_fold_factor = 1
        self._table = capi.Table(fp_bitcount)
        current = 0
        for block in data:
            if isinstance(block, (bytes, bytearray, memoryview)):
                block = np.frombuffer(block, dtype="<u4")
            block = np.ascontiguousarray(block, dtype=np.uint32).reshape(-1, self.m_fp_intsize)
            self._table.add_rows(block)
            current += block.shape[0]
        if current != fp_count:
            # fingerprintdb_cuda.cu:153-156
            raise RuntimeError("Mismatch between FP count and data, potential database corruption.")
        self.m_total_data_size = fp_count * self.m_fp_intsize * 4
        # the reference steals the callers' vectors (:164-165)
        self.m_smiles = list(smiles_vector)
        self.m_ids = list(ids_vector)
        smiles_vector.clear()
        ids_vector.clear()
        self._on_gpu = False

    def copyToGPU(self, fold_factor: int = 1, device: int = -1, ndevices: int = 1):
        """fingerprintdb_cuda.cu:168-195.  fold_factor > 1 keeps an OR-folded copy on the
        GPU and makes search() the reference's approximate folded search (re-scored with
        the full fingerprints); the effective factor is the smallest one >= fold_factor
        that divides the word count (:170-173)."""
        if fold_factor > 1:
            self._table.set_fold_factor(fold_factor)
        self._table.finalize(device, ndevices)
        self.m_fold_factor = self._table.fold_factor()
        self._on_gpu = True

    def count(self) -> int:
        return self.m_total_count

    def getFingerprint(self, index: int) -> np.ndarray:
        """fingerprintdb_cuda.cu:212-226"""
        return self._table.row(index)

    def getSmiles(self, index: int) -> bytes:
        return self.m_smiles[index]

    def getID(self, index: int) -> bytes:
        return self.m_ids[index]

    def getFingerprintDataSize(self) -> int:
        return self.m_total_data_size

    def getFingerprintBitcount(self) -> int:
        return self.m_fp_intsize * 32

    def search(self, query, dbkey: str, max_return_count: int, similarity_cutoff: float
               ) -> Tuple[List[bytes], List[bytes], List[float], int]:
        """fingerprintdb_cuda.cu:341-381 -> (smiles, ids, scores, approximate_result_count).
        Wrong dbkey: empty result (:349-352)."""
        if dbkey != self.m_dbkey:
            return [], [], [], 0
        hits, approx = self._table.search(np.asarray(query, dtype=np.uint32), max_return_count,
                                          float(similarity_cutoff))
        h = hits[0]
        return ([self.m_smiles[r] for r in h["row"]], [self.m_ids[r] for r in h["row"]],
                [float(s) for s in h["score"]], int(approx[0]))

    def search_hits(self, query, max_return_count: int, similarity_cutoff: float = 0.0, **kw):
        """Row-level result (row, score, common, popc_db) -- what the ABI returns."""
        hits, approx = self._table.search(np.asarray(query, dtype=np.uint32), max_return_count,
                                          float(similarity_cutoff), **kw)
        return hits[0], int(approx[0])

    def search_cpu(self, query, dbkey: str, max_return_count: int, similarity_cutoff: float
                   ) -> Tuple[List[bytes], List[bytes], List[float]]:
        """fingerprintdb_cuda.cpp:20-54 (cutoff ignored, approx not produced)."""
        if dbkey != self.m_dbkey:
            return [], [], []
        h = self._table.search_cpu(np.asarray(query, dtype=np.uint32), max_return_count, float(similarity_cutoff))[0]
        return ([self.m_smiles[r] for r in h["row"]], [self.m_ids[r] for r in h["row"]],
                [float(s) for s in h["score"]])


def top_results_bubble_sort(indices: List[int], scores: List[float], number_required: int) -> None:
    """fingerprintdb_cuda.cpp:92-103, in place."""
    count = len(indices)
    for i in range(number_required):
        for j in range(count - 1, i, -1):
            if scores[j] > scores[j - 1]:
                indices[j], indices[j - 1] = indices[j - 1], indices[j]
                scores[j], scores[j - 1] = scores[j - 1], scores[j]
