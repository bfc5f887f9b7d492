"""Row-sharded search across ranks: one process per GPU, `torch.distributed`.

The table shards by rows (rows are independent); the only exchange is the final
top-k merge, exactly where the reference merges its per-storage results on the
host (fingerprintdb_cuda.cu:356-380).  Scheme:

  * rank r owns the contiguous rows [r*ceil(N/G), (r+1)*ceil(N/G)) and returns
    global row indices (``row_base`` = its first row);
  * every rank computes its exact local top-k block {header; k hits} (12 B/hit);
  * ONE all-gather of the blocks (RCCL over xGMI with the ``nccl`` backend; ``gloo``
    on CPU in the tests): k*12+16 bytes per rank -- latency-bound, no bandwidth;
  * every rank merges the G blocks (rank merge on the device, or the host twin).

Exactness: the global top-k is a subset of the union of the local top-k lists and
the canonical order (score desc, GLOBAL row asc) is shard-invariant.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

from . import capi


def shard_range(total_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """(first_row, nrows) of rank's contiguous shard."""
    per = (total_rows + world - 1) // world
    first = min(per * rank, total_rows)
    return first, min(per, total_rows - first)


class ShardedSearch:
    """Gather + merge of per-rank result blocks.

    ``local_search(query, k, block)`` must leave this rank's result block in the
    uint8 tensor ``block`` (device tensor: enqueue on the current stream; CPU
    tensor: fill synchronously).  On a GPU that is ``Table.search_device``.
    """

    def __init__(self, local_search: Callable, k: int, device, group=None, stream_ptr: Optional[int] = None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.k = k
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.blk = capi.result_block_bytes(k)
        self.local_search = local_search
        self.stream_ptr = stream_ptr
        self.local = torch.zeros(self.blk, dtype=torch.uint8, device=self.device)
        self.gathered = torch.zeros(self.blk * self.world, dtype=torch.uint8, device=self.device)
        self.merged = torch.zeros(self.blk, dtype=torch.uint8, device=self.device)
        self.on_gpu = self.device.type == "cuda"
        self.host_out = torch.zeros(self.blk, dtype=torch.uint8)
        if self.on_gpu:
            self.host_out = self.host_out.pin_memory()

    def enqueue(self, query) -> None:
        """Local search, all-gather, merge; on a GPU everything is stream-ordered and
        nothing here waits for the host."""
        self.local_search(query, self.k, self.local)
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.gathered, self.local, group=self.group)
        else:
            self.gathered.copy_(self.local)
        if self.on_gpu:
            capi.merge_device(self.device.index or 0, self.stream_ptr or 0, self.gathered.data_ptr(), self.world,
                              self.blk, self.k, self.merged.data_ptr())
            self.host_out.copy_(self.merged, non_blocking=True)
        else:
            out = capi.merge_host(self.gathered.numpy().tobytes(), self.world, self.blk, self.k)
            self.host_out.copy_(self.torch.frombuffer(bytearray(out), dtype=self.torch.uint8))

    def result(self):
        """(hits, approx, flags) of the last enqueued query (after the stream is synchronised)."""
        return capi.parse_result_block(self.host_out.numpy().tobytes(), self.k)


class ShardedBatchSearch:
    """The same gather + merge for batches of nq queries (BASELINE config 5: 256 queries per call).

    ``local_search(queries, k, blocks)`` leaves this rank's nq result blocks (query-major) in the
    uint8 tensor ``blocks`` -- on a GPU ``Table.search_batch_device``.  ONE all-gather of
    nq * (16 + 12 k) bytes per rank, then one merge launch for all queries.
    """

    def __init__(self, local_search: Callable, k: int, max_queries: int, device, group=None,
                 stream_ptr: Optional[int] = None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.k = k
        self.max_queries = max_queries
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.blk = capi.result_block_bytes(k)
        self.local_search = local_search
        self.stream_ptr = stream_ptr
        n = self.blk * max_queries
        self.local = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self.gathered = torch.zeros(n * self.world, dtype=torch.uint8, device=self.device)
        self.merged = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self.on_gpu = self.device.type == "cuda"
        self.host_out = torch.zeros(n, dtype=torch.uint8)
        if self.on_gpu:
            self.host_out = self.host_out.pin_memory()
        self.nq = 0

    def enqueue(self, queries) -> None:
        nq = len(queries)
        if nq > self.max_queries:
            raise ValueError("batch larger than max_queries")
        self.nq = nq
        n = self.blk * nq
        local = self.local[:n]
        self.local_search(queries, self.k, local)
        gathered = self.gathered[:n * self.world]
        if self.world > 1:
            self.dist.all_gather_into_tensor(gathered, local, group=self.group)
        else:
            gathered.copy_(local)
        if self.on_gpu:
            capi.merge_device_batch(self.device.index or 0, self.stream_ptr or 0, gathered.data_ptr(), self.world, nq,
                                    self.blk, self.k, self.merged.data_ptr())
            self.host_out[:n].copy_(self.merged[:n], non_blocking=True)
        else:
            raw = gathered.numpy().tobytes()
            out = bytearray()
            for q in range(nq):  # the lists of query q: block q of every rank
                lists = b"".join(raw[(r * nq + q) * self.blk:(r * nq + q + 1) * self.blk] for r in range(self.world))
                out += capi.merge_host(lists, self.world, self.blk, self.k)
            self.host_out[:n].copy_(self.torch.frombuffer(out, dtype=self.torch.uint8))

    def results(self):
        """[(hits, approx, flags)] of the last enqueued batch (after the stream is synchronised)."""
        raw = self.host_out.numpy().tobytes()
        return [capi.parse_result_block(raw[q * self.blk:(q + 1) * self.blk], self.k) for q in range(self.nq)]
