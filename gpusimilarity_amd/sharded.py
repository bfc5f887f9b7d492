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

Stream contract.  On a GPU the local search, the gather, the merge and the copy of
the merged block to pinned host memory are all ordered on ONE stream.  Pass the
:class:`capi.Table` itself and the classes here own that contract: they create (or
take) the torch stream, hand it to the table (``gsim_db_set_stream``) and run every
step under it.  A bare callable is accepted too (the CPU tests supply the local
search from the oracle); on a GPU the caller is then responsible for having put the
table on the stream given as ``stream``.
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


class _Gather:
    """What both classes share: process group facts, the stream, the block gather."""

    def _setup(self, local, device, group, stream, search_kwargs):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.on_gpu = self.device.type == "cuda"
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        # gloo cannot move device memory: blocks are staged through pinned host buffers (a test
        # configuration -- two ranks sharing one GPU, which RCCL refuses; production is nccl = RCCL)
        self.staged = self.on_gpu and self.world > 1 and self.backend != "nccl"
        self.stream = None
        self.done = None
        if self.on_gpu:
            self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
            # recorded behind every enqueue's last operation (the copy of the merged block to pinned host memory):
            # synchronize() waits for THIS object's last result only, so a second object on the same stream can have
            # the next query's search, gather and merge enqueued behind it meanwhile
            self.done = torch.cuda.Event()
        self.table = None
        if isinstance(local, capi.Table):
            if not self.on_gpu:
                raise ValueError("a capi.Table searches on a GPU: device must be cuda:N")
            self.table = local
            self.table.set_stream(self.stream.cuda_stream)  # every later enqueue is ordered on self.stream
        self.kw = dict(search_kwargs or {})
        self.local_search = local if self.table is None else None
        # optional per-phase timing (enable_phase_timing): HIP events on self.stream around the local search, the
        # all-gather, the merge and the copy to the host -- read back when the result is waited for
        self.phase_events = None
        self.phase_ms = [0.0, 0.0, 0.0, 0.0]
        self.phase_n = 0
        self._phase_pending = False

    def enable_phase_timing(self, on=True):
        """Accumulate, per enqueue, the stream time of {local search, all-gather, merge, copy to host} (GPU only)."""
        if on and self.on_gpu and self.phase_events is None:
            self.phase_events = [self.torch.cuda.Event(enable_timing=True) for _ in range(5)]
        if not on:
            self.phase_events = None
        self.phase_ms = [0.0, 0.0, 0.0, 0.0]
        self.phase_n = 0
        self._phase_pending = False

    def _mark(self, i):
        if self.phase_events is not None:
            self.phase_events[i].record(self.stream)
            if i == 4:
                self._phase_pending = True

    def phases(self):
        """Mean stream time per enqueue: {"search_ms", "gather_us", "merge_us", "d2h_us", "n"} (None: timing off)."""
        if self.phase_events is None or self.phase_n == 0:
            return None
        n = self.phase_n
        return {"search_ms": self.phase_ms[0] / n, "gather_us": 1e3 * self.phase_ms[1] / n, "merge_us": 1e3 * self.phase_ms[2] / n,
                "d2h_us": 1e3 * self.phase_ms[3] / n, "n": n}

    def _stream_ctx(self):
        if self.on_gpu:
            return self.torch.cuda.stream(self.stream)
        import contextlib
        return contextlib.nullcontext()

    def _all_gather(self, gathered, local, h_local=None, h_gathered=None):
        """gathered = concatenation of every rank's `local`, stream-ordered on a GPU."""
        if self.world == 1:
            gathered.copy_(local)
        elif self.staged:
            h_local.copy_(local, non_blocking=True)
            self.stream.synchronize()
            self.dist.all_gather_into_tensor(h_gathered, h_local, group=self.group)
            gathered.copy_(h_gathered, non_blocking=True)
        else:
            self.dist.all_gather_into_tensor(gathered, local, group=self.group)

    def synchronize(self):
        """Wait until the last enqueued result of this object is in host memory."""
        if self.on_gpu:
            self.done.synchronize()
            if self._phase_pending:
                ev = self.phase_events
                for i in range(4):
                    self.phase_ms[i] += ev[i].elapsed_time(ev[i + 1])
                self.phase_n += 1
                self._phase_pending = False

    def describe(self):
        """Facts about this rank's place in the group (bench.py puts them into its JSON line)."""
        d = {"rank": self.rank, "world": self.world, "backend": self.backend,
             "device": str(self.device), "staged_gather": bool(self.staged)}
        if self.on_gpu:
            p = self.torch.cuda.get_device_properties(self.device)
            d["device_name"] = p.name
            d["pci_bus_id"] = getattr(p, "pci_bus_id", None)
        return d


class ShardedSearch(_Gather):
    """Gather + merge of per-rank result blocks, one query at a time.

    ``local`` is this rank's :class:`capi.Table` (GPU) or a callable
    ``local_search(query, k, block)`` that leaves the rank's result block in the uint8
    tensor ``block`` (device tensor: enqueue on the current stream; CPU tensor: fill
    synchronously).
    """

    def __init__(self, local, k: int, device, group=None, stream=None, search_kwargs: Optional[dict] = None):
        self._setup(local, device, group, stream, search_kwargs)
        torch = self.torch
        self.k = k
        self.blk = capi.result_block_bytes(k)
        self.local = torch.zeros(self.blk, dtype=torch.uint8, device=self.device)
        self.gathered = torch.zeros(self.blk * self.world, dtype=torch.uint8, device=self.device)
        self.merged = torch.zeros(self.blk, dtype=torch.uint8, device=self.device)
        self.host_out = torch.zeros(self.blk, dtype=torch.uint8)
        self.h_local = self.h_gathered = None
        if self.on_gpu:
            self.host_out = self.host_out.pin_memory()
            if self.staged:
                self.h_local = torch.zeros(self.blk, dtype=torch.uint8).pin_memory()
                self.h_gathered = torch.zeros(self.blk * self.world, dtype=torch.uint8).pin_memory()
            torch.cuda.current_stream(self.device).synchronize()  # the zero fills ran on the current stream

    def enqueue(self, query) -> None:
        """Local search, all-gather, merge; on a GPU everything is ordered on self.stream and
        (with the nccl backend) nothing here waits for the host."""
        with self._stream_ctx():
            self._mark(0)
            if self.table is not None:
                self.table.search_device(query, self.k, self.local.data_ptr(), **self.kw)
            else:
                self.local_search(query, self.k, self.local)
            self._mark(1)
            self._all_gather(self.gathered, self.local, self.h_local, self.h_gathered)
            self._mark(2)
            if self.on_gpu:
                capi.merge_device(self.device.index or 0, self.stream.cuda_stream, self.gathered.data_ptr(),
                                  self.world, self.blk, self.k, self.merged.data_ptr())
                self._mark(3)
                self.host_out.copy_(self.merged, non_blocking=True)
                self._mark(4)
                self.done.record(self.stream)
            else:
                out = capi.merge_host(self.gathered.numpy().tobytes(), self.world, self.blk, self.k)
                self.host_out.copy_(self.torch.frombuffer(bytearray(out), dtype=self.torch.uint8))

    def result(self):
        """(hits, approx, flags) of the last enqueued query (after :meth:`synchronize`)."""
        return capi.parse_result_block(self.host_out.numpy().tobytes(), self.k)


class ShardedBatchSearch(_Gather):
    """The same gather + merge for batches of nq queries (BASELINE config 5: 256 queries per call).

    ``local`` is this rank's :class:`capi.Table` or a callable ``local_search(queries, k, blocks)``
    that leaves the rank's nq result blocks (query-major) in the uint8 tensor ``blocks``.  ONE
    all-gather of nq * (16 + 12 k) bytes per rank, then one merge launch for all queries.
    """

    def __init__(self, local, k: int, max_queries: int, device, group=None, stream=None,
                 search_kwargs: Optional[dict] = None):
        self._setup(local, device, group, stream, search_kwargs)
        torch = self.torch
        self.k = k
        self.max_queries = max_queries
        self.blk = capi.result_block_bytes(k)
        n = self.blk * max_queries
        self.local = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self.gathered = torch.zeros(n * self.world, dtype=torch.uint8, device=self.device)
        self.merged = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self.host_out = torch.zeros(n, dtype=torch.uint8)
        self.h_local = self.h_gathered = None
        if self.on_gpu:
            self.host_out = self.host_out.pin_memory()
            if self.staged:
                self.h_local = torch.zeros(n, dtype=torch.uint8).pin_memory()
                self.h_gathered = torch.zeros(n * self.world, dtype=torch.uint8).pin_memory()
            torch.cuda.current_stream(self.device).synchronize()
        self.nq = 0

    def enqueue(self, queries) -> None:
        nq = len(queries)
        if nq > self.max_queries:
            raise ValueError("batch larger than max_queries")
        self.nq = nq
        n = self.blk * nq
        with self._stream_ctx():
            local = self.local[:n]
            self._mark(0)
            if self.table is not None:
                self.table.search_batch_device(queries, self.k, local.data_ptr(), **self.kw)
            else:
                self.local_search(queries, self.k, local)
            self._mark(1)
            gathered = self.gathered[:n * self.world]
            self._all_gather(gathered, local, self.h_local[:n] if self.staged else None,
                             self.h_gathered[:n * self.world] if self.staged else None)
            self._mark(2)
            if self.on_gpu:
                capi.merge_device_batch(self.device.index or 0, self.stream.cuda_stream, gathered.data_ptr(),
                                        self.world, nq, self.blk, self.k, self.merged.data_ptr())
                self._mark(3)
                self.host_out[:n].copy_(self.merged[:n], non_blocking=True)
                self._mark(4)
                self.done.record(self.stream)
            else:
                raw = gathered.numpy().tobytes()
                out = bytearray()
                for q in range(nq):  # the lists of query q: block q of every rank
                    lists = b"".join(raw[(r * nq + q) * self.blk:(r * nq + q + 1) * self.blk]
                                     for r in range(self.world))
                    out += capi.merge_host(lists, self.world, self.blk, self.k)
                self.host_out[:n].copy_(self.torch.frombuffer(out, dtype=self.torch.uint8))

    def results(self):
        """[(hits, approx, flags)] of the last enqueued batch (after :meth:`synchronize`)."""
        raw = self.host_out.numpy().tobytes()
        return [capi.parse_result_block(raw[q * self.blk:(q + 1) * self.blk], self.k) for q in range(self.nq)]
