#!/usr/bin/env python3
"""What the two ways of merging per-GPU top-k blocks cost per query:
  host : the in-process multi-device path (gpusimserver --gpus N, gsim_db_finalize(db, dev, n)) -- every shard's kernel
         writes its block straight into pinned host memory, the host does a k-way merge (gsim_merge_host's code);
  rccl : the one-process-per-GPU path (sharded.py) -- block in device memory, all_gather_into_tensor, merge kernel, D2H.
Measured on ONE GPU: (a) the host merge of 8 blocks of k hits, (b) the per-query overhead of the device-block route at world
size 1 (collective degenerates to a copy: what is left is the merge launch, the D2H copy and the stream synchronisation).
    python scripts/merge_cost.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from gpusimilarity_amd import capi  # noqa: E402


def main():
    k, G = 1000, 8
    rng = np.random.default_rng(1)
    blocks = b""
    for g in range(G):
        h = np.zeros(k, dtype=capi.HIT_DTYPE)
        h["score"] = np.sort(rng.random(k).astype(np.float32))[::-1]
        h["row"] = np.arange(k, dtype=np.uint32) * G + g
        blocks += capi.make_result_block(h, 125_000_000, k)
    blk = capi.result_block_bytes(k)
    capi.merge_host(blocks, G, blk, k)
    t0 = time.perf_counter()
    reps = 2000
    for _ in range(reps):
        capi.merge_host(blocks, G, blk, k)
    host_us = 1e6 * (time.perf_counter() - t0) / reps
    out = {"host_merge_us_8x1000_hits_incl_ctypes_copy": host_us}
    import torch
    from bench import DB_SEED, query_row, synth_row
    from gpusimilarity_amd.sharded import ShardedSearch
    n = 1_000_000
    t = capi.Table(1024).generate(DB_SEED, capi.SYNTH_SPARSE, 0, n, 0)
    qs = [synth_row(DB_SEED, capi.SYNTH_SPARSE, query_row(i, n), 32) for i in range(16)]
    bufs = t.make_search_buffers(1, k)
    for i in range(200):
        t.search_into(qs[i % 16], k, bufs)
    t0 = time.perf_counter()
    for i in range(1000):
        t.search_into(qs[i % 16], k, bufs)
    direct = 1e6 * (time.perf_counter() - t0) / 1000
    ss = ShardedSearch(t, k, torch.device("cuda", 0))
    for i in range(200):
        ss.enqueue(qs[i % 16])
        ss.synchronize()
    t0 = time.perf_counter()
    for i in range(1000):
        ss.enqueue(qs[i % 16])
        ss.synchronize()
    shard = 1e6 * (time.perf_counter() - t0) / 1000
    out.update({"single_query_us_1M_rows_direct": direct, "single_query_us_1M_rows_device_block_route_world1": shard,
                "device_block_route_overhead_us": shard - direct})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
