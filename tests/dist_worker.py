"""Worker of tests/test_sharded_gloo.py: one rank of a world_size-N gloo group on CPU.

Exercises the PRODUCT's multi-rank path (gpusimilarity_amd.sharded: shard ranges,
all-gather of result blocks, gsim_merge_host) with the per-rank local search
supplied by the oracle (there is no GPU here; on a GPU box the local search is
Table.search_device and the merge is gsim_merge_device -- covered by
tests/test_gpu_parity.py::test_device_result_blocks_and_merge).
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from gpusimilarity_amd import capi  # noqa: E402
from gpusimilarity_amd.sharded import ShardedBatchSearch, ShardedSearch, shard_range  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total, W, seed = 30_011, 32, 0xD157
    first, n = shard_range(total, world, rank)
    shard = O.synth_rows(seed, 0, first, n, W)  # this rank's rows only
    state = {"cutoff": 0.0}

    def local_search(query, k, block):
        hits, approx = O.search(query, shard, k, state["cutoff"], row_base=first)
        raw = capi.make_result_block(hits, approx, k)
        block.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))

    ok = True
    whole = O.synth_rows(seed, 0, 0, total, W) if rank == 0 else None
    for k in (1, 10, 1000, 40_000):
        ss = ShardedSearch(local_search, k, "cpu")
        for cutoff in (0.0, 0.07):
            state["cutoff"] = cutoff
            for qi in range(3):
                qrow = O.query_row(qi, total)
                q = O.synth_rows(seed, 0, qrow, 1, W)[0]
                ss.enqueue(q)
                hits, approx, _ = ss.result()
                if rank == 0:
                    want, wap = O.search(q, whole, k, cutoff, nthreads=4)
                    same = (len(hits) == len(want) and approx == wap and (hits["row"] == want["row"]).all()
                            and (hits["score"].view(np.uint32) == want["score"].view(np.uint32)).all()
                            and (hits["common"] == want["common"]).all())
                    ok = ok and bool(same)
    # batches (BASELINE config 5 shape, small): nq result blocks per rank, one all-gather, per-query merge
    def local_batch(queries, k, blocks):
        raw = bytearray()
        for q in queries:
            hits, approx = O.search(q, shard, k, 0.0, row_base=first, metric=1, alpha=np.float32(0.3), beta=np.float32(0.7))
            raw += capi.make_result_block(hits, approx, k)
        blocks.copy_(torch.frombuffer(raw, dtype=torch.uint8))

    sb = ShardedBatchSearch(local_batch, 50, 16, "cpu")
    for nq in (1, 7, 16):
        qs = np.stack([O.synth_rows(seed, 0, O.query_row(100 + i, total), 1, W)[0] for i in range(nq)])
        sb.enqueue(qs)
        res = sb.results()
        ok = ok and len(res) == nq
        if rank == 0:
            for i in range(nq):
                want, wap = O.search(qs[i], whole, 50, 0.0, nthreads=4, metric=1, alpha=np.float32(0.3), beta=np.float32(0.7))
                hits, approx, _ = res[i]
                same = (len(hits) == len(want) and approx == wap and (hits["row"] == want["row"]).all()
                        and (hits["score"].view(np.uint32) == want["score"].view(np.uint32)).all())
                ok = ok and bool(same)
    # every rank ends with the same merged block
    digest = torch.tensor([int(np.frombuffer(ss.host_out.numpy().tobytes(), dtype=np.uint8).sum())])
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    ok = ok and all(int(g) == int(digest) for g in gathered)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
