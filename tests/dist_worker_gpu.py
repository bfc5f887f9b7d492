"""Worker of tests/test_gpu_sharded.py: one rank of a world_size-N group whose ranks all
search on cuda:0 (the box has one GPU; RCCL refuses two ranks per device, so the group is
gloo and the result blocks are staged through pinned host memory -- everything else is the
production path: Table.search_device / search_batch_device on this rank's shard, ONE
all-gather of the blocks, gsim_merge_device[_batch] on the GPU, merged block to pinned
host memory).  Rank 0 compares with the oracle on the WHOLE table.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402  (the checker)
from gpusimilarity_amd import capi  # noqa: E402
from gpusimilarity_amd.sharded import ShardedBatchSearch, ShardedSearch, shard_range  # noqa: E402


def same(hits, approx, want, wap):
    return bool(len(hits) == len(want) and approx == wap and (hits["row"] == want["row"]).all()
                and (hits["score"].view(np.uint32) == want["score"].view(np.uint32)).all()
                and (hits["common"] == want["common"]).all() and (hits["popc_db"] == want["popc_db"]).all())


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ok = True
    for total, W, seed in ((400_003, 32, 0xD158), (1_300_000, 32, 0xD159)):
        first, n = shard_range(total, world, rank)
        shard = O.synth_rows(seed, 0, first, n, W)  # this rank's rows only
        table = capi.Table(W * 32).add_rows(shard).finalize(0, 1)
        table.set_row_base(first)
        whole = O.synth_rows(seed, 0, 0, total, W) if rank == 0 else None
        for k in (10, 1000):
            ss = ShardedSearch(table, k, "cuda:0")
            assert ss.staged and ss.world == world and ss.backend == "gloo"
            for cutoff in (0.0, 0.07):
                ss.kw = dict(cutoff=cutoff)
                for qi in range(3):
                    q = O.synth_rows(seed, 0, O.query_row(qi, total), 1, W)[0]
                    ss.enqueue(q)
                    ss.synchronize()
                    hits, approx, _ = ss.result()
                    if rank == 0:
                        want, wap = O.search(q, whole, k, cutoff, nthreads=8)
                        ok = ok and same(hits, approx, want, wap)
        # a 64-query Tversky batch (the matrix-core pass on every rank), one gather, one merge launch
        kw = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
        sb = ShardedBatchSearch(table, 100, 64, "cuda:0", search_kwargs=kw)
        qs = np.stack([O.synth_rows(seed, 0, O.query_row(200 + i, total), 1, W)[0] for i in range(64)])
        sb.enqueue(qs)
        sb.synchronize()
        res = sb.results()
        ok = ok and len(res) == 64
        if rank == 0:
            for i in range(0, 64, 7):
                want, wap = O.search(qs[i], whole, 100, 0.0, nthreads=8, **kw)
                ok = ok and same(res[i][0], res[i][1], want, wap)
        # every rank ends with the same merged block
        digest = torch.tensor([int(np.frombuffer(ss.host_out.numpy().tobytes(), dtype=np.uint8).astype(np.int64).sum())])
        gathered = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(gathered, digest)
        ok = ok and all(int(g) == int(digest) for g in gathered)
        table.close()
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
