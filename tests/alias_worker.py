"""Worker of tests/test_gpu_alias_devices.py.  Started with GSIM_TEST_ALIAS_DEVICES=4: the library
presents four logical devices that all live on GPU 0, so the IN-PROCESS multi-device code of the
reference's FingerprintDB::search (fan-out over the storages' devices fingerprintdb_cuda.cu:356-362,
round-robin placement :54-68, :176-182, host merge :363-380) runs on a one-GPU box: every shard has
its own stream, per-query state and scratch, exactly as on four GPUs.  Everything is checked against
the oracle on the whole table."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402  (the checker)
from gpusimilarity_amd import capi  # noqa: E402
from gpusimilarity_amd.fingerprintdb import FingerprintDB, get_gpu_count, get_next_gpu  # noqa: E402


def same(hits, approx, want, wap, ctx):
    assert int(approx) == wap, (ctx, int(approx), wap)
    assert len(hits) == len(want), (ctx, len(hits), len(want))
    assert (hits["row"] == want["row"]).all(), ctx
    assert (hits["score"].view(np.uint32) == want["score"].view(np.uint32)).all(), ctx
    assert (hits["common"] == want["common"]).all() and (hits["popc_db"] == want["popc_db"]).all(), ctx


def main():
    ndev = int(os.environ["GSIM_TEST_ALIAS_DEVICES"])
    assert capi.device_count() == ndev == get_gpu_count()
    # get_next_gpu: round robin over the logical devices (fingerprintdb_cuda.cu:54-68)
    seen = [get_next_gpu(1 << 20) for _ in range(2 * ndev)]
    assert sorted(set(seen)) == list(range(ndev)) and seen[:ndev] == seen[ndev:], seen
    for kind, n, W in ((0, 300_001, 32), (2, 1_200_003, 32), (0, 150_000, 64), (2, 70_001, 28)):
        db = O.synth_rows(0x5AAD + n, kind, 0, n, W)
        # both merge routes: on the host (the reference's, the default) and through the C-ABI collective (gsim_comm:
        # every shard's block stays in HBM, all-gather, merge_kernel -- on aliased devices the gather is a loop-back of
        # device-to-device copies, RCCL refuses two ranks on one GPU)
        for ndevices, route in ((0, "host"), (3, "host"), (0, "comm"), (3, "comm")):
            t = capi.Table(W * 32).add_rows(db).finalize(0, ndevices)
            assert t.shard_count() == (ndev if ndevices == 0 else ndevices)
            if route == "comm":
                comm = capi.Comm(list(range(t.shard_count())))
                assert comm.size() == t.shard_count()
                t.set_comm(comm)
                t.enable_timing(True)
            qs = np.stack([db[O.query_row(i, n)] for i in range(70)])
            for qi, (k, cutoff) in enumerate(((1000, 0.0), (10, 0.0), (1000, 0.2), (4096, 0.0), (9000, 0.0), (100, 0.55))):
                hits, approx = t.search(qs[qi], k, cutoff)
                want, wap = O.search(qs[qi], db, k, cutoff, nthreads=8)
                same(hits[0], approx[0], want, wap, ("single", kind, n, W, ndevices, k, cutoff))
            # queries one after the other (gsim_db_search_each) and a shared pass (70 queries: the matrix cores on
            # every shard, host merge per query)
            for each in (True, False):
                bufs = t.make_search_buffers(len(qs), 100)
                kw = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
                (t.search_each_into if each else t.search_into)(np.ascontiguousarray(qs), 100, bufs, 0.0, **kw)
                for i in range(0, len(qs), 9):
                    want, wap = O.search(qs[i], db, 100, 0.0, nthreads=8, **kw)
                    same(bufs[0][i, :bufs[1][i]], bufs[2][i], want, wap, ("each" if each else "batch", kind, n, W, ndevices, i))
            if route == "comm":
                tm = t.timing()
                assert tm["collectives"] in (6 + 70 + 1, 6 + 70 + 70) and tm["gather_ms_sum"] > 0 and tm["merge_ms_sum"] > 0, tm
                # the gather fills EVERY shard's buffer (like the real all-gather): merge on each of the other devices in
                # turn -- single query and batch -- and expect the same answer
                for root in range(1, t.shard_count()):
                    t.set_comm_root(root)
                    hits, approx = t.search(qs[1], 1000, 0.0)
                    want, wap = O.search(qs[1], db, 1000, 0.0, nthreads=8)
                    same(hits[0], approx[0], want, wap, ("comm root", root, kind, n, W, ndevices))
                    bufs = t.make_search_buffers(len(qs), 50)
                    t.search_into(np.ascontiguousarray(qs), 50, bufs, 0.0)
                    for i in (0, 33, 69):
                        want, wap = O.search(qs[i], db, 50, 0.0, nthreads=8)
                        same(bufs[0][i, :bufs[1][i]], bufs[2][i], want, wap, ("comm root batch", root, kind, n, W, ndevices, i))
                t.set_comm_root(0)
                t.set_comm(None)  # back to the host merge: same answer
                hits, approx = t.search(qs[0], 1000, 0.0)
                want, wap = O.search(qs[0], db, 1000, 0.0, nthreads=8)
                same(hits[0], approx[0], want, wap, ("host after comm", kind, n, W, ndevices))
                t.close()
                comm.close()
                continue
            t.close()
    # a synthetic table generated shard by shard on its devices == the same table generated on one device
    for kind, n, W in ((0, 1_000_003, 32), (2, 500_001, 64)):
        one = capi.Table(W * 32).generate(0x5EED0003, kind, 7, n, 0)
        many = capi.Table(W * 32).generate(0x5EED0003, kind, 7, n, 0, ndevices=ndev)
        assert many.shard_count() == ndev and one.shard_count() == 1
        qs = np.stack([O.synth_rows(0x5EED0003, kind, 7 + O.query_row(i, n), 1, W)[0] for i in range(20)])
        for route in ("each", "batch"):
            b1, b2 = one.make_search_buffers(len(qs), 200), many.make_search_buffers(len(qs), 200)
            (one.search_each_into if route == "each" else one.search_into)(np.ascontiguousarray(qs), 200, b1, 0.0)
            (many.search_each_into if route == "each" else many.search_into)(np.ascontiguousarray(qs), 200, b2, 0.0)
            assert (b1[1] == b2[1]).all() and (b1[2] == b2[2]).all(), route
            assert b1[0].tobytes() == b2[0].tobytes(), route
        one.close()
        many.close()
    # a folded table of three storages: one shard per add_rows slice, placed round robin (:184-194)
    n, W = 90_000, 32
    db = O.synth_rows(0xF01D, 2, 0, n, W)
    t = capi.Table(1024)
    for lo, hi in ((0, 30_000), (30_000, 70_001), (70_001, n)):
        t.add_rows(db[lo:hi])
    t.set_fold_factor(4).finalize(-1, 1)
    assert t.shard_count() == 3
    try:  # folded tables merge on the host: their re-score needs every storage's candidates
        t.set_comm(capi.Comm([0, 1, 2]))
        raise AssertionError("a folded table accepted a communicator")
    except capi.GsimError:
        pass
    for qi in range(4):
        q = db[O.query_row(qi, n)]
        hits, approx = t.search(q, 50, 0.0)
        parts, ap = [], 0
        for lo, hi in ((0, 30_000), (30_000, 70_001), (70_001, n)):
            h, a = O.search_folded(q, db[lo:hi], 4, 50, 0.0, row_base=lo)
            parts.append(h)
            ap += a
        want = O.merge_hits(parts, 50)
        assert int(approx[0]) == ap
        assert (hits[0]["row"] == want["row"]).all() and (hits[0]["score"].view(np.uint32) == want["score"].view(np.uint32)).all()
    t.close()
    # the Python twin of the reference class, sharded over all logical devices
    blocks = [db[:40_000].tobytes(), db[40_000:].tobytes()]
    smiles = [b"S%d" % i for i in range(n)]
    ids = [b"I%d" % i for i in range(n)]
    fdb = FingerprintDB(1024, n, "key", blocks, smiles, ids)
    fdb.copyToGPU(1, ndevices=0)
    q = db[O.query_row(9, n)]
    sm, idl, sc, ap = fdb.search(q, "key", 20, 0.0)
    want, wap = O.search(q, db, 20, 0.0, nthreads=8)
    assert ap == wap and [int(x[1:]) for x in idl] == list(want["row"])
    print("alias worker ok")


if __name__ == "__main__":
    main()
