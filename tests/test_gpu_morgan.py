"""Morgan-shaped tables (GSIM_SYNTH_MORGAN: popcounts 20..53, frequent bits, series of analogs, exact
duplicates -- the shape of the fingerprints the reference's numbers are quoted on,
python/gpusim_utils.py:21,55-66): coarse scores, a k-th score shared by tens to hundreds of rows.
The table is generated ON THE DEVICE (generate_morgan_kernel) and searched; the oracle generates the
same rows with its own restatement and scans them on the CPU."""
import numpy as np
import pytest

import oracle_lib as O
from gpusimilarity_amd import capi

pytestmark = pytest.mark.gpu
SEED = 0x5EED0001


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def same(got, approx, want, wap, ctx):
    assert int(approx) == wap, (ctx, int(approx), wap)
    assert len(got) == len(want), (ctx, len(got), len(want))
    assert (got["row"] == want["row"]).all(), ctx
    assert (bits(got["score"]) == bits(want["score"])).all(), ctx
    assert (got["common"] == want["common"]).all() and (got["popc_db"] == want["popc_db"]).all(), ctx


@pytest.mark.parametrize("n,W", [(1_200_000, 32), (20_000_000, 32), (3_000_001, 64), (700_001, 16)])
def test_device_generated_morgan_table_matches_the_oracle(n, W):
    db = O.synth_rows(SEED, O.KIND_MORGAN if hasattr(O, "KIND_MORGAN") else 2, 0, n, W)
    t = capi.Table(W * 32).generate(SEED, capi.SYNTH_MORGAN, 0, n, 0)
    for r in (0, 1, n // 2, n - 1):  # the device generator against the oracle's, row by row
        assert (t.row(r) == db[r]).all()
    nq = 24 if n <= 3_000_001 else 8
    qs = np.stack([db[O.query_row(i, n)] for i in range(nq)])
    t.enable_timing(True)
    for i in range(nq):
        for k, cutoff in (((1000, 0.0), (10, 0.0), (1000, 0.3)) if i % 4 else ((1000, 0.0), (4096, 0.0), (100, 0.6), (5000, 0.0))):
            hits, approx = t.search(qs[i], k, cutoff)
            want, wap = O.search(qs[i], db, k, cutoff, nthreads=32)
            same(hits[0], approx[0], want, wap, ("single", n, W, i, k, cutoff))
    tm = t.timing()
    print("morgan %d x %d-bit: %d queries, handed back %d, published/query %.0f" % (
        n, W * 32, tm["queries"], tm["handed_back"], tm["finalists_sum"] / max(1, tm["queries"])))
    t.enable_timing(False)
    # shared passes: Tanimoto and Tversky, with and without a cutoff
    for kw in (dict(), dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))):
        okw = dict(metric=kw["metric"], alpha=kw["alpha"], beta=kw["beta"]) if kw else {}
        for cutoff in (0.0, 0.45):
            bh, bap = t.search(qs, 100, cutoff, **kw)
            for i in range(0, nq, 3):
                want, wap = O.search(qs[i], db, 100, cutoff, nthreads=32, **okw)
                same(bh[i], bap[i], want, wap, ("batch", n, W, i, cutoff, bool(kw)))
    t.close()


def test_exact_duplicates_come_back_in_row_order():
    """Duplicates of the query row all score 1.0; the canonical order returns them by ascending row."""
    n, W = 2_000_000, 32
    db = O.synth_rows(SEED, 2, 0, n, W)
    # find a few rows that have a twin
    _, inv, cnt = np.unique(db[:400_000], axis=0, return_inverse=True, return_counts=True)
    twins = np.flatnonzero(cnt[inv] > 1)[:6]
    assert len(twins) >= 3
    t = capi.Table(W * 32).generate(SEED, capi.SYNTH_MORGAN, 0, n, 0)
    for r in twins:
        hits, approx = t.search(db[r], 50, 0.0)
        want, wap = O.search(db[r], db, 50, 0.0, nthreads=32)
        same(hits[0], approx[0], want, wap, ("twin", int(r)))
        ones = hits[0][hits[0]["score"] == np.float32(1.0)]
        assert len(ones) >= 2 and int(r) in ones["row"] and (np.diff(ones["row"].astype(np.int64)) > 0).all()
    t.close()
