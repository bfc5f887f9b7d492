"""Seeded random sweep of the hot path through the C ABI against the oracle: table sizes, widths, row kinds, k, cutoffs
and metrics drawn at random (the parametrised suites sit on the geometry switches; this one walks between them).
Single queries (the single-launch kernel and whatever it hands back) and small batches (the multi-query passes).
Reference: FingerprintDB::search, fingerprintdb_cuda.cu:228-380."""
import numpy as np
import pytest

import oracle_lib as O
from gpusimilarity_amd import capi
from test_gpu_parity import assert_hits_equal, make_table

pytestmark = pytest.mark.gpu


def draw(rng):
    W = int(rng.choice([4, 8, 16, 32, 32, 32, 64, 128]))
    n = int(np.exp(rng.uniform(np.log(200), np.log(2_500_000 if W <= 32 else 500_000))))
    kind = int(rng.choice([0, 0, 1, O.KIND_MORGAN])) if W == 32 else int(rng.choice([0, 1]))
    return W, n, kind


@pytest.mark.parametrize("seed", range(32))
def test_random_tables_and_queries_match_the_oracle(seed):
    rng = np.random.default_rng(0xF022 + seed)
    W, n, kind = draw(rng)
    db = O.synth_rows(0xF0220000 + seed, kind, 0, n, W)
    t = make_table(db)
    tv = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(rng.choice([0.3, 0.5, 1.0, 0.0])), beta=np.float32(rng.choice([0.7, 0.5, 1.0])))
    for case in range(14):
        own = rng.random() < 0.75
        q = db[int(rng.integers(n))] if own else O.synth_rows(0xF0229999 + seed, 0 if kind != 1 else 1, 50 + case, 1, W)[0]
        if case == 13 and seed % 4 == 0:
            q = np.zeros(W, dtype=np.uint32)  # (0 / 0 -> NaN -> 0: every row ties)
        k = int(rng.choice([1, 3, 20, 100, 1000, 1500, 2048, 3000, 5000, 8192, 9000, 20000]))
        cutoff = float(rng.choice([0.0, 0.0, 0.0, 0.05, 0.2, 0.5, 0.9]))
        kw = tv if rng.random() < 0.3 else {}
        hits, approx = t.search(q, k, np.float32(cutoff), **kw)
        want, wap = O.search(q, db, k, np.float32(cutoff), nthreads=8, **kw)
        ctx = "seed %d case %d: n=%d W=%d kind=%d k=%d cutoff=%g %s" % (seed, case, n, W, kind, k, cutoff, "tversky" if kw else "tanimoto")
        assert int(approx[0]) == wap, ctx
        assert_hits_equal(hits[0], want, ctx)
    # a batch of queries through the multi-query passes
    nq = int(rng.choice([4, 9, 33, 70]))
    qs = db[rng.integers(n, size=nq)]
    k = int(rng.choice([1, 50, 1000]))
    cutoff = float(rng.choice([0.0, 0.1]))
    kw = tv if rng.random() < 0.5 else {}
    hits, approx = t.search(qs, k, np.float32(cutoff), **kw)
    for i in range(nq):
        want, wap = O.search(qs[i], db, k, np.float32(cutoff), nthreads=8, **kw)
        ctx = "seed %d batch q %d: n=%d W=%d kind=%d k=%d cutoff=%g" % (seed, i, n, W, kind, k, cutoff)
        assert int(approx[i]) == wap, ctx
        assert_hits_equal(hits[i], want, ctx)
    t.close()


ODD_WIDTHS = [3, 5, 6, 7, 10, 14, 12, 20, 24, 28, 36, 44, 52, 60, 72, 9, 11, 18, 22, 13, 17, 33]


@pytest.mark.parametrize("seed", range(22))
def test_random_odd_width_tables_match_the_oracle(seed):
    """The same walk over the widths off the power-of-two template: rows of 3 ... 14 words (word-granular streaming inside the
    single launch), of 3 ... 15 (x 2^i) sixteen-byte units (register-streamed), and a few that keep the LDS-staged scan; sizes
    on both sides of the seeding rule (1500 rows per wave), k on both sides of 8192 (the four-kernel pipeline and its
    two-launch sort above it)."""
    rng = np.random.default_rng(0x0DD0 + seed)
    W = ODD_WIDTHS[seed % len(ODD_WIDTHS)] if seed < len(ODD_WIDTHS) else int(rng.choice(ODD_WIDTHS))
    n = int(np.exp(rng.uniform(np.log(200), np.log(3_000_000 if W <= 16 else 600_000))))
    kind = int(rng.choice([0, 1]))
    db = O.synth_rows(0x0DD00000 + seed, kind, 0, n, W)
    t = make_table(db)
    tv = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(rng.choice([0.3, 0.5, 1.0, 0.0])), beta=np.float32(rng.choice([0.7, 0.5, 1.0])))
    for case in range(12):
        own = rng.random() < 0.75
        q = db[int(rng.integers(n))] if own else O.synth_rows(0x0DD09999 + seed, kind, 50 + case, 1, W)[0]
        if case == 11 and seed % 4 == 0:
            q = np.zeros(W, dtype=np.uint32)
        k = int(rng.choice([1, 3, 20, 100, 1000, 1500, 2048, 3000, 5000, 8192, 9000, 20000, 40000]))
        cutoff = float(rng.choice([0.0, 0.0, 0.0, 0.05, 0.2, 0.5, 0.9]))
        kw = tv if rng.random() < 0.3 else {}
        hits, approx = t.search(q, k, np.float32(cutoff), **kw)
        want, wap = O.search(q, db, k, np.float32(cutoff), nthreads=8, **kw)
        ctx = "seed %d case %d: n=%d W=%d kind=%d k=%d cutoff=%g %s" % (seed, case, n, W, kind, k, cutoff, "tversky" if kw else "tanimoto")
        assert int(approx[0]) == wap, ctx
        assert_hits_equal(hits[0], want, ctx)
    t.close()


def large_k_walk(seed, stats=None):
    """One random table of 150 k (every third seed: 20 k) ... 3 M rows of 512 ... 4096 bits (i.i.d., dense, Morgan-shaped; every fifth with heavy
    duplication), eight queries with k from 2049 to 200 000 (cutoffs, Tversky) through gsim_db_search and the same queries through
    the pipelined gsim_db_search_each, each against the oracle: the single launch publishes and the rows are placed by score
    bin and ranked inside their bins -- or (ties, k above 32768, hand-backs) the radix tail / the four-kernel pipeline answers."""
    rng = np.random.default_rng(0x1A26E + seed)
    W = int(rng.choice([16, 32, 32, 32, 64, 128])) if seed % 4 else int(rng.choice([4, 8]))  # (every fourth: 128 / 256-bit rows)
    n = int(np.exp(rng.uniform(np.log(20_000 if seed % 3 == 1 else 150_000), np.log(3_000_000 if W <= 32 else 700_000))))  # (every third: short tables too)
    kind = int(rng.choice([0, 0, 1, O.KIND_MORGAN])) if W == 32 else int(rng.choice([0, 1]))
    db = O.synth_rows(0x1A260000 + seed, kind, 0, n, W)
    if seed % 5 == 4:  # heavy duplication: a few distinct rows
        db = np.ascontiguousarray(db[rng.integers(0, int(rng.choice([8, 300, 5000])), size=n)])
    t = make_table(db)
    tv = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(rng.choice([0.3, 0.5, 1.0])), beta=np.float32(rng.choice([0.7, 0.5, 1.0])))
    cases = []
    for case in range(8):
        q = db[int(rng.integers(n))] if rng.random() < 0.8 else O.synth_rows(0x1A269999 + seed, 0 if kind != 1 else 1, 50 + case, 1, W)[0]
        k = int(rng.choice([2049, 2500, 3000, 4096, 4097, 6000, 8192, 8193, 12000, 20000, 32768, 32769, 40000, 50000, 65536, 65537, 100000, 131072, 131073, 200000]))
        cutoff = float(rng.choice([0.0, 0.0, 0.0, 0.03, 0.1, 0.4]))
        cases.append((q, k, cutoff, tv if rng.random() < 0.25 else {}))
    nq = 0
    try:
        for q, k, cutoff, kw in cases:
            hits, approx = t.search(q, k, np.float32(cutoff), **kw)
            want, wap = O.search(q, db, k, np.float32(cutoff), nthreads=16, **kw)
            ctx = "seed %d: n=%d W=%d kind=%d k=%d cutoff=%g %s" % (seed, n, W, kind, k, cutoff, "tversky" if kw else "tanimoto")
            assert int(approx[0]) == wap, ctx
            assert_hits_equal(hits[0], want, ctx)
            nq += 1
        k = cases[0][1]  # the pipelined entry point: one k, eight queries in flight
        qs = np.ascontiguousarray(np.stack([c[0] for c in cases]))
        bufs = t.make_search_buffers(len(qs), k)
        t.search_each_into(qs, k, bufs)
        for i in range(len(qs)):
            want, wap = O.search(qs[i], db, k, np.float32(0.0), nthreads=16)
            assert int(bufs[2][i]) == wap, "seed %d pipelined %d" % (seed, i)
            assert_hits_equal(bufs[0][i][:int(bufs[1][i])], want, "seed %d pipelined %d k=%d" % (seed, i, k))
            nq += 1
    finally:
        if stats is not None:
            tm = t.timing()
            stats["queries"] = stats.get("queries", 0) + nq
            stats["published"] = stats.get("published", 0) + tm["large_k_single_scan"]
            stats["handed_back"] = stats.get("handed_back", 0) + tm["handed_back"]
            stats["why"] = stats.get("why", 0) | tm["handed_back_why"]
        t.close()


@pytest.mark.parametrize("seed", [0, 4, 9, 14, 23, 31])
def test_large_k_random_tables_match_the_oracle(seed):
    large_k_walk(seed)
