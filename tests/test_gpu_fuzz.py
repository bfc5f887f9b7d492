"""Seeded differential fuzz of the HIP path against the oracle: random table sizes around the
chunk / wave / sample boundaries, widths, k, cutoffs, metrics, tie-heavy tables (rows drawn
from a small alphabet), single queries and multi-query calls."""
import numpy as np
import pytest

import oracle_lib as O
from gpusimilarity_amd import capi

pytestmark = pytest.mark.gpu


def hits_equal(a, b):
    return (len(a) == len(b) and (a["row"] == b["row"]).all()
            and (a["score"].view(np.uint32) == b["score"].view(np.uint32)).all()
            and (a["common"] == b["common"]).all() and (a["popc_db"] == b["popc_db"]).all())


def random_table(rng, n, W, style):
    if style == "sparse":
        return O.synth_rows(int(rng.integers(1, 2**31)), 0, int(rng.integers(0, 10**6)), n, W)
    if style == "dense":
        return O.synth_rows(int(rng.integers(1, 2**31)), 1, 0, n, W)
    # tie-heavy: rows drawn from a small alphabet of fingerprints (plus a few empty rows)
    alpha = O.synth_rows(int(rng.integers(1, 2**31)), 0, 0, int(rng.integers(2, 40)), W)
    alpha[0] = 0
    return np.ascontiguousarray(alpha[rng.integers(0, len(alpha), size=n)])


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_against_oracle(seed):
    rng = np.random.default_rng(0xF022 + seed)
    sizes = [1, 2, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 65535, 65536, 65537, 262144 + 7, 524288 - 1, 524288 + 65, 1_200_003]
    for case in range(22):
        W = int(rng.choice([1, 3, 4, 8, 16, 32, 32, 32, 64, 64, 128]))
        n = int(rng.choice(sizes)) if rng.random() < 0.6 else int(rng.integers(1, 300_000))
        if W >= 64:
            n = min(n, 120_000)
        elif W != 32:
            n = min(n, 600_000)
        style = str(rng.choice(["sparse", "dense", "ties"]))
        db = random_table(rng, n, W, style)
        t = capi.Table(W * 32).add_rows(db).finalize(0, 1)
        for _ in range(3):
            k = int(rng.choice([0, 1, 2, 7, 20, 100, 1000, 1001, 5000, n, n + 3]))
            cutoff = float(rng.choice([0.0, 0.0, -0.5, 1e-6, 0.05, 0.2, 0.5, 1.0, 1.5]))
            metric = int(rng.choice([0, 0, 1]))
            al, be = (np.float32(rng.choice([0.0, 0.3, 0.5, 1.0])), np.float32(rng.choice([0.0, 0.7, 0.5, 1.0])))
            nq = int(rng.choice([1, 1, 1, 2, 5, 9]))
            qs = np.stack([db[rng.integers(0, n)] if rng.random() < 0.7 else
                           O.synth_rows(int(rng.integers(1, 2**31)), int(rng.integers(0, 2)), 0, 1, W)[0] for _ in range(nq)])
            kw = dict(metric=metric, alpha=al, beta=be) if metric else {}
            hits, approx = t.search(qs, k, np.float32(cutoff), **kw)
            for i in range(nq):
                want, wap = O.search(qs[i], db, k, np.float32(cutoff), nthreads=4, **kw)
                ctx = "seed=%d case=%d W=%d n=%d style=%s k=%d cutoff=%g metric=%d nq=%d q=%d" % (
                    seed, case, W, n, style, k, cutoff, metric, nq, i)
                assert int(approx[i]) == wap, ctx
                assert hits_equal(hits[i], want), ctx
        t.close()


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_large_batches_against_oracle(seed):
    """Batches of 64..200 queries (the matrix-core pass for 256..2048-bit rows and cutoff <= 0, the
    VALU pass otherwise), random table shapes, metrics and k."""
    rng = np.random.default_rng(0xBA7C4 + seed)
    for case in range(8):
        W = int(rng.choice([32, 64, 32, 64, 16]))
        n = int(rng.choice([1, 33, 255, 256, 257, 511, 513, 4097, 50_000, 100_003]))
        style = str(rng.choice(["sparse", "dense", "ties"]))
        db = random_table(rng, n, W, style)
        t = capi.Table(W * 32).add_rows(db).finalize(0, 1)
        k = int(rng.choice([1, 10, 100, 1000, n + 1]))
        cutoff = float(rng.choice([0.0, 0.0, 0.0, -1.0, 0.1]))
        metric = int(rng.choice([0, 1]))
        al, be = (np.float32(rng.choice([0.0, 0.3, 1.0])), np.float32(rng.choice([0.7, 0.5, 1.0])))
        nq = int(rng.integers(64, 201))
        qs = np.stack([db[rng.integers(0, n)] if rng.random() < 0.7 else
                       O.synth_rows(int(rng.integers(1, 2**31)), int(rng.integers(0, 2)), 0, 1, W)[0] for _ in range(nq)])
        kw = dict(metric=metric, alpha=al, beta=be) if metric else {}
        hits, approx = t.search(qs, k, np.float32(cutoff), **kw)
        for i in range(nq):
            want, wap = O.search(qs[i], db, k, np.float32(cutoff), nthreads=4, **kw)
            ctx = "seed=%d case=%d W=%d n=%d style=%s k=%d cutoff=%g metric=%d (%g,%g) nq=%d q=%d" % (
                seed, case, W, n, style, k, cutoff, metric, al, be, nq, i)
            assert int(approx[i]) == wap, ctx
            assert hits_equal(hits[i], want), ctx
        t.close()


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_generic_widths_against_oracle(seed):
    """Widths that are not a power-of-two number of 16-byte lanes (scan_generic_kernel: rows transposed through LDS,
    sample pass on large tables), single queries and small multi-query calls."""
    rng = np.random.default_rng(0x6E2 + seed)
    sizes = [1, 3, 63, 64, 65, 255, 257, 4095, 4097, 65537, 262144 + 7, 2_100_001]
    for case in range(14):
        W = int(rng.choice([2, 5, 6, 7, 9, 12, 20, 24, 28, 48, 100, 130]))
        n = int(rng.choice(sizes)) if rng.random() < 0.6 else int(rng.integers(1, 200_000))
        if W >= 48:
            n = min(n, 150_000)
        style = str(rng.choice(["sparse", "dense", "ties"]))
        db = random_table(rng, n, W, style)
        t = capi.Table(W * 32).add_rows(db).finalize(0, 1)
        for _ in range(3):
            k = int(rng.choice([0, 1, 7, 100, 1000, 3000, n, n + 3]))
            cutoff = float(rng.choice([0.0, 0.0, -0.5, 0.05, 0.3, 1.0]))
            metric = int(rng.choice([0, 0, 1]))
            al, be = (np.float32(rng.choice([0.0, 0.3, 1.0])), np.float32(rng.choice([0.0, 0.7, 1.0])))
            nq = int(rng.choice([1, 1, 2, 6]))
            qs = np.stack([db[rng.integers(0, n)] if rng.random() < 0.7 else
                           O.synth_rows(int(rng.integers(1, 2**31)), int(rng.integers(0, 2)), 0, 1, W)[0] for _ in range(nq)])
            kw = dict(metric=metric, alpha=al, beta=be) if metric else {}
            hits, approx = t.search(qs, k, np.float32(cutoff), **kw)
            for i in range(nq):
                want, wap = O.search(qs[i], db, k, np.float32(cutoff), nthreads=4, **kw)
                ctx = "generic seed=%d case=%d W=%d n=%d style=%s k=%d cutoff=%g metric=%d nq=%d q=%d" % (
                    seed, case, W, n, style, k, cutoff, metric, nq, i)
                assert int(approx[i]) == wap, ctx
                assert hits_equal(hits[i], want), ctx
        t.close()
