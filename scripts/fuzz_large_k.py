#!/usr/bin/env python3
"""Differential fuzz of the large-k routes against the oracle: tables of 150 k ... 3 M rows of 512 ... 4096 bits (i.i.d., dense,
Morgan-shaped, and tables with heavy duplication), k from 2049 to 40000, cutoffs, Tversky, the synchronous entry points
(gsim_db_search, gsim_db_search_each) -- the single launch publishes, the rows are placed by score bin and ranked inside their
bins, or (ties, k above 32768, hand-backs) the radix tail / the four-kernel pipeline answers.
    python scripts/fuzz_large_k.py [first_seed] [seeds]      (on the GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402

import oracle_lib as O  # noqa: E402
from gpusimilarity_amd import capi  # noqa: E402
from test_gpu_parity import assert_hits_equal, make_table  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
t0, bad, nq, routes = time.time(), 0, 0, {"published": 0, "handed_back": 0, "why": 0}
for seed in range(first, first + count):
    rng = np.random.default_rng(0x1A26E + seed)
    W = int(rng.choice([16, 32, 32, 32, 64, 128]))
    n = int(np.exp(rng.uniform(np.log(150_000), np.log(3_000_000 if W <= 32 else 700_000))))
    kind = int(rng.choice([0, 0, 1, O.KIND_MORGAN])) if W == 32 else int(rng.choice([0, 1]))
    db = O.synth_rows(0x1A260000 + seed, kind, 0, n, W)
    if seed % 5 == 4:  # heavy duplication: a few thousand distinct rows
        db = np.ascontiguousarray(db[rng.integers(0, int(rng.choice([8, 300, 5000])), size=n)])
    t = make_table(db)
    tv = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(rng.choice([0.3, 0.5, 1.0])), beta=np.float32(rng.choice([0.7, 0.5, 1.0])))
    cases = []
    for case in range(8):
        q = db[int(rng.integers(n))] if rng.random() < 0.8 else O.synth_rows(0x1A269999 + seed, 0 if kind != 1 else 1, 50 + case, 1, W)[0]
        k = int(rng.choice([2049, 2500, 3000, 4096, 4097, 6000, 8192, 8193, 12000, 20000, 32768, 32769, 40000]))
        cutoff = float(rng.choice([0.0, 0.0, 0.0, 0.03, 0.1, 0.4]))
        kw = tv if rng.random() < 0.25 else {}
        cases.append((q, k, cutoff, kw))
    try:
        for q, k, cutoff, kw in cases:
            hits, approx = t.search(q, k, np.float32(cutoff), **kw)
            want, wap = O.search(q, db, k, np.float32(cutoff), nthreads=16, **kw)
            ctx = "seed %d: n=%d W=%d kind=%d k=%d cutoff=%g %s" % (seed, n, W, kind, k, cutoff, "tversky" if kw else "tanimoto")
            assert int(approx[0]) == wap, ctx
            assert_hits_equal(hits[0], want, ctx)
            nq += 1
        # the pipelined entry point: the same k for a handful of queries, eight in flight
        k = cases[0][1]
        qs = np.ascontiguousarray(np.stack([c[0] for c in cases]))
        bufs = t.make_search_buffers(len(qs), k)
        t.search_each_into(qs, k, bufs)
        for i in range(len(qs)):
            want, wap = O.search(qs[i], db, k, np.float32(0.0), nthreads=16)
            assert int(bufs[2][i]) == wap, "seed %d pipelined %d" % (seed, i)
            assert_hits_equal(bufs[0][i][:int(bufs[1][i])], want, "seed %d pipelined %d k=%d" % (seed, i, k))
            nq += 1
    except AssertionError as e:
        bad += 1
        print("FAIL seed %d: %s" % (seed, str(e)[:300]), flush=True)
    tm = t.timing()
    routes["published"] += tm["large_k_single_scan"]
    routes["handed_back"] += tm["handed_back"]
    routes["why"] |= tm["handed_back_why"]
    t.close()
    if time.time() - t0 > 1500:
        print("stopped at seed", seed)
        break
print("large-k fuzz: %d tables, %d queries against the oracle, failures %d; shard queries scanned by the publishing launch %d, handed back %d (reasons %d); %.0f s"
      % (seed - first + 1, nq, bad, routes["published"], routes["handed_back"], routes["why"], time.time() - t0))
