#!/usr/bin/env python3
"""Differential fuzz of the pipelined entry point (gsim_db_search_each: two half-grid lanes on small tables, the publishing route on
them for large k, one stream beyond 4 GB) against the same handle answering one query at a time (gsim_db_search_timed: whole grid, the
route the oracle fuzz of tests/test_gpu_fuzz.py covers): random tables, widths (odd ones included), kinds, k, cutoffs, metrics.
    python scripts/fuzz_lanes.py [first_seed] [seeds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gpusimilarity_amd import capi

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0, bad, nq_total, lane_q, hb = time.time(), 0, 0, 0, 0
for seed in range(first, first + count):
    rng = np.random.default_rng(0x1A9E0000 + seed)
    W = int(rng.choice([4, 8, 16, 32, 32, 32, 64, 128, 5, 12, 28, 7]))
    n = int(np.exp(rng.uniform(np.log(70_000), np.log(6_000_000 if W <= 32 else 1_500_000))))
    kind = int(rng.choice([0, 0, 1, 2])) if W == 32 else int(rng.choice([0, 1]))
    t = capi.Table(32 * W)
    t.generate(0x1A9E + seed, kind, 0, n, 0)
    t.enable_timing(True)
    NQ = int(rng.choice([5, 16, 33, 64]))
    rows = rng.integers(0, n, size=NQ)
    qs = np.ascontiguousarray(np.stack([capi.synth_row(0x1A9E + seed, kind, int(r), 32 * W) if rng.random() < 0.8 else
                                        capi.synth_row(0x77 + seed, 0 if kind != 1 else 1, int(r), 32 * W) for r in rows]))
    for case in range(4):
        k = int(rng.choice([1, 10, 100, 1000, 2048, 3000, 8192, 20000]))
        cutoff = float(rng.choice([0.0, 0.0, 0.05, 0.3]))
        kw = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7)) if rng.random() < 0.25 else {}
        a, b = t.make_search_buffers(NQ, k), t.make_search_buffers(NQ, k)
        t.search_each_into(qs, k, a, np.float32(cutoff), **kw)
        t.search_timed_into(qs, k, b, np.float32(cutoff), **kw)
        for i in range(NQ):
            same = a[1][i] == b[1][i] and a[2][i] == b[2][i] and a[0][i, :a[1][i]].tobytes() == b[0][i, :b[1][i]].tobytes()
            if not same:
                bad += 1
                if bad < 10:
                    print("MISMATCH seed %d: n=%d W=%d kind=%d k=%d cutoff=%g %s query %d" % (seed, n, W, kind, k, cutoff, "tversky" if kw else "tanimoto", i), flush=True)
        nq_total += NQ
    tm = t.timing()
    lane_q += tm["lane_queries"]
    hb += tm["handed_back"]
    t.close()
    if time.time() - t0 > 1200:
        print("stopped at seed", seed)
        break
print("lanes fuzz: %d tables, %d queries through gsim_db_search_each vs one at a time, mismatches %d; lane queries %d, handed back %d; %.0f s"
      % (seed - first + 1, nq_total, bad, lane_q, hb, time.time() - t0))
sys.exit(1 if bad else 0)
