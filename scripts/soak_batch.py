#!/usr/bin/env python3
"""Soak test of the multi-query (matrix-core) pass: many batches of fresh queries, every query's result compared byte
for byte with what the single-query path returns for it (itself pinned to the oracle and soaked by soak_fused.py).
    python scripts/soak_batch.py [rows] [batches] [fp_bits]      (on the GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import DB_SEED, synth_row
from gpusimilarity_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
W = bits // 32
t = capi.Table(bits)
t.generate(DB_SEED, capi.SYNTH_SPARSE, 0, n, 0)
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "7")))
bad = done = 0
t0 = time.time()
for b in range(batches):
    Q = int(rng.choice([7, 64, 130, 256]))
    k = int(rng.choice([10, 100, 1000]))
    metric = int(rng.integers(0, 2))
    kw = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7)) if metric else {}
    cutoff = float(rng.choice([0.0, 0.0, 0.0, 0.4]))
    rows = rng.integers(0, n, size=Q)
    qs = np.ascontiguousarray(np.stack([synth_row(DB_SEED if i % 5 else DB_SEED + 3, capi.SYNTH_SPARSE, int(r), W) for i, r in enumerate(rows)]))
    a = t.make_search_buffers(Q, k)
    e = t.make_search_buffers(Q, k)
    tb = time.perf_counter()
    t.search_into(qs, k, a, cutoff, **kw)
    tb = time.perf_counter() - tb
    te = time.perf_counter()
    t.search_each_into(qs, k, e, cutoff, **kw)
    te = time.perf_counter() - te
    if os.environ.get("SOAK_SLOW") and (tb > 0.05 or te > 0.2):
        print("slow: batch %d Q %d k %d metric %d cutoff %g: shared pass %.1f ms, one by one %.1f ms" % (b, Q, k, metric, cutoff, 1e3 * tb, 1e3 * te), flush=True)
    for i in range(Q):
        c = int(e[1][i])
        if int(a[1][i]) != c or int(a[2][i]) != int(e[2][i]) or a[0][i, :c].tobytes() != e[0][i, :c].tobytes():
            bad += 1
            if bad < 5:
                print("MISMATCH batch %d query %d Q %d k %d metric %d cutoff %g" % (b, i, Q, k, metric, cutoff), flush=True)
    done += Q
print("soak batch: rows %d x %d-bit, %d queries in %d batches, %.1f s, mismatches %d" % (n, bits, done, batches, time.time() - t0, bad))
sys.exit(1 if bad else 0)
