#!/usr/bin/env python3
"""Time one multi-query batch (no result checks: usable with ablated builds via GSIM_LIB)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from gpusimilarity_amd import capi  # noqa: E402
import bench  # noqa: E402

N = int(os.environ.get("BB_ROWS", "32000000"))
BITS = int(os.environ.get("BB_BITS", "2048"))
Q = int(os.environ.get("BB_Q", "256"))
K = int(os.environ.get("BB_K", "1000"))
REPS = int(os.environ.get("BB_REPS", "3"))
CUTOFF = float(os.environ.get("BB_CUTOFF", "0"))
W = BITS // 32
t = capi.Table(BITS)
t.generate(bench.DB_SEED, 0, 0, N, 0)
qs = np.stack([bench.synth_row(bench.DB_SEED, 0, bench.query_row(i, N), W) for i in range(Q)])
kw = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
bufs = t.make_search_buffers(Q, K)
qs = np.ascontiguousarray(qs, dtype=np.uint32)
t.search_into(qs, K, bufs, CUTOFF, **kw)
t0 = time.perf_counter()
for _ in range(REPS):
    t.search_into(qs, K, bufs, CUTOFF, **kw)  # the C ABI call alone (no per-batch Python allocations)
approx = bufs[2]
el = (time.perf_counter() - t0) / REPS
print("%s rows=%d bits=%d Q=%d cutoff=%g: %.2f ms/batch, %.3e pairs/s, mean approx %.1f" % (
    os.environ.get("GSIM_LIB", "default"), N, BITS, Q, CUTOFF, el * 1e3, Q * N / el, float(np.mean(approx))))
