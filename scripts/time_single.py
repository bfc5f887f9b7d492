"""Latency of single queries through the synchronous C-ABI call (gsim_db_search) on synthetic
tables of several sizes: ms per query, HIP-event time of the dominant kernel, candidates.
    python scripts/time_single.py [rows ...]      (env: TS_BITS, TS_K, TS_REPS, TS_KIND=sparse|dense|morgan, GSIM_FUSED=0|1)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bench import DB_SEED, query_row, synth_row  # noqa: E402
from gpusimilarity_amd import capi  # noqa: E402

bits = int(os.environ.get("TS_BITS", "1024"))
k = int(os.environ.get("TS_K", "1000"))
reps = int(os.environ.get("TS_REPS", "200"))
W = bits // 32
kind = {"sparse": capi.SYNTH_SPARSE, "dense": capi.SYNTH_DENSE, "morgan": capi.SYNTH_MORGAN}[os.environ.get("TS_KIND", "sparse")]
sizes = [int(x) for x in sys.argv[1:]] or [100_000, 1_000_000, 10_000_000, 100_000_000]
for n in sizes:
    t = capi.Table(bits)
    t.generate(DB_SEED, kind, 0, n, 0)
    qs = [synth_row(DB_SEED, kind, query_row(i, n), W) for i in range(16)]
    bufs = t.make_search_buffers(1, k)
    t_warm = time.perf_counter()  # (an idle GPU takes tens of ms to clock up: short tables would be timed in that ramp)
    i = 0
    while i < 10 or time.perf_counter() - t_warm < 0.3:
        t.search_into(qs[i % 16], k, bufs)
        i += 1
    t.search_into(qs[9], k, bufs)
    h9 = bufs[0][0, :bufs[1][0]]
    assert h9["score"][0] == 1.0 and query_row(9, n) in h9["row"][h9["score"] == 1.0]
    t0 = time.perf_counter()
    for i in range(reps):
        t.search_into(qs[i % 16], k, bufs)
    el = time.perf_counter() - t0
    # the same through ONE C call per 16 queries (no Python between the queries)
    qblock = np.ascontiguousarray(np.stack(qs))
    bufs16 = t.make_search_buffers(16, k)
    t.search_each_into(qblock, k, bufs16)
    t0 = time.perf_counter()
    for i in range(max(1, reps // 16)):
        t.search_each_into(qblock, k, bufs16)
    el_c = (time.perf_counter() - t0) / (max(1, reps // 16) * 16)
    t.enable_timing(True)  # (HIP events around the kernels slow the call down: a separate, short loop)
    for i in range(64):
        t.search_into(qs[i % 16], k, bufs)
    tm = t.timing()
    nq = max(1, tm["queries"])
    us = 1e6 * el / reps
    floor = n * (bits // 8) / 8e12 * 1e6
    print("rows %11d  %8.1f us/query (C loop %7.1f)  (HBM floor %7.1f us, frac %.3f)  kernel %8.1f us  select %6.1f us  cand/q %8.0f  final/q %6.0f  handed back %d/64 (why %d)  fused=%s"
          % (n, us, 1e6 * el_c, floor, floor / us, 1e3 * tm["scan_ms_sum"] / nq, 1e3 * tm["select_ms_sum"] / nq,
             tm["candidates_sum"] / nq, tm["finalists_sum"] / nq, tm["handed_back"], tm["handed_back_why"], os.environ.get("GSIM_FUSED", "1")), flush=True)
    t.close()
