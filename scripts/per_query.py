"""Kernel time and counters of each of 16 single queries (HIP events), to see which queries are slow.
    python scripts/per_query.py rows      (env: TS_BITS, TS_K, TS_KIND=sparse|dense|morgan)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import DB_SEED, query_row, synth_row  # noqa: E402
from gpusimilarity_amd import capi  # noqa: E402

bits = int(os.environ.get("TS_BITS", "1024"))
k = int(os.environ.get("TS_K", "1000"))
W = bits // 32
kind = {"sparse": capi.SYNTH_SPARSE, "dense": capi.SYNTH_DENSE, "morgan": capi.SYNTH_MORGAN}[os.environ.get("TS_KIND", "sparse")]
n = int(sys.argv[1])
t = capi.Table(bits)
t.generate(DB_SEED, kind, 0, n, 0)
bufs = t.make_search_buffers(1, k)
qs = [synth_row(DB_SEED, kind, query_row(i, n), W) for i in range(16)]
for i in range(200):
    t.search_into(qs[i % 16], k, bufs)
for i in range(16):
    t.enable_timing(False)
    t.enable_timing(True)
    for _ in range(16):
        t.search_into(qs[i], k, bufs)
    tm = t.timing()
    nq = max(1, tm["queries"])
    sc = bufs[0][0, :bufs[1][0]]["score"]
    print("query %2d  kernel %7.1f us  cand %8.0f  published %6.0f  handed back %d  score[k-1] %.4f  ties at the cut %d"
          % (i, 1e3 * tm["scan_ms_sum"] / nq, tm["candidates_sum"] / nq, tm["finalists_sum"] / nq, tm["handed_back"],
             sc[-1], int((sc == sc[-1]).sum())), flush=True)
