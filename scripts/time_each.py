#!/usr/bin/env python3
"""gsim_db_search_each (up to eight queries enqueued ahead): microseconds per query by table size, kind and k, for the route the
environment selects (GSIM_EACH_LANES=0|1: consecutive queries alternate between two half-grid lanes; GSIM_EACH_LANES_MAX_MB).
Results of the call are compared with one-at-a-time results of the same handle (gsim_db_search_timed).
    python scripts/time_each.py [rows ...]        (env: TE_K, TE_KINDS=sparse,morgan)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import DB_SEED, query_row, synth_row
from gpusimilarity_amd import capi

NQ = int(os.environ.get("TE_NQ", "512"))
BITS = int(os.environ.get("TE_BITS", "1024"))
ks = [int(x) for x in os.environ.get("TE_K", "1000").split(",")]
kinds = os.environ.get("TE_KINDS", "sparse,morgan").split(",")
sizes = [int(x) for x in sys.argv[1:]] or [100_000, 300_000, 1_000_000, 2_000_000, 4_000_000, 8_000_000]
label = " ".join("%s=%s" % (e, os.environ[e]) for e in ("GSIM_EACH_LANES", "GSIM_EACH_LANES_SHARE", "GSIM_EACH_LANES_MAX_MB", "TE_TIMING", "GSIM_FUSED_PUBLISH_MAX_K", "GSIM_LARGEK_BINRANK", "GSIM_PUBLISH_NARROW", "TE_BITS", "GSIM_EACH_LANES_PUBLISH") if e in os.environ) or "default"
for n in sizes:
    for kn in kinds:
        kind = {"sparse": capi.SYNTH_SPARSE, "morgan": capi.SYNTH_MORGAN}[kn]
        t = capi.Table(BITS)
        t.generate(DB_SEED, kind, 0, n, 0)
        qs = np.ascontiguousarray(np.stack([synth_row(DB_SEED, kind, query_row(i % 64, n), BITS // 32) for i in range(NQ)]))
        for k in ks:
            bufs = t.make_search_buffers(NQ, k)
            ref = t.make_search_buffers(64, k)
            sec = t.search_timed_into(qs[:64], k, ref)
            t_warm = time.perf_counter()
            while time.perf_counter() - t_warm < float(os.environ.get("TE_WARM_S", "0.3")):
                t.search_each_into(qs, k, bufs)
            t.enable_timing(os.environ.get("TE_TIMING", "0") == "1")  # (HIP events around every kernel lengthen the gaps between launches)
            reps = int(os.environ.get("TE_REPS", "6"))
            t0 = time.perf_counter()
            for _ in range(reps):
                t.search_each_into(qs, k, bufs)
            us = (time.perf_counter() - t0) / (reps * NQ) * 1e6
            tm = t.timing()
            t.enable_timing(False)
            same = all(bufs[1][i] == ref[1][i % 64] and bufs[2][i] == ref[2][i % 64] and
                       bufs[0][i, :bufs[1][i]].tobytes() == ref[0][i % 64, :ref[1][i % 64]].tobytes() for i in range(NQ))
            print("%-22s rows %9d %-6s k %5d  each %7.2f us/query  one-at-a-time median %7.2f us  kernel(avg, HIP events) %7.2f us  lane_queries %5d handed_back %d  %s"
                  % (label, n, kn, k, us, float(np.median(sec)) * 1e6, 1e3 * tm["scan_ms_sum"] / max(1, tm["queries"]), tm["lane_queries"], tm["handed_back"],
                     "identical" if same else "RESULTS DIFFER"), flush=True)
        t.close()
