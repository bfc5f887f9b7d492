#!/usr/bin/env python3
"""Large k on SHORT tables (fewer than 64 rows per hit): microseconds per query and hand-backs, pipelined and one at a time,
for the route the environment selects (GSIM_PUBLISH_MIN_ROWS_PER_K, GSIM_FUSED, GSIM_FUSED_SELECT_MAX_K ...).
    python scripts/time_short_largek.py        -> one line per (rows, kind, k)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import DB_SEED, query_row, synth_row
from gpusimilarity_amd import capi

NQ = 48
label = " ".join("%s=%s" % (e, os.environ[e]) for e in ("GSIM_PUBLISH_MIN_ROWS_PER_K", "GSIM_FUSED", "GSIM_FUSED_SELECT_MAX_K", "GSIM_FUSED_BACKOFF") if e in os.environ) or "default"
for n in (100_000, 300_000, 1_000_000, 2_000_000):
    for kind_name, kind in (("sparse", capi.SYNTH_SPARSE), ("morgan", capi.SYNTH_MORGAN)):
        t = capi.Table(1024)
        t.generate(DB_SEED, kind, 0, n, 0)
        qs = np.ascontiguousarray(np.stack([synth_row(DB_SEED, kind, query_row(i, n), 32) for i in range(NQ)]))
        for k in (3000, 4096, 8192, 16384, 32768):
            if k * 2 > n:
                continue
            bufs = t.make_search_buffers(NQ, k)
            t.search_each_into(qs, k, bufs)
            t.enable_timing(True)
            t0 = time.perf_counter()
            for _ in range(5):
                t.search_each_into(qs, k, bufs)
            pipe = (time.perf_counter() - t0) / (5 * NQ) * 1e6
            one = float(np.median(t.search_timed_into(qs, k, bufs))) * 1e6
            tm = t.timing()
            print("%-28s rows %8d %-6s k %6d  pipelined %7.1f us  one-at-a-time %7.1f us  handed_back %3d / %d (own %d publish %d skips %d) large_k_single_scan %d"
                  % (label, n, kind_name, k, pipe, one, tm["handed_back"], 6 * NQ, tm["rerun_own"], tm["rerun_publish"], tm["backoff_skips"], tm["large_k_single_scan"]), flush=True)
            t.enable_timing(False)
        t.close()
