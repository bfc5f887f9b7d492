#!/usr/bin/env python3
"""Sweep the scan kernel's launch geometry on the GPU box (GSIM_SCAN_WAVES_PER_CU x
GSIM_SCAN_UNROLL) and print scan / select / end-to-end times per configuration."""
import itertools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from gpusimilarity_amd import capi  # noqa: E402
import bench  # noqa: E402

N = int(os.environ.get("TUNE_ROWS", "100000000"))
K = int(os.environ.get("TUNE_K", "1000"))
STEPS = int(os.environ.get("TUNE_STEPS", "20"))
FP_BITS = int(os.environ.get("TUNE_BITS", "1024"))
wpcs = [int(x) for x in os.environ.get("TUNE_WPC", "4,8,12,16,24,32").split(",")]
unrolls = [int(x) for x in os.environ.get("TUNE_UNROLL", "4,8,16").split(",")]
W = FP_BITS // 32
queries = [bench.synth_row(bench.DB_SEED, 0, bench.query_row(i, N), W) for i in range(STEPS + 3)]
rows = []
for wpc, un in itertools.product(wpcs, unrolls):
    os.environ["GSIM_SCAN_WAVES_PER_CU"] = str(wpc)
    os.environ["GSIM_SCAN_UNROLL"] = str(un)
    t = capi.Table(FP_BITS)
    t.generate(bench.DB_SEED, 0, 0, N, 0)
    for q in queries[:3]:
        t.search(q, K)
    t.enable_timing(True)
    t0 = time.perf_counter()
    for q in queries[3:]:
        t.search(q, K)
    el = (time.perf_counter() - t0) / STEPS * 1e3
    tm = t.timing()
    n = max(1, tm["queries"])
    r = {"wpc": wpc, "unroll": un, "e2e_ms": round(el, 4), "scan_ms": round(tm["scan_ms_sum"] / n, 4),
         "select_ms": round(tm["select_ms_sum"] / n, 4), "cand": int(tm["candidates_sum"] / n),
         "final": int(tm["finalists_sum"] / n),
         "scan_GBs": round(N * (FP_BITS // 8) / (tm["scan_ms_sum"] / n * 1e-3) / 1e9, 1)}
    rows.append(r)
    print(json.dumps(r), flush=True)
    t.close()
best = min(rows, key=lambda r: r["e2e_ms"])
print("best", json.dumps(best))
