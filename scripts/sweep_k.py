#!/usr/bin/env python3
"""Latency of one query on a 100 M x 1024-bit table for different k / cutoff (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from gpusimilarity_amd import capi
import bench
N = int(os.environ.get("SW_ROWS", "100000000"))
t = capi.Table(1024); t.generate(bench.DB_SEED, 0, 0, N, 0)
qs = [bench.synth_row(bench.DB_SEED, 0, bench.query_row(i, N), 32) for i in range(12)]
cases = ((1, 0.0), (20, 0.0), (1000, 0.0), (8192, 0.0), (10000, 0.0), (100000, 0.0), (1000, 0.05), (1000, 0.3), (20, 0.5))
if os.environ.get("SW_K"):
    cases = tuple((int(x), 0.0) for x in os.environ["SW_K"].split(","))
for k, cutoff in cases:
    for q in qs[:2]:
        t.search(q, k, cutoff)
    t0 = time.perf_counter()
    for q in qs[2:]:
        h, ap = t.search(q, k, cutoff)
    el = (time.perf_counter() - t0) / 10 * 1e3
    print("k=%-7d cutoff=%-5g %.3f ms/query = %.3f of the HBM roofline  returned=%d approx=%d" % (k, cutoff, el, N * 128 / (el * 1e-3) / 8e12, len(h[0]), int(ap[0])), flush=True)
