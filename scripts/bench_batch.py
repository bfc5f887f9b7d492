#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: Tversky(0.3, 0.7), 2048-bit fingerprints, 256-query batches,
top-1000.  This configuration is VALU-bound (2 VALU ops per 32-bit word per (query, row)
pair), so it is reported as pairs/s, as a fraction of the v_and/v_bcnt issue rate, and as
"effective" bytes (Q x table bytes / time) next to the real HBM traffic (passes x table)."""
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

N = int(os.environ.get("BB_ROWS", "125000000"))
BITS = int(os.environ.get("BB_BITS", "2048"))
Q = int(os.environ.get("BB_Q", "256"))
K = int(os.environ.get("BB_K", "1000"))
REPS = int(os.environ.get("BB_REPS", "3"))
W = BITS // 32
t = capi.Table(BITS)
t.generate(bench.DB_SEED, 0, 0, N, 0)
qs = np.stack([bench.synth_row(bench.DB_SEED, 0, bench.query_row(i, N), W) for i in range(Q)])
kw = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
hits, approx = t.search(qs, K, 0.0, **kw)  # warm-up (allocates the batch buffers)
assert all(int(h["row"][0]) == bench.query_row(i, N) and h["score"][0] == 1.0 for i, h in enumerate(hits))
t0 = time.perf_counter()
for _ in range(REPS):
    t.search(qs, K, 0.0, **kw)
el = (time.perf_counter() - t0) / REPS
# single-query path for comparison (a few queries)
os.environ["GSIM_BATCH"] = "0"
t1 = time.perf_counter()
t.search(qs[:4], K, 0.0, **kw)
single = (time.perf_counter() - t1) / 4
pairs = Q * N / el
valu_ops = pairs * (2 * W)               # v_and + v_bcnt per word
peak_lane_ops = 256 * 4 * 32 * 2.4e9     # CUs x SIMDs x 32 lanes/clk x 2.4 GHz (MI355X_MICROARCH.md: 2 cyc per wave64 VALU op)
print(json.dumps({
    "config": "Tversky(0.3,0.7) %d-bit, %d-query batch, top-%d, %d rows, 1 GPU" % (BITS, Q, K, N),
    "batch_s": el, "pairs_per_s": pairs, "queries_per_s": Q / el,
    "valu_lane_ops_per_s": valu_ops, "frac_of_valu_issue_peak": valu_ops / peak_lane_ops,
    "effective_GBs": Q * N * (BITS // 8) / el / 1e9,
    "hbm_GBs": ((Q + 31) // 32) * N * (BITS // 8) / el / 1e9,
    "single_query_path_s_per_query": single, "speedup_vs_single_query_path": single * Q / el}))
