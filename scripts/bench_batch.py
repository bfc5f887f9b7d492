#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: Tversky(0.3, 0.7), 2048-bit fingerprints, 256-query batches,
top-1000.  The pass runs on the matrix cores (gsim_batch_mfma.hip): reported as
pairs/s, as 0/1 multiply-adds per second against the dense MX-FP4 MFMA peak, and as "effective"
bytes (Q x table bytes / time) next to the real HBM traffic (the table is read once per batch).
GSIM_BATCH_MFMA_MIN_Q=0 in the environment selects the VALU pass (32 queries per table pass)."""
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
CUTOFF = float(os.environ.get("BB_CUTOFF", "0"))
W = BITS // 32
t = capi.Table(BITS)
t.generate(bench.DB_SEED, 0, 0, N, 0)
qs = np.stack([bench.synth_row(bench.DB_SEED, 0, bench.query_row(i, N), W) for i in range(Q)])
kw = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
hits, approx = t.search(qs, K, CUTOFF, **kw)  # warm-up (allocates the batch buffers)
assert os.environ.get("BB_NOCHECK") or all(int(h["row"][0]) == bench.query_row(i, N) and h["score"][0] == 1.0 for i, h in enumerate(hits))
t0 = time.perf_counter()
for _ in range(REPS):
    t.search(qs, K, CUTOFF, **kw)
el = (time.perf_counter() - t0) / REPS
# single-query path for comparison (a few queries)
t1 = time.perf_counter()
t.search_each_into(np.ascontiguousarray(qs[:4]), K, t.make_search_buffers(4, K), 0.0, **kw)
single = (time.perf_counter() - t1) / 4
pairs = Q * N / el
mfma = int(os.environ.get("GSIM_BATCH_MFMA_MIN_Q", "4")) > 0 and Q >= int(os.environ.get("GSIM_BATCH_MFMA_MIN_Q", "4")) and W in (32, 64)
passes = 1 if mfma else (Q + 31) // 32
out = {
    "config": "Tversky(0.3,0.7) %d-bit, %d-query batch, top-%d, %d rows, cutoff %g, 1 GPU" % (BITS, Q, K, N, CUTOFF),
    "pass": "matrix cores (MX-FP4 MFMA)" if mfma else "VALU (v_and + v_bcnt)",
    "batch_s": el, "pairs_per_s": pairs, "queries_per_s": Q / el,
    "effective_GBs": Q * N * (BITS // 8) / el / 1e9,
    "hbm_GBs": passes * N * (BITS // 8) / el / 1e9,
    "single_query_path_s_per_query": single, "speedup_vs_single_query_path": single * Q / el}
if mfma:
    out["mfma_TFLOPs_equivalent"] = 2.0 * pairs * BITS / 1e12
    out["frac_of_fp4_mfma_peak_10PF"] = out["mfma_TFLOPs_equivalent"] / 10000.0
else:
    valu_ops = pairs * (2 * W)            # v_and + v_bcnt per word
    peak_lane_ops = 256 * 4 * 16 * 2.4e9  # CUs x SIMDs x (64 lanes / 4 cycles: both ops issue in 4 cycles,
    out["valu_lane_ops_per_s"] = valu_ops  # scripts/valu_op_rate_probe.hip) x 2.4 GHz
    out["frac_of_valu_issue_peak"] = valu_ops / peak_lane_ops
print(json.dumps(out))
