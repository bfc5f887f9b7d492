"""Multi-query batches with a cutoff on the BASELINE configs[4] per-GPU shape (Tversky 0.3/0.7, 2048-bit rows, 256 queries):
ms per batch and the matrix-core kernels' time for cutoffs from selective to dense, kept rows per query.
    python scripts/time_batch_cutoff.py [rows] [cutoff ...]      (env: TB_BITS, TB_Q, GSIM_BATCH_MFMA_DENSE=0 = the VALU route)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import DB_SEED, query_row, synth_row
from gpusimilarity_amd import capi

bits = int(os.environ.get("TB_BITS", "2048"))
nq = int(os.environ.get("TB_Q", "256"))
W = bits // 32
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
cutoffs = [float(x) for x in sys.argv[2:]] or [0.0, 0.3, 0.15, 0.1, 0.05]
t = capi.Table(bits)
t.generate(DB_SEED, capi.SYNTH_SPARSE, 0, n, 0)
qs = np.ascontiguousarray(np.stack([synth_row(DB_SEED, capi.SYNTH_SPARSE, query_row(i, n), W) for i in range(nq)]))
kw = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
for cutoff in cutoffs:
    t.search(qs, 1000, np.float32(cutoff), **kw)
    t.enable_timing(True)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        hits, approx = t.search(qs, 1000, np.float32(cutoff), **kw)
    el = (time.perf_counter() - t0) / reps
    tm = t.timing()
    t.enable_timing(False)
    print("rows %d x %d-bit, %d queries, cutoff %.3f: %7.2f ms per batch, matrix-core kernels %7.2f ms, kept per query (mean) %.0f"
          % (n, bits, nq, cutoff, 1e3 * el, tm["batch_kernel_ms_sum"] / max(1, tm["batches"]), float(np.mean(approx))), flush=True)
