"""Time of one 256-query Tversky(0.3, 0.7) batch (BASELINE configs[4]'s per-GPU shape) through the synchronous C ABI:
ms per batch, HIP-event time of the batch kernels, fraction of the 10 PFLOP/s dense MX-FP4 peak.
    python scripts/time_batch.py [rows]      (env: TB_BITS, TB_Q)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import DB_SEED, query_row, synth_row
from gpusimilarity_amd import capi
bits = int(os.environ.get("TB_BITS", "2048")); Q = int(os.environ.get("TB_Q", "256")); n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
W = bits // 32; k = 1000
t = capi.Table(bits); t.generate(DB_SEED, capi.SYNTH_SPARSE, 0, n, 0)
qs = np.ascontiguousarray(np.stack([synth_row(DB_SEED, capi.SYNTH_SPARSE, query_row(i, n), W) for i in range(Q)]), dtype=np.uint32)
kw = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
bufs = t.make_search_buffers(Q, k)
for i in range(2): t.search_into(qs, k, bufs, 0.0, **kw)
t.enable_timing(True)
t0 = time.perf_counter()
for i in range(5): t.search_into(qs, k, bufs, 0.0, **kw)
el = (time.perf_counter() - t0) / 5
tm = t.timing()
kms = tm["batch_kernel_ms_sum"] / max(1, tm["batches"])
print("bits %d Q %d rows %d: %.2f ms/batch, kernel %.2f ms, frac %.3f, hits0 %d" % (bits, Q, n, 1e3 * el, kms, 2.0 * Q * n * bits / (kms * 1e-3) / 1e16, int(bufs[1][0])))
