"""Single queries whose result block stays in DEVICE memory (gsim_db_search_device, the path the one-process-per-GPU
gather uses): us per query over a stream-ordered run of queries, against the synchronous call whose block is pinned
host memory.    python scripts/time_device_block.py rows [k ...]      (env: TS_BITS, TS_KIND)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import DB_SEED, query_row, synth_row
from gpusimilarity_amd import capi

bits = int(os.environ.get("TS_BITS", "1024"))
W = bits // 32
kind = {"sparse": capi.SYNTH_SPARSE, "morgan": capi.SYNTH_MORGAN}[os.environ.get("TS_KIND", "sparse")]
n = int(sys.argv[1])
ks = [int(x) for x in sys.argv[2:]] or [1000, 8192]
t = capi.Table(bits)
t.generate(DB_SEED, kind, 0, n, 0)
stream = torch.cuda.Stream()
t.set_stream(stream.cuda_stream)
qs = [synth_row(DB_SEED, kind, query_row(i, n), W) for i in range(16)]
for k in ks:
    blk = torch.zeros(capi.result_block_bytes(k), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for reps in (200, 2000):
        t0 = time.perf_counter()
        for i in range(reps):
            t.search_device(qs[i % 16], k, blk.data_ptr())
        stream.synchronize()
        el = time.perf_counter() - t0
    bufs = t.make_search_buffers(1, k)
    t0 = time.perf_counter()
    for i in range(500):
        t.search_into(qs[i % 16], k, bufs)
    el2 = time.perf_counter() - t0
    print("rows %d k %5d: device block %7.1f us/query (enqueued back to back), host block %7.1f us/query (synchronous)"
          % (n, k, 1e6 * el / reps, 1e6 * el2 / 500), flush=True)
