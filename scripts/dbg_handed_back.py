"""Which of the soak's queries the single launch hands back at a given k, and their counters.
    python scripts/dbg_handed_back.py [rows] [k]      (env: SOAK_KIND=sparse|morgan)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import DB_SEED, query_row, synth_row
from gpusimilarity_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
W, NQ = 32, 48
KIND = {"sparse": capi.SYNTH_SPARSE, "morgan": capi.SYNTH_MORGAN}[os.environ.get("SOAK_KIND", "morgan")]
own = [synth_row(DB_SEED, KIND, query_row(i, n), W) for i in range(NQ - 8)]
fresh = [synth_row(DB_SEED + 7, capi.SYNTH_SPARSE, 1000 + i, W) for i in range(8)]
t = capi.Table(32 * W)
t.generate(DB_SEED, KIND, 0, n, 0)
bufs = t.make_search_buffers(1, k)
for i, q in enumerate(own + fresh):
    t.enable_timing(True)
    for _ in range(20):
        t.search_into(q, k, bufs)
    tm = t.timing()
    sc = bufs[0][0, :bufs[1][0]]["score"]
    if tm["handed_back"]:
        print("query %2d popcount %3d: handed back %2d/20 (why, since the start: %d)  published/q %7.0f  score[k-1] %.4f  rows at that score in the top k %d  distinct scores in the top k %d"
              % (i, int(np.unpackbits(q.view(np.uint8)).sum()), tm["handed_back"], tm["handed_back_why"], tm["finalists_sum"] / max(1, tm["queries"] - tm["handed_back"]) if tm["queries"] > tm["handed_back"] else -1,
                 sc[-1], int((sc == sc[-1]).sum()), len(np.unique(sc))), flush=True)
print("done")
