"""The queries of tests/test_gpu_fused.py::test_single_launch_path_is_taken_and_exact for one table, naming those handed back.
    python scripts/dbg_fused_test.py rows words"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from gpusimilarity_amd import capi
n, W = int(sys.argv[1]), int(sys.argv[2])
db = O.synth_rows(0xF05ED + n, 0, 0, n, W)
t = capi.Table(W * 32).add_rows(db).finalize(0, 1)
t.enable_timing(True)
prev = 0
def run(q, k, cutoff=0.0, name="", **kw):
    global prev
    hits, ap = t.search(q, k, cutoff, **kw)
    tm = t.timing()
    if tm["handed_back"] != prev:
        print("HANDED BACK:", name, "k", k, "cutoff", cutoff, kw, "hits", len(hits[0]), "why", tm["handed_back_why"])
    prev = tm["handed_back"]
for qi in range(3):
    q = db[O.query_row(qi, n)]
    for k in (1, 10, 1000, 2048, 4096, 8192):
        run(q, k, 0.0, "q%d" % qi)
    run(q, 100, 0.05, "cutoff")
    run(q, 100, 0.9, "high cutoff")
    run(q, 50, 0.0, "tversky", metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
run(O.synth_rows(0xF05EE, 0, 5, 1, W)[0], 1000, 0.0, "fresh")
run(np.zeros(W, dtype=np.uint32), 10, 0.0, "zero")
print("done", prev)
