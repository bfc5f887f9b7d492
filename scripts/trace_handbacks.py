#!/usr/bin/env python3
"""Which queries does the single launch hand back, and why are queries run twice -- one at a time against the pipelined
entry point (gsim_db_search_each, eight in flight)?  VERDICT r05 item 2: the pipelined soak counted 2011 hand-backs with
the ranked election and 1704 with the sampled one on a 300 k-row Morgan-shaped table at k = 8192.

    python scripts/trace_handbacks.py [rows] [k] [passes]      (env: SOAK_KIND=sparse|morgan; GSIM_FUSED_FLAGS, GSIM_FUSED_BACKOFF
                                                                 are read by the library when the handle is created)
Prints ONE JSON line: per mode the device's hand-back count, the host's reasons (gsim_timing.rerun_*), and per soak query
(0..47) how often its own launch handed it back / it was routed around the launch by the back-off."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import DB_SEED, query_row, synth_row
from gpusimilarity_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 20
W, NQ = 32, 48
KIND = {"sparse": capi.SYNTH_SPARSE, "morgan": capi.SYNTH_MORGAN}[os.environ.get("SOAK_KIND", "morgan")]
own = [synth_row(DB_SEED, KIND, query_row(i, n), W) for i in range(NQ - 8)]
fresh = [synth_row(DB_SEED + 7, capi.SYNTH_SPARSE, 1000 + i, W) for i in range(8)]
qs = np.ascontiguousarray(np.stack(own + fresh))
out = {"rows": n, "k": k, "kind": os.environ.get("SOAK_KIND", "morgan"), "passes": passes,
       "GSIM_FUSED_FLAGS": os.environ.get("GSIM_FUSED_FLAGS", "0"), "GSIM_FUSED_BACKOFF": os.environ.get("GSIM_FUSED_BACKOFF", "1")}
ref = None
for mode in ("one_at_a_time", "pipelined"):
    t = capi.Table(32 * W)
    t.generate(DB_SEED, KIND, 0, n, 0)
    t.enable_timing(True)
    rng = np.random.default_rng(1)
    per_q = np.zeros((NQ, 64), dtype=np.int64)  # [query][flag bit] counts
    bufs = t.make_search_buffers(NQ, k)
    for _ in range(passes):
        order = rng.permutation(NQ)
        q = np.ascontiguousarray(qs[order])
        if mode == "pipelined":
            t.search_each_into(q, k, bufs)
        else:
            t.search_timed_into(q, k, bufs)  # gsim_db_search_timed: search_one per query, nothing in flight behind it
        fl = t.query_flags(NQ)
        assert len(fl) == NQ
        for j, qi in enumerate(order):
            for b in range(6):
                per_q[qi, b] += (int(fl[j]) >> b) & 1
        got = {int(qi): (bufs[0][j, :bufs[1][j]].tobytes(), int(bufs[2][j])) for j, qi in enumerate(order)}
        if ref is None:
            ref = got
        assert got == ref, "results differ between passes / modes"
    tm = t.timing()
    out[mode] = {"device_handed_back": tm["handed_back"], "why": tm["handed_back_why"], "rerun_own": tm["rerun_own"], "rerun_publish": tm["rerun_publish"],
                 "rerun_behind": tm["rerun_behind"], "rerun_torn": tm["rerun_torn"], "backoff_skips": tm["backoff_skips"],
                 "handed_back_by_query": {str(i): int(per_q[i, 0] + per_q[i, 4]) for i in range(NQ) if per_q[i, 0] + per_q[i, 4]},
                 "skipped_by_query": {str(i): int(per_q[i, 3] + per_q[i, 5]) for i in range(NQ) if per_q[i, 3] + per_q[i, 5]},
                 "rerun_behind_by_query": {str(i): int(per_q[i, 1]) for i in range(NQ) if per_q[i, 1]}}
    t.close()
print(json.dumps(out))
