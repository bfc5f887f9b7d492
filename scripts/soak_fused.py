#!/usr/bin/env python3
"""Soak test of the single-launch query path: many thousand queries over the same table, every result compared with
the one the four-kernel pipeline (GSIM_FUSED=0, a child process) gave for that query -- looks for rare races in the
in-kernel protocol (thresholds, tickets, publication) that the parity suite's few hundred queries would not meet.
    python scripts/soak_fused.py [rows] [iterations]      (on the GPU box; SOAK_KIND=sparse|morgan, SOAK_BITS=128...8192, SOAK_LARGE_K=1)"""
import os, pickle, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import DB_SEED, query_row, synth_row
from gpusimilarity_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
W, NQ = int(os.environ.get("SOAK_BITS", "1024")) // 32, 48
cases = [(1000, 0.0), (10, 0.0), (100, 0.07), (2048, 0.0), (8192, 0.0)]
if os.environ.get("SOAK_LARGE_K") == "1":  # the single launch publishing for the large-k kernels (k in (2048, 32768]), mixed with ordinary queries
    cases = [(9000, 0.0), (1000, 0.0), (20000, 0.0), (12000, 0.05), (32768, 0.0), (100, 0.0), (3000, 0.0), (4096, 0.02)]
KIND = {"sparse": capi.SYNTH_SPARSE, "morgan": capi.SYNTH_MORGAN}[os.environ.get("SOAK_KIND", "sparse")]


def queries():
    own = [synth_row(DB_SEED, KIND, query_row(i, n), W) for i in range(NQ - 8)]
    fresh = [synth_row(DB_SEED + 7, capi.SYNTH_SPARSE, 1000 + i, W) for i in range(8)]
    return np.ascontiguousarray(np.stack(own + fresh))


def table():
    t = capi.Table(32 * W)
    t.generate(DB_SEED, KIND, 0, n, 0)
    return t


if os.environ.get("SOAK_CHILD") == "1":
    t, qs, ref = table(), queries(), {}
    for k, cutoff in cases:
        bufs = t.make_search_buffers(NQ, k)
        t.search_each_into(qs, k, bufs, cutoff)
        ref[(k, cutoff)] = (bufs[0].copy(), bufs[1].copy(), bufs[2].copy())
    pickle.dump(ref, open(sys.argv[3], "wb"))
    sys.exit(0)

path = "/tmp/soak_ref_%d.pkl" % os.getpid()
subprocess.check_call([sys.executable, __file__, str(n), "0", path], env=dict(os.environ, GSIM_FUSED="0", SOAK_CHILD="1"))
ref = pickle.load(open(path, "rb"))
t, qs = table(), queries()
t.enable_timing(True)
bad, done, t0 = 0, 0, time.time()
hb_by_case, hb_seen = {}, 0
rng = np.random.default_rng(1)
while done < iters:
    k, cutoff = cases[int(rng.integers(len(cases)))]
    bufs = t.make_search_buffers(NQ, k)
    order = rng.permutation(NQ)
    t.search_each_into(np.ascontiguousarray(qs[order]), k, bufs, cutoff)
    h, c, a = ref[(k, cutoff)]
    for j, qi in enumerate(order):
        cnt = int(c[qi])
        same = int(bufs[1][j]) == cnt and int(bufs[2][j]) == int(a[qi]) and bufs[0][j, :cnt].tobytes() == h[qi, :cnt].tobytes()
        if not same:
            bad += 1
            if bad < 5:
                print("MISMATCH iteration %d query %d k %d cutoff %g" % (done + j, qi, k, cutoff), flush=True)
    done += NQ
    hb = t.timing()["handed_back"]
    if hb != hb_seen:
        hb_by_case[(k, cutoff)] = hb_by_case.get((k, cutoff), 0) + hb - hb_seen
        hb_seen = hb
tm = t.timing()
if hb_by_case:
    print("handed back by (k, cutoff):", sorted(hb_by_case.items()))
print("soak (%s rows, %d-bit): rows %d, %d queries in %.1f s, mismatches %d, handed back %d%s, blocks rechecked %d torn %d" % (
      os.environ.get("SOAK_KIND", "sparse"), 32 * W, n, done, time.time() - t0, bad, tm["handed_back"],
      " (reasons, gsim_timing.handed_back_why: %d)" % tm["handed_back_why"] if tm["handed_back"] else "", tm["blocks_rechecked"], tm["blocks_torn"]))
sys.exit(1 if bad else 0)
