#!/usr/bin/env python3
"""Folded search (fingerprintdb_cuda.cu:184-194, 284-331): ms per query at fold factor F with the candidates'
re-score on the device (full rows in HBM as well) or on the host (GSIM_FOLD_RESCORE=host), next to the unfolded search.
    python scripts/time_folded.py [rows]   (env: TF_FOLD, TF_K)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle_lib as O  # noqa: E402  (only its row generator: tables for add_rows are made on the host)
from gpusimilarity_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
F = int(os.environ.get("TF_FOLD", "8"))
k = int(os.environ.get("TF_K", "1000"))
W = 32
db = O.synth_rows(0x5EED0001, 2, 0, n, W)
qs = [db[O.query_row(i, n)] for i in range(8)]
for fold in (1, F):
    t = capi.Table(1024)
    if fold > 1:
        t.set_fold_factor(fold)
    t.add_rows(db).finalize(0, 1)
    for i in range(5):
        t.search(qs[i % 8], k, 0.0)
    t0 = time.perf_counter()
    reps = 30
    for i in range(reps):
        t.search(qs[i % 8], k, 0.0)
    ms = 1e3 * (time.perf_counter() - t0) / reps
    print("rows %d  fold %d  k %d  %.3f ms/query  (re-score: %s)" % (n, fold, k, ms, "-" if fold == 1 else os.environ.get("GSIM_FOLD_RESCORE", "device")))
    t.close()
