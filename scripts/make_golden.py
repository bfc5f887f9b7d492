#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REFERENCE's own scoring functor.

Run in the dev container (needs /root/reference, oracle/_ref built by
`make -C oracle ref`).  Scores come from gpusim::TanimotoFunctorCPU (the
reference's calculation_functors.cpp compiled in place, oracle/_ref); the cutoff
rule (fingerprintdb_cuda.cu:101), compaction (:263-273) and the canonical order
(score desc, row asc -- what Thrust's stable sort_by_key yields)
are applied with numpy; the integer popcounts come from an independent numpy
bit count.  The oracle (oracle/gsim_oracle.c) is NOT used to make these
vectors -- it is checked against them.

Output is data only: inputs are regenerable (small.fsim rows / the counter-based
synthetic generator parameters) and the expected outputs are listed explicitly.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402  (only for synth_rows / RefTable / numpy helpers)
from gpusimilarity_amd.fsim import read_fsim  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def popcount_rows(a):
    return np.unpackbits(np.ascontiguousarray(a).view(np.uint8), axis=-1).sum(axis=-1)


def case_from_scores(db, query, scores, k, cutoff):
    rows, sc, approx = O.canonical_topk_from_scores(scores, k, cutoff)
    common = popcount_rows(db[rows] & query[None, :]) if len(rows) else np.zeros(0, int)
    popc = popcount_rows(db[rows]) if len(rows) else np.zeros(0, int)
    return {
        "k": int(k), "cutoff": float(cutoff), "approx": int(approx),
        "rows": [int(r) for r in rows],
        "score_bits": ["%08x" % int(b) for b in sc.view(np.uint32)],
        "common": [int(c) for c in common],
        "popc_db": [int(c) for c in popc],
    }


def main():
    assert O.ref_lib() is not None, "build oracle/_ref first: make -C oracle ref"
    # ---- A: small.fsim ----------------------------------------------------
    fs = read_fsim(os.path.join(GOLD, "small.fsim"))
    db = fs.rows()
    tab = O.RefTable(db)
    out = {"source": "reference TanimotoFunctorCPU over tests/golden/small.fsim", "queries": []}
    for qrow in (0, 3):
        q = db[qrow]
        sc = tab.scan(q)
        cases = [case_from_scores(db, q, sc, k, c) for k in (10, 15) for c in (0.0, 0.1, 0.3, 0.4)]
        out["queries"].append({"query_row": qrow, "cases": cases})
    with open(os.path.join(GOLD, "small_fsim_topk.json"), "w") as f:
        json.dump(out, f, indent=1)

    # ---- B: synthetic tables ---------------------------------------------
    syn = {"source": "reference TanimotoFunctorCPU over gso_synth_rows tables", "tables": []}
    for (seed, kind, n, W) in ((0x5EED0001, 0, 5000, 32), (0x5EED0001, 1, 3000, 32), (0x5EED0003, 0, 2500, 64),
                               (0x5EED0004, 0, 777, 16)):
        db = O.synth_rows(seed, kind, 0, n, W)
        tab = O.RefTable(db)
        entry = {"seed": seed, "kind": kind, "nrows": n, "W": W, "queries": []}
        for qi in range(3):
            qrow = O.query_row(qi, n)
            q = db[qrow]
            sc = tab.scan(q)
            cases = [case_from_scores(db, q, sc, k, c) for (k, c) in
                     ((1, 0.0), (10, 0.0), (100, 0.0), (100, 0.05), (50, 0.2), (100, 0.9))]
            entry["queries"].append({"kind": "db_row", "query_row": qrow, "cases": cases})
        # a fresh random query (not a DB row)
        q = O.synth_rows(0x5EED0002, kind, 12345, 1, W)[0]
        sc = tab.scan(q)
        entry["queries"].append({"kind": "fresh", "fresh_seed": 0x5EED0002, "fresh_row": 12345,
                                 "cases": [case_from_scores(db, q, sc, k, c) for (k, c) in ((10, 0.0), (100, 0.0), (100, 0.1))]})
        syn["tables"].append(entry)
    with open(os.path.join(GOLD, "synthetic_topk.json"), "w") as f:
        json.dump(syn, f, indent=1)

    # ---- C: ties (SURVEY.md Appendix C tie experiment layout) -------------
    base = O.synth_rows(0x71E5, 0, 0, 4, 32)
    db = np.ascontiguousarray(np.tile(base, (10, 1)))  # rows r, r+4, r+8... identical
    tab = O.RefTable(db)
    q = db[0]
    sc = tab.scan(q)
    ties = {"source": "40-row table = 4 distinct synthetic FPs (seed 0x71E5) repeated 10x; query = row 0",
            "seed": 0x71E5, "cases": [case_from_scores(db, q, sc, k, c) for (k, c) in
                                      ((5, 0.0), (10, 0.0), (12, 0.0), (40, 0.0), (12, 0.5))]}
    # an all-zero query / all-zero rows table: NaN -> 0 rule
    z = np.zeros((6, 32), dtype=np.uint32)
    z[1, 0] = 0xF
    z[4, 3] = 0x1
    tabz = O.RefTable(z)
    scz = tabz.scan(z[0])  # all-zero query: 0/0 = NaN for zero rows, 0 for others
    ties["nan_cases"] = {"rows_hex_nonzero": {"1": [0, 0xF], "4": [3, 0x1]}, "nrows": 6,
                         "raw_is_nan": [bool(np.isnan(x)) for x in scz],
                         "cases": [case_from_scores(z, z[0], scz, 6, c) for c in (0.0, -1.0, 0.1)]}
    with open(os.path.join(GOLD, "ties_topk.json"), "w") as f:
        json.dump(ties, f, indent=1)
    print("golden vectors written to", GOLD)


if __name__ == "__main__":
    main()
