"""Pins oracle/gsim_oracle.c against the reference's own known answers.

(1) reference test/test_gpusim.cpp:101-128 TestSimilarityCutoff (counts / approx),
(2) :134-146 CPUSort, (3) :148-166 FoldFingerprint,
(4) the vectors captured from the RUNNING reference for small.fsim (SURVEY.md
    Appendix C -- transcribed below as literals),
(5) tests/golden/*.json -- made by scripts/make_golden.py from the reference's own
    TanimotoFunctorCPU (oracle/_ref), and
(6) oracle/_ref itself when it is present (dev container).
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from gpusimilarity_amd.fsim import read_fsim

# row, common, popc_db, score bits -- output of the reference's
# search()/search_cpu() (both agreed) on test/small.fsim, k=10, cutoff 0.
APPENDIX_C = {
    0: [(0, 40, 40, "3f800000"), (74, 20, 31, "3ec8c8c9"), (42, 19, 38, "3ea4e1a1"), (10, 16, 31, "3e94f209"),
        (99, 14, 28, "3e84bda1"), (4, 16, 38, "3e842108"), (96, 13, 25, "3e800000"), (97, 12, 21, "3e7ac688"),
        (95, 13, 29, "3e6db6db"), (25, 13, 32, "3e61a08b")],
    3: [(3, 32, 32, "3f800000"), (92, 17, 33, "3eb55555"), (86, 16, 31, "3eae4c41"), (67, 13, 22, "3ea2576a"),
        (30, 14, 28, "3e9bd37a"), (16, 17, 41, "3e9b6db7"), (8, 11, 20, "3e895da9"), (81, 12, 26, "3e8590b2"),
        (14, 12, 27, "3e82b931"), (25, 13, 32, "3e828283")],
}
APPENDIX_C_NEXT5 = {0: [98, 73, 59, 31, 2], 3: [84, 85, 62, 59, 33]}
APPENDIX_C_APPROX = {0: {0.1: 86, 0.3: 3, 0.4: 1}, 3: {0.1: 88, 0.3: 6, 0.4: 1}}


@pytest.fixture(scope="module")
def small(golden_dir):
    return read_fsim(os.path.join(golden_dir, "small.fsim"))


def bits(a):
    return ["%08x" % int(x) for x in np.asarray(a, dtype=np.float32).view(np.uint32)]


def check_case(db, q, case, **kw):
    hits, approx = O.search(q, db, case["k"], case["cutoff"], **kw)
    assert approx == case["approx"]
    assert [int(r) for r in hits["row"]] == case["rows"]
    assert bits(hits["score"]) == case["score_bits"]
    assert [int(c) for c in hits["common"]] == case["common"]
    assert [int(c) for c in hits["popc_db"]] == case["popc_db"]


def test_small_fsim_fixture(small):
    assert (small.version, small.dbkey, small.fp_bitcount, small.fp_count) == (3, "pass", 1024, 100)
    assert small.ids[3] == b"ZINC00000022" and small.ids[0] == b"ZINC00000007"


def test_reference_kat_similarity_cutoff(small):
    """test/test_gpusim.cpp:101-128"""
    db = small.rows()
    for cutoff, n_ret, approx in zip((0.0, 0.1, 0.3, 0.4), (10, 10, 3, 1), (100, 86, 3, 1)):
        hits, ap = O.search(db[0], db, 10, np.float32(cutoff))
        assert len(hits) == n_ret
        assert ap == approx


def test_reference_kat_cpusort():
    """test/test_gpusim.cpp:134-146"""
    idx, sc = O.bubble_sort([0, 1, 2, 3, 4, 5], [1, 3, 2, 4, 0, 7], 3)
    assert idx[0] == 5 and sc[0] == 7
    assert idx[2] == 1 and sc[2] == 3


def test_reference_kat_fold():
    """test/test_gpusim.cpp:148-166"""
    assert list(O.fold([32, 24, 11, 7], 2)) == [43, 31]
    assert list(O.fold([32, 24, 11, 7], 4)) == [63]


def test_appendix_c_vectors(small):
    db = small.rows()
    for qrow, rows in APPENDIX_C.items():
        hits, ap = O.search(db[qrow], db, 15, 0.0)
        assert ap == 100
        got = [(int(h["row"]), int(h["common"]), int(h["popc_db"]), b) for h, b in zip(hits[:10], bits(hits["score"][:10]))]
        assert got == rows
        assert [int(r) for r in hits["row"][10:15]] == APPENDIX_C_NEXT5[qrow]
        for cutoff, approx in APPENDIX_C_APPROX[qrow].items():
            assert O.search(db[qrow], db, 10, np.float32(cutoff))[1] == approx
        # search_cpu (fingerprintdb_cuda.cpp:20-54) agrees on a tie-free top-15
        r, s = O.search_cpu(db[qrow], db, 15)
        assert list(r) == [int(x) for x in hits["row"]]
        assert bits(s) == bits(hits["score"])


def test_golden_small_fsim(small, golden_dir):
    db = small.rows()
    g = json.load(open(os.path.join(golden_dir, "small_fsim_topk.json")))
    for qe in g["queries"]:
        for case in qe["cases"]:
            check_case(db, db[qe["query_row"]], case)


def test_golden_synthetic(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "synthetic_topk.json")))
    for t in g["tables"]:
        db = O.synth_rows(t["seed"], t["kind"], 0, t["nrows"], t["W"])
        for qe in t["queries"]:
            if qe["kind"] == "db_row":
                assert qe["query_row"] == O.query_row(t["queries"].index(qe), t["nrows"])
                q = db[qe["query_row"]]
            else:
                q = O.synth_rows(qe["fresh_seed"], t["kind"], qe["fresh_row"], 1, t["W"])[0]
            for case in qe["cases"]:
                check_case(db, q, case)
                check_case(db, q, case, nthreads=3)


def test_golden_ties_and_nan(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "ties_topk.json")))
    base = O.synth_rows(g["seed"], 0, 0, 4, 32)
    db = np.ascontiguousarray(np.tile(base, (10, 1)))
    for case in g["cases"]:
        check_case(db, db[0], case)
        check_case(db, db[0], case, nthreads=4)
    # SURVEY App. C tie experiment: the tie group keeps its LOWEST row indices
    hits, _ = O.search(db[0], db, 5, 0.0)
    assert [int(r) for r in hits["row"]] == [0, 4, 8, 12, 16]
    nc = g["nan_cases"]
    z = np.zeros((nc["nrows"], 32), dtype=np.uint32)
    for r, (w, v) in nc["rows_hex_nonzero"].items():
        z[int(r), w] = v
    raw, _, _ = O.tanimoto_raw(z[0], z)
    assert [bool(np.isnan(x)) for x in raw] == nc["raw_is_nan"]
    for case in nc["cases"]:
        check_case(z, z[0], case)


def test_synth_density():
    sp = O.synth_rows(0x5EED0001, O.KIND_SPARSE, 0, 2000, 32)
    de = O.synth_rows(0x5EED0001, O.KIND_DENSE, 0, 2000, 32)
    ds = np.unpackbits(sp.view(np.uint8)).mean()
    dd = np.unpackbits(de.view(np.uint8)).mean()
    assert abs(ds - 1 / 16) < 0.003 and abs(dd - 0.5) < 0.005
    # regenerable row by row
    assert (O.synth_rows(0x5EED0001, 0, 1234, 3, 32) == sp[1234:1237]).all()


def test_tversky_identity_and_definition():
    """Tversky(1,1) == Tanimoto bit for bit (the only reference cross-check)."""
    db = O.synth_rows(0x5EED0001, 0, 0, 3000, 32)
    q = db[17]
    a, _ = O.search(q, db, 200, 0.0)
    b, _ = O.search(q, db, 200, 0.0, metric=O.METRIC_TVERSKY, alpha=1.0, beta=1.0)
    assert (a == b).all()
    L = O.lib()
    s = L.gso_score_one(O.METRIC_TVERSKY, 0.3, 0.7, 60, 70, 20)
    t1 = np.float32(0.3) * np.float32(40)
    t2 = np.float32(0.7) * np.float32(50)
    expect = np.float32(20) / np.float32(np.float32(t1 + t2) + np.float32(20))
    assert np.float32(s) == expect


def test_merge_matches_whole_table_search():
    db = O.synth_rows(0x5EED0009, 0, 0, 4000, 32)
    q = db[5]
    whole, ap = O.search(q, db, 64, 0.0)
    parts = []
    for g in range(4):
        h, _ = O.search(q, db[g * 1000:(g + 1) * 1000], 64, 0.0, row_base=g * 1000)
        parts.append(h)
    merged = O.merge_hits(parts, 64)
    assert (merged == whole).all()


@pytest.mark.skipif(O.ref_lib() is None, reason="oracle/_ref not built (dev container only)")
def test_against_reference_functors_live():
    rng = np.random.default_rng(7)
    for (n, W, kind) in ((4000, 32, 0), (1500, 64, 1), (999, 5, 0)):
        db = O.synth_rows(0xABCDEF, kind, 0, n, W)
        tab = O.RefTable(db)
        for _ in range(3):
            q = db[rng.integers(0, n)]
            ref = tab.scan(q, nthreads=2)
            mine, _, _ = O.tanimoto_raw(q, db)
            assert (ref.view(np.uint32) == mine.view(np.uint32)).all()
    for factor in (2, 4, 8):
        fp = rng.integers(-2**31, 2**31 - 1, size=32, dtype=np.int64).astype(np.int32)
        assert (O.ref_fold(fp, factor) == O.fold(fp, factor)).all()


@pytest.mark.skipif(O.ref_sort_lib() is None, reason="oracle/_ref sort lib not built / Qt missing")
def test_against_reference_bubble_sort_live():
    import ctypes as C
    L = O.ref_sort_lib()
    L.gsref_bubble_sort.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int, C.c_int]
    rng = np.random.default_rng(3)
    for n, k in ((6, 3), (50, 10), (200, 200), (64, 1)):
        sc = rng.integers(0, 12, size=n).astype(np.float32)  # many ties
        idx = np.arange(n, dtype=np.int32)
        ri, rs = idx.copy(), sc.copy()
        L.gsref_bubble_sort(ri.ctypes.data_as(C.POINTER(C.c_int)), rs.ctypes.data_as(C.POINTER(C.c_float)), n, k)
        mi, ms = O.bubble_sort(idx, sc, k)
        assert (ri == mi).all() and (rs == ms).all()
