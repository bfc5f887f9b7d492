"""Parity of the HIP path (through the C ABI) against the oracle and the golden
vectors.  Bar: bit-exact rows, integer popcounts, approx counts AND float score
bit patterns (the only float op is one correctly rounded f32 divide).

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from gpusimilarity_amd import capi
from gpusimilarity_amd.fingerprintdb import FingerprintDB, get_gpu_count, get_next_gpu
from gpusimilarity_amd.fsim import read_fsim

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def assert_hits_equal(got, want, ctx=""):
    assert len(got) == len(want), "%s: %d hits, oracle %d" % (ctx, len(got), len(want))
    assert (got["row"] == want["row"]).all(), ctx
    assert (bits(got["score"]) == bits(want["score"])).all(), ctx
    assert (got["common"] == want["common"]).all(), ctx
    assert (got["popc_db"] == want["popc_db"]).all(), ctx


ORACLE_THREADS = min(64, os.cpu_count() or 1)
_host_tables = {}


def _host_table(seed, n, W):
    """Synthetic table regenerated on the host (twin of gsim_db_generate), kept for the tests that share it."""
    key = (seed, n, W)
    if key not in _host_tables:
        _host_tables.clear()  # one at a time: they are gigabytes
        _host_tables[key] = O.synth_rows_mt(seed, 0, 0, n, W)
    return _host_tables[key]


def make_table(db, device=0, ndevices=1):
    t = capi.Table(db.shape[1] * 32)
    t.add_rows(db)
    t.finalize(device, ndevices)
    return t


def check_against_oracle(t, db, q, k, cutoff=0.0, ctx="", **kw):
    hits, approx = t.search(q, k, cutoff, **kw)
    okw = {}
    if "metric" in kw:
        okw = dict(metric=kw["metric"], alpha=kw.get("alpha", 1.0), beta=kw.get("beta", 1.0))
    want, wapprox = O.search(q, db, k, cutoff, nthreads=8, **okw)
    assert int(approx[0]) == wapprox, ctx
    assert_hits_equal(hits[0], want, ctx)
    return hits[0]


# ---------------------------------------------------------------------------
# the f32 divide, pinned bit for bit over the whole integer domain
# ---------------------------------------------------------------------------

def test_score_arithmetic_bit_exact():
    L = O.lib()
    for a in (0, 1, 37, 64, 512, 1024):
        tab = capi.debug_score_table(capi.METRIC_TANIMOTO, 0.0, 0.0, a, 1100, 64 if a > 64 else a)
        for c in range(tab.shape[0]):
            for b in (0, 1, c, c + 1, 2 * c + 3, 77, 500, 1024, 1100):
                if b > 1100:
                    continue
                ref = np.float32(L.gso_score_one(0, 0.0, 0.0, a, b, c))
                got = tab[c, b]
                assert bits(got) == bits(ref) or (np.isnan(got) and np.isnan(ref)), (a, b, c)
    # dense sweep for one query popcount: every (c, b) pair
    a = 48
    tab = capi.debug_score_table(capi.METRIC_TANIMOTO, 0.0, 0.0, a, 2048, a)
    cc, bb = np.meshgrid(np.arange(a + 1), np.arange(2049), indexing="ij")
    with np.errstate(invalid="ignore", divide="ignore"):
        ref = cc.astype(np.float32) / (a + bb - cc).astype(np.float32)
    same = (bits(tab) == bits(ref)) | (np.isnan(tab) & np.isnan(ref))
    assert same.all()
    # Tversky: same expression order as the oracle
    al, be = np.float32(0.3), np.float32(0.7)
    tab = capi.debug_score_table(capi.METRIC_TVERSKY, al, be, 60, 300, 60)
    for c in range(0, 61, 7):
        for b in range(c, 301, 13):
            ref = np.float32(L.gso_score_one(1, al, be, 60, b, c))
            assert bits(tab[c, b]) == bits(ref) or (np.isnan(tab[c, b]) and np.isnan(ref)), (b, c)


# ---------------------------------------------------------------------------
# the reference's own tests (test/test_gpusim.cpp), through the FingerprintDB twin
# ---------------------------------------------------------------------------

@pytest.fixture(scope="module")
def small_db():
    fs = read_fsim(os.path.join(GOLD, "small.fsim"))
    db = FingerprintDB(fs.fp_bitcount, fs.fp_count, fs.dbkey, fs.fp_blocks, list(fs.smiles), list(fs.ids))
    db.copyToGPU(1, device=0)
    return db, fs


def test_ref_CompareGPUtoCPU(small_db):
    """test_gpusim.cpp:29-69 -- rand()%20 == 3 in the reference's per-test process."""
    db, fs = small_db
    fp = db.getFingerprint(3)
    for return_count in (10, 15):
        gs, gi, gsc, _ = db.search(fp, "pass", return_count, 0.0)
        cs, ci, csc = db.search_cpu(fp, "pass", return_count, 0.0)
        assert len(gs) == return_count
        assert gs == cs and gi == ci
        assert (bits(gsc) == bits(csc)).all()
    assert db.search(fp, "wrong-key", 10, 0.0) == ([], [], [], 0)  # fingerprintdb_cuda.cu:349-352


def test_ref_TestSimilarityCutoff(small_db):
    """test_gpusim.cpp:101-128 -- the reference's numeric known-answer test."""
    db, fs = small_db
    fp = db.getFingerprint(0)
    for cutoff, n_ret, approx in zip((0, 0.1, 0.3, 0.4), (10, 10, 3, 1), (100, 86, 3, 1)):
        smiles, ids, scores, ap = db.search(fp, "pass", 10, cutoff)
        assert len(smiles) == n_ret
        assert ap == approx
    assert db.getID(3) == b"ZINC00000022"  # the id TestSearchMultiple pins (:97)


def test_ref_getNextGPU():
    """test_gpusim.cpp:168-181"""
    n = get_gpu_count()
    assert n >= 1
    first = [get_next_gpu(1) for _ in range(n)]
    second = [get_next_gpu(1) for _ in range(n)]
    assert sorted(first) == list(range(n)) and first == second


def test_golden_small_fsim(small_db):
    db, fs = small_db
    rows = fs.rows()
    g = json.load(open(os.path.join(GOLD, "small_fsim_topk.json")))
    for qe in g["queries"]:
        for case in qe["cases"]:
            h, ap = db.search_hits(rows[qe["query_row"]], case["k"], case["cutoff"])
            assert ap == case["approx"]
            assert [int(r) for r in h["row"]] == case["rows"]
            assert ["%08x" % int(b) for b in bits(h["score"])] == case["score_bits"]
            assert [int(c) for c in h["common"]] == case["common"]
            assert [int(c) for c in h["popc_db"]] == case["popc_db"]


def test_golden_synthetic_and_ties():
    g = json.load(open(os.path.join(GOLD, "synthetic_topk.json")))
    for tb in g["tables"]:
        db = O.synth_rows(tb["seed"], tb["kind"], 0, tb["nrows"], tb["W"])
        t = make_table(db)
        for qe in tb["queries"]:
            q = db[qe["query_row"]] if qe["kind"] == "db_row" else \
                O.synth_rows(qe["fresh_seed"], tb["kind"], qe["fresh_row"], 1, tb["W"])[0]
            for case in qe["cases"]:
                h, ap = t.search(q, case["k"], case["cutoff"])
                assert int(ap[0]) == case["approx"]
                assert [int(r) for r in h[0]["row"]] == case["rows"]
                assert ["%08x" % int(b) for b in bits(h[0]["score"])] == case["score_bits"]
                assert [int(c) for c in h[0]["common"]] == case["common"]
                assert [int(c) for c in h[0]["popc_db"]] == case["popc_db"]
        t.close()
    g = json.load(open(os.path.join(GOLD, "ties_topk.json")))
    base = O.synth_rows(g["seed"], 0, 0, 4, 32)
    db = np.ascontiguousarray(np.tile(base, (10, 1)))
    t = make_table(db)
    for case in g["cases"]:
        h, ap = t.search(db[0], case["k"], case["cutoff"])
        assert [int(r) for r in h[0]["row"]] == case["rows"] and int(ap[0]) == case["approx"]
        assert ["%08x" % int(b) for b in bits(h[0]["score"])] == case["score_bits"]
        assert [int(c) for c in h[0]["common"]] == case["common"]
        assert [int(c) for c in h[0]["popc_db"]] == case["popc_db"]
    nc = g["nan_cases"]
    z = np.zeros((nc["nrows"], 32), dtype=np.uint32)
    for r, (w, v) in nc["rows_hex_nonzero"].items():
        z[int(r), w] = v
    tz = make_table(z)
    for case in nc["cases"]:
        h, ap = tz.search(z[0], case["k"], case["cutoff"])
        assert [int(r) for r in h[0]["row"]] == case["rows"] and int(ap[0]) == case["approx"]
        assert ["%08x" % int(b) for b in bits(h[0]["score"])] == case["score_bits"]
        assert [int(c) for c in h[0]["common"]] == case["common"]
        assert [int(c) for c in h[0]["popc_db"]] == case["popc_db"]


# ---------------------------------------------------------------------------
# seeded tables vs the oracle
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("n,W,kind", [(200_000, 32, 0), (150_000, 32, 1), (60_000, 64, 0), (70_001, 16, 0),
                                      (33_333, 8, 1), (20_000, 4, 0), (9_000, 128, 0), (3_000, 256, 1),
                                      (5_000, 5, 0), (4_097, 33, 1), (10_000, 1, 1)])
def test_seeded_tables_match_oracle(n, W, kind):
    db = O.synth_rows(0x5EED0001 + W, kind, 0, n, W)
    t = make_table(db)
    for qi in range(2):
        q = db[O.query_row(qi, n)]
        for k, cutoff in ((1000, 0.0), (10, 0.0), (100, 0.15), (1, 0.0)):
            check_against_oracle(t, db, q, k, cutoff, ctx="n=%d W=%d kind=%d q=%d k=%d c=%g" % (n, W, kind, qi, k, cutoff))
    fresh = O.synth_rows(0x5EED0002, kind, 999, 1, W)[0]
    check_against_oracle(t, db, fresh, 257, 0.0, ctx="fresh")
    t.close()


@pytest.mark.parametrize("n,W", [(300_001, 28), (200_003, 6), (150_000, 48), (100_001, 12), (64, 3), (65, 7),
                                 (40_000, 100), (9_001, 200), (3_000, 511), (1_000, 1023), (500_000, 2)])
def test_generic_widths_match_oracle(n, W):
    """Widths that are not a power-of-two number of 16-byte lanes take scan_generic_kernel (coalesced loads, rows
    transposed through LDS): even and odd row strides, rows wider than a 64-row LDS region holds (32 ... 4 rows per
    chunk), ragged last chunks, the Tversky metric."""
    db = O.synth_rows(0x6E6E + W, n % 2, 0, n, W)
    t = make_table(db)
    q = db[O.query_row(1, n)]
    for k, cutoff in ((1000, 0.0), (7, 0.0), (50, 0.2)):
        check_against_oracle(t, db, q, k, cutoff, ctx="generic n=%d W=%d k=%d c=%g" % (n, W, k, cutoff))
    check_against_oracle(t, db, O.synth_rows(0x5EED0003, 0, 5, 1, W)[0], 100, 0.0, ctx="generic tversky W=%d" % W,
                         metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
    t.close()


@pytest.mark.parametrize("n,W", [(300_001, 28), (150_003, 48), (100_001, 12), (250_000, 20), (120_007, 24), (90_001, 56), (70_000, 40),
                                 (30_001, 96), (9_001, 192), (5_003, 448), (63, 28), (64, 28), (65, 28), (1, 12), (17, 48), (2_000_001, 28),
                                 (200_001, 36), (150_000, 44), (100_003, 52), (90_001, 60), (60_000, 72), (20_001, 240), (65, 36)])
def test_register_streamed_odd_widths_match_oracle(n, W):
    """Rows of 3, 5, ... 15 (x 2^i) sixteen-byte units -- 384-, 640-, 896-, 768-, 1536-, 1792-, 1280-, 3072-, 6144-, 14336-bit,
    1152-, 1408-, 1664-, 1920-, 2304-, 7680-bit --
    stream through registers (scan_rows_ragged: prefix sums over the units of a chunk, a row's counts = the difference of
    two prefixes) -- inside the single launch for k <= 8192, in scan_ragged_kernel on the four-kernel pipeline (k = 9000
    here, and GSIM_FUSED=0 in test_gpu_fused.py).  Whole chunks, ragged last chunks, tables shorter than one chunk, dense
    and sparse rows, a cutoff, Tversky."""
    db = O.synth_rows(0x6A66 + W, n % 2, 0, n, W)
    t = make_table(db)
    q = db[O.query_row(1, n)]
    for k, cutoff in ((1000, 0.0), (7, 0.0), (50, 0.2), (9000, 0.0)):
        check_against_oracle(t, db, q, k, cutoff, ctx="ragged n=%d W=%d k=%d c=%g" % (n, W, k, cutoff))
    check_against_oracle(t, db, O.synth_rows(0x5EED0003, 0, 5, 1, W)[0], 100, 0.0, ctx="ragged tversky W=%d" % W,
                         metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
    qs = np.stack([db[O.query_row(i, n)] for i in range(5)])
    hits, approx = t.search(qs, 100, 0.0)  # (a batch on a width without a shared pass: five single queries)
    for i in range(5):
        want, wap = O.search(qs[i], db, 100, 0.0, nthreads=8)
        assert int(approx[i]) == wap
        assert_hits_equal(hits[i], want, "ragged batch W=%d q=%d" % (W, i))
    t.close()


@pytest.mark.parametrize("n,W", [(400_003, 5), (2_000_001, 5), (300_001, 6), (1_700_000, 6), (200_000, 3), (2_100_007, 3), (150_001, 7),
                                 (120_000, 10), (90_003, 14), (1_600_001, 14), (255, 5), (256, 5), (257, 5), (1, 3), (100, 7), (513, 6),
                                 (200_001, 9), (1_700_003, 9), (150_000, 11), (100_001, 18), (1_600_000, 22), (129, 22), (257, 9)])
def test_word_streamed_narrow_odd_widths_match_oracle(n, W):
    """Rows of 3, 5, 7, 9, 11 or twice that many 32-bit words (96- ... 704-bit: not whole 16-byte units) stream
    through registers at word granularity inside the single launch (scan_rows_wragged: a prefix sum over the words of a
    chunk through a per-wave LDS area); k = 9000 goes to the four-kernel pipeline and its LDS-staged scan.  Tables above
    1500 rows per wave are seeded by the sample first.  Whole chunks, partial last trips whose last 16-byte unit is cut by
    the end of the table, tables shorter than one chunk, a cutoff, Tversky."""
    db = O.synth_rows(0x7A66 + W, n % 2, 0, n, W)
    t = make_table(db)
    q = db[O.query_row(1, n)]
    for k, cutoff in ((1000, 0.0), (7, 0.0), (50, 0.2), (9000, 0.0)):
        check_against_oracle(t, db, q, k, cutoff, ctx="words n=%d W=%d k=%d c=%g" % (n, W, k, cutoff))
    check_against_oracle(t, db, O.synth_rows(0x5EED0003, 0, 5, 1, W)[0], 100, 0.0, ctx="words tversky W=%d" % W,
                         metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
    t.close()


def test_ragged_and_edge_sizes():
    W = 32
    for n in (1, 7, 63, 64, 65, 511, 513, 4095, 4097):
        db = O.synth_rows(0xE46E, 0, 0, n, W)
        t = make_table(db)
        for k in (0, 1, n, n + 5, 1000):
            check_against_oracle(t, db, db[n // 2], k, 0.0, ctx="n=%d k=%d" % (n, k))
            check_against_oracle(t, db, db[0], k, 0.05, ctx="n=%d k=%d cut" % (n, k))
        t.close()
    # empty table
    t = capi.Table(1024)
    t.add_rows(np.zeros((0, 32), np.uint32))
    t.finalize(0, 1)
    h, ap = t.search(np.ones(32, np.uint32), 10)
    assert len(h[0]) == 0 and int(ap[0]) == 0


def test_cutoff_semantics():
    db = O.synth_rows(0xC0701, 0, 0, 50_000, 32)
    t = make_table(db)
    q = db[123]
    for cutoff in (-1.0, 0.0, 1e-9, 0.05, 0.0999, 0.1, 0.5, 1.0, 1.0001, 5.0):
        check_against_oracle(t, db, q, 300, np.float32(cutoff), ctx="cutoff=%g" % cutoff)
    # all-zero query: every score is 0/0 = NaN or 0 -> 0 (fingerprintdb_cuda.cu:101)
    z = np.zeros(32, np.uint32)
    check_against_oracle(t, db, z, 50, 0.0, ctx="zero query")
    check_against_oracle(t, db, z, 50, 0.2, ctx="zero query cut")


def test_heavy_ties_force_radix_select():
    """> SELECT_CAP finalists share the k-th score: the in-kernel radix select must
    return the LOWEST row indices of the tie group (SURVEY App. C tie experiment)."""
    base = O.synth_rows(0x71E5, 0, 0, 3, 32)
    n = 40_000
    db = np.ascontiguousarray(np.tile(base, (n // 3 + 1, 1))[:n])
    t = make_table(db)
    for k in (5, 100, 1000, 8192):
        h = check_against_oracle(t, db, db[0], k, 0.0, ctx="ties k=%d" % k)
        assert (h["row"] == np.arange(k) * 3).all()
    # all rows identical AND equal to the query: one giant tie group
    same = np.ascontiguousarray(np.tile(base[:1], (30_000, 1)))
    t2 = make_table(same)
    h = check_against_oracle(t2, same, same[0], 777, 0.5, ctx="all identical")
    assert (h["row"] == np.arange(777)).all()


def test_large_k_path():
    db = O.synth_rows(0xB16, 0, 0, 120_000, 32)
    t = make_table(db)
    q = db[77]
    for k in (8193, 20_000, 120_000, 200_000):
        check_against_oracle(t, db, q, k, 0.0, ctx="large k=%d" % k)
    check_against_oracle(t, db, q, 50_000, 0.04, ctx="large k cutoff")


def test_large_k_with_ties_in_the_boundary_bin():
    """k > 8192 where the k-th score is shared by tens of thousands of rows: the one-workgroup route's boundary bin holds more
    keys than its LDS (its radix passes read global memory), the grid route histograms them; then ordinary tables right
    after, so that the route guessed from the previous query's finalist count is wrong in both directions."""
    base = O.synth_rows(0x71E7, 0, 0, 8, 32)
    n = 300_000
    tied = np.ascontiguousarray(base[np.random.default_rng(5).integers(0, 8, size=n)])  # ~37 k copies of each fingerprint
    t = make_table(tied)
    for k in (9_000, 20_000, 60_000, 100_000):
        check_against_oracle(t, tied, base[3], k, 0.0, ctx="tied large k=%d" % k)
    t.close()
    db = O.synth_rows(0xB17, 0, 0, 200_000, 32)
    t = make_table(db)
    t2 = make_table(tied)
    for i, k in enumerate((9_000, 50_000, 8_500, 150_000, 9_500)):  # few finalists, many, few, many ... (each table has its own hint)
        check_against_oracle(t, db, db[100 + i], k, 0.0, ctx="alternating k=%d" % k)
        check_against_oracle(t2, tied, base[i % 8], 10_000 + 7_000 * i, 0.0, ctx="alternating tied %d" % i)
    t.close()
    t2.close()


def test_large_k_single_launch_scan():
    """k in (2048, 32768] on a table of at least 64 k rows: the single launch scans and publishes, the hand-off kernel makes its
    lists the finalists of the large-k kernels (gsim_timing.large_k_single_scan counts the route).  Whole-table oracle at
    several widths and k, with a cutoff, with Tversky; a table of eight fingerprints (the k-th score is shared by 125 k rows); a table of ONE fingerprint, which the single
    launch hands back (a wave's store overflows: the four-kernel pipeline's scan answers, handed_back counts it); ordinary queries
    in between on the same handles."""
    for W, n, ks in ((32, 2_200_000, (2049, 4097, 8192, 8193, 20_000, 32_768)), (16, 1_400_000, (3_000, 6_000, 12_000, 20_000)), (64, 700_000, (10_000,))):
        db = O.synth_rows(0x5CA7 + W, 0, 0, n, W)
        t = make_table(db)
        before = t.timing()["large_k_single_scan"]
        for i, k in enumerate(ks):
            check_against_oracle(t, db, db[1000 + i], k, 0.0, ctx="single-scan large k W=%d k=%d" % (W, k))
        check_against_oracle(t, db, db[5], ks[0], 0.08, ctx="single-scan large k cutoff W=%d" % W)
        check_against_oracle(t, db, db[6], ks[0], 0.0, ctx="single-scan large k tversky W=%d" % W, metric=capi.METRIC_TVERSKY,
                             alpha=np.float32(0.3), beta=np.float32(0.7))
        tm = t.timing()
        assert tm["large_k_single_scan"] - before == len(ks) + 2, tm
        # (normally none; a wave scans ~2100 rows of the 2.2 M-row table and its store holds 2048: the very first launch on a fresh
        # handle, its code fetched cold, may elect its first threshold too late for one wave -- handed back, run again, exact)
        assert tm["handed_back"] <= 1 and not (tm["handed_back_why"] & ~1), tm
        check_against_oracle(t, db, db[7], 100, 0.0, ctx="small k after large W=%d" % W)  # (the state was left clean)
        check_against_oracle(t, db, db[8], 33_000, 0.0, ctx="k above the route's range W=%d" % W)
        t.close()
    base = O.synth_rows(0x71E8, 0, 0, 8, 32)
    n = 1_000_000
    tied = np.ascontiguousarray(base[np.random.default_rng(6).integers(0, 8, size=n)])  # 125 k copies of each fingerprint
    t = make_table(tied)
    for i, k in enumerate((9_000, 15_000)):
        check_against_oracle(t, tied, base[i], k, 0.0, ctx="single-scan tied k=%d" % k)
        check_against_oracle(t, tied, base[i + 2], 64, 0.0, ctx="small k on the tied table")
    tm = t.timing()
    # (125 k rows tie at the k-th score: the published lists hold them, but their score bin holds more rows than the bin-ranked
    # emission takes -- the first query is handed back for that (reason 64) and run again, the next ones take the radix tail)
    assert tm["large_k_single_scan"] == 2, tm
    if os.environ.get("GSIM_LARGEK_BINRANK", "1") != "0":
        assert tm["handed_back"] >= 1 and tm["handed_back_why"] & 64, tm
    t.close()
    same = np.ascontiguousarray(np.tile(base[:1], (2_400_000, 1)))  # every wave meets more tied rows than its store holds
    t = make_table(same)
    for k in (9_000, 30_000):
        h = check_against_oracle(t, same, base[0], k, 0.0, ctx="single-scan, all rows tie, k=%d" % k)
        assert (h["row"] == np.arange(k)).all()
        check_against_oracle(t, same, base[1], 500, 0.0, ctx="small k on the all-tie table")
    check_against_oracle(t, same, base[2], 12_000, 0.01, ctx="single-scan, all rows tie, cutoff")
    tm = t.timing()
    # (after a hand-back that was not the bin-ranked emission's the shard's next 2, 4, ... 64 large-k queries skip the publishing launch)
    assert 1 <= tm["large_k_single_scan"] <= 3 and tm["handed_back"] >= 3 and tm["handed_back_why"] & 1, tm
    # the pipelined per-query entry point: handed-back large-k queries with later ones already enqueued behind them
    qs = np.ascontiguousarray(np.stack([base[0], base[1], base[0], base[3], base[0]]))
    bufs = t.make_search_buffers(len(qs), 8500)
    t.search_each_into(qs, 8500, bufs)
    hits, counts, approx = bufs
    for i in range(len(qs)):
        want, wap = O.search(qs[i], same, 8500, 0.0, nthreads=8)
        assert int(approx[i]) == wap
        assert_hits_equal(hits[i][:int(counts[i])], want, "pipelined all-tie large k, query %d" % i)
    # enqueue-only callers (gsim_db_search_device: the RCCL route's shard step) cannot look at the block: the gated classic
    # kernels run behind the publishing launch -- the all-tie table -- or return at once -- an ordinary one
    import torch
    rnd = O.synth_rows(0x71EB, 0, 0, 1_200_000, 32)
    t2 = make_table(rnd)
    st = torch.cuda.Stream(device=0)
    for tab, db, q in ((t, same, base[0]), (t2, rnd, rnd[99])):
        tab.set_stream(st.cuda_stream)
        for k, cutoff in ((9_500, 0.0), (16_000, 0.02)):
            blk = capi.result_block_bytes(k)
            out = torch.zeros(blk, dtype=torch.uint8, device="cuda:0")
            with torch.cuda.stream(st):
                tab.search_device(q, k, out.data_ptr(), cutoff)
            st.synchronize()
            got, gap, _ = capi.parse_result_block(out.cpu().numpy().tobytes(), k)
            want, wap = O.search(q, db, k, cutoff, nthreads=8)
            assert gap == wap
            assert_hits_equal(got, want, "device block, large k=%d cutoff=%g" % (k, cutoff))
    assert t2.timing()["large_k_single_scan"] == 2 and t2.timing()["handed_back"] == 0
    t2.close()
    t.close()


def test_large_k_routes_forced():
    """GSIM_LARGEK_ONE_BLOCK_MAX (read once per process: child processes) forces the one-workgroup route for every finalist
    count and the grid route for every count: the large-k, tie and folded tests pass either way."""
    import subprocess
    import sys
    for var, val in (("GSIM_LARGEK_ONE_BLOCK_MAX", "0"), ("GSIM_LARGEK_ONE_BLOCK_MAX", "2000000000"), ("GSIM_LARGEK_BINRANK", "0")):
        env = dict(os.environ, **{var: val})
        if var == "GSIM_LARGEK_BINRANK":  # (the published rows through the radix tail for every caller; and through both of its routes)
            env["GSIM_LARGEK_ONE_BLOCK_MAX"] = "20000"
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-k",
                            "test_large_k_path or test_large_k_with_ties_in_the_boundary_bin or test_folded_search_matches or test_folded or "
                            "test_large_k_single_launch_scan"],
                           env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, var + "=" + val + ": " + r.stdout[-2000:] + r.stderr[-2000:]
        assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-500:]


def test_tversky_matches_oracle():
    for W, kind in ((64, 0), (32, 1)):
        db = O.synth_rows(0x7E25, kind, 0, 80_000, W)
        t = make_table(db)
        q = db[4242]
        for al, be in ((0.3, 0.7), (1.0, 1.0), (0.0, 1.0), (0.5, 0.5)):
            check_against_oracle(t, db, q, 500, 0.0, ctx="tversky %g %g" % (al, be), metric=capi.METRIC_TVERSKY,
                                 alpha=np.float32(al), beta=np.float32(be))
        check_against_oracle(t, db, q, 500, 0.3, ctx="tversky cutoff", metric=capi.METRIC_TVERSKY,
                             alpha=np.float32(0.3), beta=np.float32(0.7))
        # Tversky(1,1) == Tanimoto bit for bit
        a, _ = t.search(q, 500, 0.0)
        b, _ = t.search(q, 500, 0.0, metric=capi.METRIC_TVERSKY, alpha=1.0, beta=1.0)
        assert_hits_equal(a[0], b[0], "tversky(1,1)")


def test_multi_query_batch():
    db = O.synth_rows(0xBA7C4, 0, 0, 30_000, 32)
    t = make_table(db)
    qs = np.stack([db[O.query_row(i, 30_000)] for i in range(9)])
    hits, approx = t.search(qs, 64, 0.0)
    for i in range(9):
        want, wap = O.search(qs[i], db, 64, 0.0)
        assert_hits_equal(hits[i], want, "batch q=%d" % i)
        assert int(approx[i]) == wap


def test_generate_matches_oracle_generator():
    for kind, W in ((0, 32), (1, 32), (0, 64), (1, 5)):
        t = capi.Table(W * 32)
        t.generate(0x5EED0001, kind, 1_000, 20_000, 0)
        assert t.count() == 20_000
        for r in (0, 1, 63, 64, 19_999, 7_777):
            want = O.synth_rows(0x5EED0001, kind, 1_000 + r, 1, W)[0]
            assert (t.row(r) == want).all()
        db = O.synth_rows(0x5EED0001, kind, 1_000, 20_000, W)
        check_against_oracle(t, db, db[5], 100, 0.0, ctx="generated table")
        t.close()


def test_device_result_blocks_and_merge():
    """Shards as separate handles + gsim_merge_device == whole-table search
    (the multi-GPU data path, exercised on one GPU)."""
    import torch
    n, W, k, G = 100_000, 32, 1000, 4
    db = O.synth_rows(0x6A7E4, 0, 0, n, W)
    whole = make_table(db)
    q = db[31337]
    blk = capi.result_block_bytes(k)
    # a NON-default stream: the handle's kernels, the merge and torch's copies are
    # all ordered on it (a NULL stream would select the handle's own stream)
    st = torch.cuda.Stream(device=0)
    torch.cuda.set_stream(st)
    stream = st.cuda_stream
    assert stream != 0
    gathered = torch.zeros(G * blk, dtype=torch.uint8, device="cuda:0")
    shards = []
    per = n // G
    for g in range(G):
        t = make_table(db[g * per:(g + 1) * per])
        t.set_stream(stream)
        t.set_row_base(g * per)
        shards.append(t)
    for cutoff in (0.0, 0.08):
        for g, t in enumerate(shards):
            t.search_device(q, k, gathered.data_ptr() + g * blk, cutoff)
        out = torch.zeros(blk, dtype=torch.uint8, device="cuda:0")
        capi.merge_device(0, stream, gathered.data_ptr(), G, blk, k, out.data_ptr())
        torch.cuda.synchronize()
        hits, approx, flags = capi.parse_result_block(out.cpu().numpy().tobytes(), k)
        want, wap = O.search(q, db, k, cutoff, nthreads=8)
        assert approx == wap
        assert_hits_equal(hits, want, "merged cutoff=%g" % cutoff)
        # each shard's block alone equals the oracle on that shard
        for g in range(G):
            h, ap, _ = capi.parse_result_block(gathered[g * blk:(g + 1) * blk].cpu().numpy().tobytes(), k)
            w2, _ = O.search(q, db[g * per:(g + 1) * per], k, cutoff, row_base=g * per)
            assert_hits_equal(h, w2, "shard %d" % g)
    wh, _ = whole.search(q, k, 0.0)
    want, _ = O.search(q, db, k, 0.0, nthreads=8)
    assert_hits_equal(wh[0], want, "whole")


def test_attach_device_rows_borrowed_torch_tensor():
    import torch
    db = O.synth_rows(0xA77AC4, 0, 0, 40_000, 32)
    ten = torch.from_numpy(db.view(np.int32)).to("cuda:0")
    t = capi.Table(1024)
    t.attach_device_rows(ten.data_ptr(), 40_000, 0)
    check_against_oracle(t, db, db[9], 200, 0.0, ctx="attached")
    assert (t.row(39_999) == db[39_999]).all()


@pytest.mark.skipif(capi.device_count() < 2, reason="needs >= 2 GPUs in one process")
def test_in_process_multi_device_shards():
    db = O.synth_rows(0x5AAD, 0, 0, 300_001, 32)
    t = make_table(db, device=0, ndevices=0)
    assert t.shard_count() == capi.device_count()
    for qi in range(3):
        check_against_oracle(t, db, db[O.query_row(qi, len(db))], 1000, 0.0, ctx="multi-device")


# ---------------------------------------------------------------------------
# BASELINE sizes (configs[1], configs[2]): the WHOLE table against the oracle
# ---------------------------------------------------------------------------

def verify_hits_by_regeneration(hits, q, seed, kind, W, first_row=0):
    """Every returned row is regenerated on the CPU and rescored by the oracle."""
    a = int(np.unpackbits(q.view(np.uint8)).sum())
    L = O.lib()
    for h in hits:
        row = O.synth_rows(seed, kind, first_row + int(h["row"]), 1, W)[0]
        c = int(np.unpackbits((row & q).view(np.uint8)).sum())
        b = int(np.unpackbits(row.view(np.uint8)).sum())
        assert (c, b) == (int(h["common"]), int(h["popc_db"]))
        assert bits(np.float32(L.gso_score_one(0, 0, 0, a, b, c))) == bits(h["score"])


def canonical_sorted(hits):
    s, r = hits["score"], hits["row"].astype(np.int64)
    return bool(np.all((s[:-1] > s[1:]) | ((s[:-1] == s[1:]) & (r[:-1] < r[1:]))))


@pytest.mark.parametrize("nrows,kind", [(1_000_000, capi.SYNTH_SPARSE), (1_000_000, capi.SYNTH_MORGAN),
                                        (100_000_000, capi.SYNTH_SPARSE), (100_000_000, capi.SYNTH_MORGAN)])
def test_baseline_sizes_whole_table_against_oracle(nrows, kind):
    """BASELINE configs[1] / configs[2] (the sizes the metric is quoted on), i.i.d. and Morgan-shaped rows: the table is
    generated in HBM by the product and, row for row, on the host by the oracle's generator (all host threads); the
    GPU's full top-k -- rows, score bits, common, popc_db, approx -- is compared with the oracle's scan of ALL rows
    (fingerprintdb_cuda.cu:258-290 is the rule: every row scored, cutoff, sort, first k), for a self hit, a fresh
    fingerprint, k = 10, k = 1000, a selective and a dense cutoff and the k-th score as cutoff.  Then the
    size-independent properties on top: idempotence, prefix, pipelined == one at a time."""
    W, k, seed = 32, 1000, 0x5EED0001
    if capi.device_free_bytes(0) < nrows * 128 * 1.3:
        pytest.skip("not enough free HBM: %d rows x 1024-bit, kind %d NOT compared with the oracle on this box" % (nrows, kind))
    from conftest import record_parity
    record_parity("configs[%d] %d M x 1024-bit, kind %d" % (1 if nrows == 1_000_000 else 2, nrows // 1_000_000, kind), "oracle_all_rows")
    nt = os.cpu_count() or 1
    host = O.synth_rows_mt(seed, kind, 0, nrows, W, nt)  # 12.8 GB at 100 M rows
    t = capi.Table(1024)
    t.generate(seed, kind, 0, nrows, 0)
    qrow = O.query_row(0, nrows)
    q = t.row(qrow)
    assert (q == host[qrow]).all() and (t.row(nrows - 1) == host[nrows - 1]).all()
    fresh = O.synth_rows(0x5EED0002, kind, 77, 1, W)[0]  # not a row of the table
    cases = [(q, k, 0.0), (fresh, k, 0.0), (host[O.query_row(1, nrows)], 10, 0.0), (host[O.query_row(2, nrows)], k, 0.0),
             (q, k, 0.3), (fresh, 100, 0.12), (host[O.query_row(3, nrows)], 4000, 0.0)]
    results = []
    for i, (qq, kk, cut) in enumerate(cases):
        hits, approx = t.search(qq, kk, cut)
        want, wap = O.search(qq, host, kk, cut, nthreads=nt)
        assert int(approx[0]) == wap, "case %d" % i
        assert_hits_equal(hits[0], want, "case %d (%d rows, kind %d)" % (i, nrows, kind))
        results.append(hits[0])
    h = results[0]
    assert len(h) == k and h["score"][0] == np.float32(1.0) and qrow in h["row"][h["score"] == np.float32(1.0)]  # self hit
    assert canonical_sorted(h)
    # the k-th score as cutoff keeps exactly the rows scoring >= it (ties at the k-th score included)
    kth = float(h["score"][-1])
    h4, ap4 = t.search(q, 5000, kth)
    want4, wap4 = O.search(q, host, 5000, kth, nthreads=nt)
    assert int(ap4[0]) == wap4
    assert_hits_equal(h4[0], want4, "cutoff at the k-th score")
    # idempotence, prefix property, and the same queries through the pipelined entry point
    h2, _ = t.search(q, k, 0.0)
    assert_hits_equal(h2[0], h, "repeat")
    h3, _ = t.search(q, 100, 0.0)
    assert_hits_equal(h3[0], h[:100], "prefix")
    qs = np.stack([c[0] for c in cases[:4]] * 3)
    bufs = t.make_search_buffers(len(qs), k)
    t.search_each_into(qs, k, bufs)
    for i in range(len(qs)):
        kk = cases[i % 4][1]
        assert_hits_equal(bufs[0][i, :bufs[1][i]][:kk], results[i % 4][:kk], "pipelined %d" % i)
    assert t.timing()["blocks_torn"] == 0
    t.close()
    del host


@pytest.mark.parametrize("nrows,W", [(300_000_000, 5), (400_000_000, 4), (45_000_000, 36), (250_000_000, 6)])
def test_large_tables_of_other_widths_whole_table_against_oracle(nrows, W):
    """The widths that got their own streaming loops in round 4, at sizes whose byte offsets pass 4 GB: 160-bit and 192-bit rows
    (word-granular streaming), 128-bit rows (seeded by sample_rows_kernel), 1152-bit rows (nine sixteen-byte units through
    registers) -- generated in HBM by the product and row for row on the host by the oracle's generator, the GPU's full top-k
    compared with the oracle's scan of ALL rows: self hit, fresh fingerprint, small k, a cutoff, and k above 8192 (the
    four-kernel pipeline with its own scan for the width, radix select, two-launch sort)."""
    seed, kind = 0x5EED0077 + W, capi.SYNTH_SPARSE
    if capi.device_free_bytes(0) < nrows * W * 4 * 1.3:
        pytest.skip("not enough free HBM")
    nt = os.cpu_count() or 1
    host = O.synth_rows_mt(seed, kind, 0, nrows, W, nt)
    t = capi.Table(32 * W)
    t.generate(seed, kind, 0, nrows, 0)
    qrow = O.query_row(0, nrows)
    q = t.row(qrow)
    assert (q == host[qrow]).all() and (t.row(nrows - 1) == host[nrows - 1]).all()
    fresh = O.synth_rows(0x5EED0002, kind, 77, 1, W)[0]
    for i, (qq, kk, cut) in enumerate([(q, 1000, 0.0), (fresh, 1000, 0.0), (host[nrows - 3], 10, 0.0), (q, 500, 0.4), (host[O.query_row(2, nrows)], 9000, 0.0)]):
        hits, approx = t.search(qq, kk, cut)
        want, wap = O.search(qq, host, kk, cut, nthreads=nt)
        assert int(approx[0]) == wap, "case %d" % i
        assert_hits_equal(hits[0], want, "case %d (%d rows of %d words)" % (i, nrows, W))
    tm = t.timing()
    assert tm["blocks_torn"] == 0
    t.close()
    del host


@pytest.mark.parametrize("lg", [1, 2, 5, 10, 11, 12, 13, 15, 16, 18])
def test_device_sort_of_the_large_k_and_folded_paths(lg):
    """launch_sort_desc (tiles of 2048 keys sorted in LDS, then every key's position by counting over the other tiles): random
    unique 64-bit keys, zero padding of any length (equal keys: a permutation all the same), keys that differ in the low or
    the high word only, already sorted and reversed inputs -- against numpy's sort."""
    n = 1 << lg
    rng = np.random.default_rng(0x50B7 + lg)
    cases = []
    a = rng.integers(1, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    cases.append(a)
    b = a.copy()
    b[rng.integers(0, n, size=max(1, n // 3))] = 0  # padding anywhere, any amount
    cases.append(b)
    cases.append(np.zeros(n, dtype=np.uint64))
    cases.append((np.uint64(0x3F800000) << np.uint64(32)) | np.arange(n, dtype=np.uint64))  # one score, rows 0 .. n - 1
    cases.append(np.sort(a))
    cases.append(np.sort(a)[::-1].copy())
    cases.append(np.arange(n, dtype=np.uint64) << np.uint64(32))
    for i, keys in enumerate(cases):
        got = capi.debug_sort_desc(keys)
        want = np.sort(keys)[::-1]
        assert (got == want).all(), "n = 2^%d, case %d" % (lg, i)


def test_folded_search_host_rescore_route_and_nan_scores():
    """The re-score of a folded table's candidates runs on the device when the full fingerprints are in HBM as well;
    the host route (the reference's, fingerprintdb_cuda.cu:307-331) stays: forced here through GSIM_FOLD_RESCORE=host in a
    child process, and taken automatically when a re-scored value is NaN (two empty fingerprints: 0 / 0), for which only
    the literal bubble sort reproduces the reference's order."""
    import subprocess
    import sys
    env = dict(os.environ, GSIM_FOLD_RESCORE="host")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-k",
                        "test_folded_search_matches_reference_semantics"], env=env, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stdout.decode("utf-8", "replace")[-3000:]
    # NaN: empty rows in the table, an empty query
    n, W, F = 40_000, 32, 4
    db = O.synth_rows(0xF01E, 0, 0, n, W)
    db[::7] = 0
    t = capi.Table(W * 32).set_fold_factor(F)
    t.add_rows(db)
    t.finalize(0, 1)
    for q in (np.zeros(W, dtype=np.uint32), db[3], db[7]):
        for k, cutoff in ((20, 0.0), (200, 0.0)):
            hits, approx = t.search(q, k, cutoff)
            want, wap = O.search_folded(q, db, F, k, cutoff)
            assert int(approx[0]) == wap
            assert len(hits[0]) == len(want) and (hits[0]["row"] == want["row"]).all()
            assert (bits(hits[0]["score"]) == bits(want["score"])).all()
    t.close()


def test_sampled_threshold_and_adversarial_row_orders():
    """6 M rows: large enough for the sample kernel (starting threshold) to run.  The
    same rows in random, score-ascending (every row beats the running threshold: the
    filter's worst case, all rows become candidates) and score-descending order."""
    n, W, k = 6_000_000, 32, 1000
    db = O.synth_rows(0xAD7E, 0, 0, n, W)
    q = db[O.query_row(3, n)].copy()
    raw, _, _ = O.tanimoto_raw(q, db)
    orders = {"random": None, "ascending": np.argsort(raw, kind="stable"), "descending": np.argsort(-raw, kind="stable")}
    for name, perm in orders.items():
        tab = db if perm is None else np.ascontiguousarray(db[perm])
        t = make_table(tab)
        for kk, cutoff in ((k, 0.0), (10, 0.0), (k, 0.12)):
            check_against_oracle(t, tab, q, kk, cutoff, ctx="%s k=%d cutoff=%g" % (name, kk, cutoff))
        fresh = O.synth_rows(0x5EED0002, 0, 4242, 1, W)[0]
        check_against_oracle(t, tab, fresh, k, 0.0, ctx="%s fresh query" % name)
        t.close()


def test_folded_search_matches_reference_semantics():
    """copyToGPU(fold_factor > 1): approximate search on OR-folded fingerprints, re-scored
    with the full ones (fingerprintdb_cuda.cu:184-194, 284-331), one candidate list per
    storage, merged as FingerprintDB::search does (:363-380)."""
    n, W = 60_000, 32
    db = O.synth_rows(0xF01D, 0, 0, n, W)
    for requested in (2, 3, 4, 8):
        F = O.effective_fold_factor(W, requested)
        t = capi.Table(W * 32).set_fold_factor(requested)
        t.add_rows(db)
        t.finalize(0, 1)
        assert t.fold_factor() == F
        for qi, (k, cutoff) in enumerate(((20, 0.0), (100, 0.0), (50, 0.15), (10, 0.3))):
            q = db[O.query_row(qi, n)]
            hits, approx = t.search(q, k, cutoff)
            want, wap = O.search_folded(q, db, F, k, cutoff)
            assert int(approx[0]) == wap
            assert_hits_equal(hits[0], want, "fold %d k=%d cutoff=%g" % (F, k, cutoff))
        t.close()
    # three storages (three add_rows slices): per-storage candidate lists, then merge
    F, k = 4, 30
    t = capi.Table(W * 32).set_fold_factor(F)
    cuts = [0, 25_000, 31_000, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        t.add_rows(db[a:b])
    t.finalize(0, 1)
    assert t.shard_count() == 3
    q = db[4321]
    hits, approx = t.search(q, k, 0.0)
    parts, ap = [], 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        h, c = O.search_folded(q, db[a:b], F, k, 0.0, row_base=a)
        parts.append(h)
        ap += c
    want = O.merge_hits(parts, k)
    assert int(approx[0]) == ap
    assert_hits_equal(hits[0], want, "three storages")
    # the FingerprintDB twin
    fs = read_fsim(os.path.join(GOLD, "small.fsim"))
    fdb = FingerprintDB(fs.fp_bitcount, fs.fp_count, fs.dbkey, fs.fp_blocks, list(fs.smiles), list(fs.ids))
    fdb.copyToGPU(2, device=0)
    rows = fs.rows()
    h, ap = fdb.search_hits(rows[3], 10, 0.0)
    want, wap = O.search_folded(rows[3], rows, 2, 10, 0.0)
    assert ap == wap
    assert_hits_equal(h, want, "small.fsim folded")
    assert int(h["row"][0]) == 3 and h["score"][0] == 1.0


def batch_check(t, db, qs, k, cutoff=0.0, ctx="", **kw):
    hits, approx = t.search(qs, k, cutoff, **kw)
    okw = {}
    if "metric" in kw:
        okw = dict(metric=kw["metric"], alpha=kw.get("alpha", 1.0), beta=kw.get("beta", 1.0))
    for i in range(len(qs)):
        want, wap = O.search(qs[i], db, k, cutoff, nthreads=8, **okw)
        assert int(approx[i]) == wap, "%s q=%d" % (ctx, i)
        assert_hits_equal(hits[i], want, "%s q=%d" % (ctx, i))


@pytest.mark.parametrize("W,kind,n", [(64, 0, 150_000), (32, 0, 200_000), (32, 1, 120_000), (16, 0, 90_000),
                                      (8, 1, 50_000), (4, 0, 40_000)])
def test_multi_query_pass_matches_single_query_results(W, kind, n):
    """Multi-query passes, 70 queries: identical results to the oracle for every query.  128-bit rows
    and the cutoff case take the VALU pass (kBQ = 32 queries per table pass: 32 + 32 + 6), 256..2048-bit
    rows without a cutoff the matrix-core pass (the VALU pass on those:
    test_valu_pass_on_wide_rows_without_cutoff)."""
    db = O.synth_rows(0xBA7C0 + W, kind, 0, n, W)
    t = make_table(db)
    qs = np.stack([db[O.query_row(i, n)] for i in range(66)] +
                  [O.synth_rows(0x5EED0002, kind, 100 + i, 1, W)[0] for i in range(4)])
    batch_check(t, db, qs, 100, 0.0, ctx="batch W=%d" % W)
    batch_check(t, db, qs[:37], 1000, 0.0, ctx="batch k=1000 W=%d" % W)
    batch_check(t, db, qs[:9], 25, 0.11, ctx="batch cutoff W=%d" % W)
    batch_check(t, db, qs[:33], 300, 0.0, ctx="batch tversky W=%d" % W, metric=capi.METRIC_TVERSKY,
                alpha=np.float32(0.3), beta=np.float32(0.7))
    t.close()


def test_multi_query_pass_large_table_with_sampling():
    """3 M rows x 2048 bit: the batch sample kernel sets the starting thresholds."""
    n, W = 3_000_000, 64
    t = capi.Table(W * 32)
    t.generate(0x5EED0001, capi.SYNTH_SPARSE, 0, n, 0)
    db = O.synth_rows(0x5EED0001, 0, 0, n, W)
    qs = np.stack([db[O.query_row(i, n)] for i in range(40)])
    batch_check(t, db, qs, 1000, 0.0, ctx="3M x 2048-bit tversky", metric=capi.METRIC_TVERSKY,
                alpha=np.float32(0.3), beta=np.float32(0.7))
    t.close()


def test_multi_query_pass_fallbacks():
    """Heavy ties (> SELECT_CAP finalists for a query) and candidate-segment overflow: those
    queries are re-run through the single-query path; results stay exact."""
    base = O.synth_rows(0x71E5, 0, 0, 3, 32)
    n = 40_000
    db = np.ascontiguousarray(np.tile(base, (n // 3 + 1, 1))[:n])
    t = make_table(db)
    qs = np.stack([db[i % 3] for i in range(8)])
    batch_check(t, db, qs, 50, 0.0, ctx="batch heavy ties")
    t.close()
    os.environ["GSIM_BATCH_SEG_CAP"] = "256"  # force candidate-segment overflow
    try:
        db2 = O.synth_rows(0x0F10, 0, 0, 300_000, 32)
        t2 = make_table(db2)
        qs2 = np.stack([db2[O.query_row(i, len(db2))] for i in range(12)])
        batch_check(t2, db2, qs2, 200, 0.0, ctx="batch overflow fallback")
        t2.close()
    finally:
        del os.environ["GSIM_BATCH_SEG_CAP"]
    # the segments start small (4 Ki slots per wave instead of round 3's 64 Ki worst case: ~130 MB instead of ~2 GB on a
    # 256-CU part) and GROW when a wave asks for more: the batch runs again, nothing falls back, results stay exact
    os.environ["GSIM_BATCH_SEG_CAP_INIT"] = "16"
    try:
        for W in (32, 4):  # the matrix-core pass and the VALU pass
            db3 = O.synth_rows(0x0F11, 0, 0, 400_000, W)
            t3 = make_table(db3)
            qs3 = np.stack([db3[O.query_row(i, len(db3))] for i in range(40)])
            batch_check(t3, db3, qs3, 300, 0.0, ctx="batch segments regrown W=%d" % W)
            tm = t3.timing()
            assert tm["batches_regrown"] >= 1, tm
            batch_check(t3, db3, qs3, 300, 0.0, ctx="batch after regrowth W=%d" % W)
            assert t3.timing()["batches_regrown"] == tm["batches_regrown"]  # (large enough now)
            t3.close()
    finally:
        del os.environ["GSIM_BATCH_SEG_CAP_INIT"]


# ---------------------------------------------------------------------------
# multi-query pass on the matrix cores (gsim_batch_mfma.hip): cutoff <= 0 (with a cutoff: large tables),
# 1024- and 2048-bit rows
# ---------------------------------------------------------------------------

def _mixed_queries(db, kind, nq, W):
    n = len(db)
    own = [db[O.query_row(i, n)] for i in range(nq - nq // 4)]
    fresh = [O.synth_rows(0x5EED0002, kind, 500 + i, 1, W)[0] for i in range(nq // 4)]
    return np.stack(own + fresh)


@pytest.mark.parametrize("W,kind,n,nq", [(64, 0, 130_001, 256), (32, 0, 90_000, 97), (32, 1, 70_003, 64),
                                         (64, 1, 40_000, 130), (32, 0, 31, 64), (64, 0, 257, 65),
                                         (32, 0, 50_001, 200), (32, 1, 1_025, 256)])
def test_matrix_core_pass_matches_oracle(W, kind, n, nq):
    """Every query of a 64..256-query batch: rows, score bits, popcounts and approx identical to the
    oracle -- ragged table sizes (not a multiple of the 256/512-row LDS blocks, smaller than one
    tile), query counts that leave padding slots in the last 32-query tile; 1024-bit rows with more
    than 128 queries take the two-query-tiles-per-wave variant."""
    db = O.synth_rows(0x3FA4 + W + n, kind, 0, n, W)
    t = make_table(db)
    qs = _mixed_queries(db, kind, nq, W)
    batch_check(t, db, qs, 100, 0.0, ctx="mfma W=%d n=%d nq=%d" % (W, n, nq))
    batch_check(t, db, qs[:64], 1000, -1.0, ctx="mfma k=1000 W=%d n=%d" % (W, n))
    batch_check(t, db, qs, 7, 0.0, ctx="mfma tversky W=%d n=%d" % (W, n), metric=capi.METRIC_TVERSKY,
                alpha=np.float32(0.3), beta=np.float32(0.7))
    t.close()


def test_matrix_core_pass_borrowed_rows_changed_between_batches():
    """The matrix-core pass reads popc(row) from a side array made once per table; for rows the caller owns
    (gsim_db_attach_device_rows) it is recounted on every call: rewrite the tensor in place between two batches
    and both answers equal the oracle on the rows as they were at that call."""
    import torch
    n, W = 60_001, 32
    db1 = O.synth_rows(0xB0220, 0, 0, n, W)
    db2 = O.synth_rows(0xB0221, 1, 0, n, W)
    ten = torch.from_numpy(db1.view(np.int32).copy()).to("cuda:0")
    t = capi.Table(32 * W)
    t.attach_device_rows(ten.data_ptr(), n, 0)
    batch_check(t, db1, _mixed_queries(db1, 0, 96, W), 50, 0.0, ctx="borrowed, first")
    ten.copy_(torch.from_numpy(db2.view(np.int32).copy()))
    torch.cuda.synchronize()
    batch_check(t, db2, _mixed_queries(db2, 1, 96, W), 50, 0.0, ctx="borrowed, rewritten")
    t.close()


@pytest.mark.parametrize("W,n,nq", [(8, 300_007, 256), (16, 120_001, 130), (8, 2_049, 64)])
def test_matrix_core_pass_narrow_rows(W, n, nq):
    """256- and 512-bit rows on the matrix cores (2048 and 1024 rows per LDS block: the popcount side array's
    largest staging), ragged sizes."""
    db = O.synth_rows(0x3FA5 + W + n, 0, 0, n, W)
    t = make_table(db)
    qs = _mixed_queries(db, 0, nq, W)
    batch_check(t, db, qs, 100, 0.0, ctx="mfma narrow W=%d n=%d nq=%d" % (W, n, nq))
    batch_check(t, db, qs[:64], 20, 0.0, ctx="mfma narrow tversky W=%d" % W, metric=capi.METRIC_TVERSKY,
                alpha=np.float32(0.3), beta=np.float32(0.7))
    t.close()


def test_matrix_core_pass_tversky_weight_corners():
    """Weights for which the linear pre-filter is switched off or ill-conditioned (alpha = beta = 0,
    alpha + beta << 1) and asymmetric ones: the exact path decides, results equal the oracle."""
    n, W = 20_000, 32
    db = O.synth_rows(0xC0A7, 0, 0, n, W)
    t = make_table(db)
    qs = _mixed_queries(db, 0, 64, W)
    for al, be in [(0.0, 0.0), (0.01, 0.02), (1.0, 0.0), (0.0, 1.0), (2.0, 0.5), (1.0, 1.0)]:
        batch_check(t, db, qs, 50, 0.0, ctx="mfma tversky(%g,%g)" % (al, be), metric=capi.METRIC_TVERSKY,
                    alpha=np.float32(al), beta=np.float32(be))
    t.close()


def test_matrix_core_pass_ties_and_fallbacks():
    """Heavy ties (> SELECT_CAP finalists) and candidate-segment overflow on the matrix-core pass:
    flagged queries are re-run through the single-query path."""
    base = O.synth_rows(0x71E6, 0, 0, 5, 64)
    base[4] = 0
    n = 60_000
    rng = np.random.default_rng(5)
    db = np.ascontiguousarray(base[rng.integers(0, 5, size=n)])
    t = make_table(db)
    qs = np.stack([base[i % 5] for i in range(64)])
    batch_check(t, db, qs, 40, 0.0, ctx="mfma heavy ties")
    t.close()
    os.environ["GSIM_BATCH_SEG_CAP"] = "256"
    try:
        db2 = O.synth_rows(0x0F11, 0, 0, 200_000, 32)
        t2 = make_table(db2)
        qs2 = _mixed_queries(db2, 0, 64, 32)
        batch_check(t2, db2, qs2, 200, 0.0, ctx="mfma overflow fallback")
        t2.close()
    finally:
        del os.environ["GSIM_BATCH_SEG_CAP"]


def test_matrix_core_pass_large_table_with_sampling():
    """4 M rows x 1024 bit, 128 queries: sample passes set the starting thresholds, the contraction
    pass raises them while it streams; spot-checked against the oracle, every self hit present."""
    n, W, nq = 4_000_000, 32, 128
    t = capi.Table(W * 32)
    t.generate(0x5EED0001, capi.SYNTH_SPARSE, 0, n, 0)
    db = O.synth_rows(0x5EED0001, 0, 0, n, W)
    qs = np.stack([db[O.query_row(i, n)] for i in range(nq)])
    hits, approx = t.search(qs, 1000, 0.0)
    for i in range(nq):
        assert int(approx[i]) == n
        assert int(hits[i]["row"][0]) == O.query_row(i, n) and hits[i]["score"][0] == 1.0
    for i in (0, 31, 32, 77, 127):
        want, _ = O.search(qs[i], db, 1000, 0.0, nthreads=8)
        assert_hits_equal(hits[i], want, "4M x 1024-bit q=%d" % i)
    t.close()


def test_batch_device_blocks_and_batched_merge():
    """The multi-rank data path of a query batch on one GPU: shards as separate handles,
    gsim_db_search_batch_device per shard, rank-major gathered buffer, ONE
    gsim_merge_device_batch launch -- equal to the oracle on the whole table for every
    query; also the ShardedBatchSearch wrapper at world size 1, and the fallback of a
    chunk (heavy ties) to the single-query pipeline."""
    import torch
    from gpusimilarity_amd.sharded import ShardedBatchSearch
    n, W, k, G, nq = 90_000, 64, 100, 3, 70
    db = O.synth_rows(0x6A7E5, 0, 0, n, W)
    qs = _mixed_queries(db, 0, nq, W)
    blk = capi.result_block_bytes(k)
    st = torch.cuda.Stream(device=0)
    torch.cuda.set_stream(st)
    stream = st.cuda_stream
    per = n // G
    shards = []
    for g in range(G):
        t = make_table(db[g * per:(g + 1) * per])
        t.set_stream(stream)
        t.set_row_base(g * per)
        shards.append(t)
    kw = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
    gathered = torch.zeros(G * nq * blk, dtype=torch.uint8, device="cuda:0")
    for g, t in enumerate(shards):
        t.search_batch_device(qs, k, gathered.data_ptr() + g * nq * blk, 0.0, **kw)
    out = torch.zeros(nq * blk, dtype=torch.uint8, device="cuda:0")
    capi.merge_device_batch(0, stream, gathered.data_ptr(), G, nq, blk, k, out.data_ptr())
    torch.cuda.synchronize()
    raw = out.cpu().numpy().tobytes()
    for i in range(nq):
        hits, approx, _ = capi.parse_result_block(raw[i * blk:(i + 1) * blk], k)
        want, wap = O.search(qs[i], db, k, 0.0, nthreads=8, **kw)
        assert approx == wap
        assert_hits_equal(hits, want, "batched merge q=%d" % i)
    # wrapper, world size 1 (no process group): local blocks -> merge -> pinned host
    whole = make_table(db)
    sb = ShardedBatchSearch(whole, k, 128, "cuda:0", search_kwargs=kw)  # owns its stream, hands it to the table
    sb.enqueue(qs[:65])
    sb.synchronize()
    for i, (hits, approx, _) in enumerate(sb.results()):
        want, wap = O.search(qs[i], db, k, 0.0, nthreads=8, **kw)
        assert approx == wap
        assert_hits_equal(hits, want, "ShardedBatchSearch q=%d" % i)
    # heavy ties: the chunk is re-enqueued on the single-query pipeline, blocks stay exact
    base = O.synth_rows(0x71E7, 0, 0, 4, W)
    tied = np.ascontiguousarray(base[np.random.default_rng(9).integers(0, 4, size=50_000)])
    tt = make_table(tied)
    tt.set_stream(stream)
    q2 = np.stack([base[i % 4] for i in range(64)])
    buf = torch.zeros(64 * blk, dtype=torch.uint8, device="cuda:0")
    tt.search_batch_device(q2, k, buf.data_ptr(), 0.0)
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().tobytes()
    for i in (0, 1, 2, 3, 63):
        hits, approx, _ = capi.parse_result_block(raw[i * blk:(i + 1) * blk], k)
        want, wap = O.search(q2[i], tied, k, 0.0, nthreads=8)
        assert approx == wap
        assert_hits_equal(hits, want, "batch device fallback q=%d" % i)
    torch.cuda.set_stream(torch.cuda.default_stream(0))


def test_valu_pass_on_wide_rows_without_cutoff():
    """GSIM_BATCH_MFMA_MIN_Q=0 (read once per process, hence the child process) routes 1024/2048-bit
    batches without a cutoff through the VALU pass: it must stay exact too."""
    import subprocess
    import sys
    env = dict(os.environ, GSIM_BATCH_MFMA_MIN_Q="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        "test_multi_query_pass_matches_single_query_results and (64-0 or 32-1) or "
                        "test_multi_query_pass_large_table_with_sampling"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "3 passed" in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize("W,n", [(32, 20_000_000), (64, 17_000_000)])
def test_matrix_core_sample_pass_thresholds_are_safe(W, n):
    """Tables large enough for the matrix-core sample kernel (one launch sets the starting thresholds
    of all queries from ~1 M sampled rows): every query of the batch must return exactly what the
    single-query pipeline (itself pinned to the oracle) returns -- a threshold that is too high
    would lose hits."""
    t = capi.Table(W * 32)
    t.generate(0x5EED0001, capi.SYNTH_SPARSE, 0, n, 0)
    own = [O.synth_rows(0x5EED0001, 0, O.query_row(i, n), 1, W)[0] for i in range(70)]
    fresh = [O.synth_rows(0x5EED0002, 0, 900 + i, 1, W)[0] for i in range(10)]
    qs = np.stack(own + fresh)
    results = {}
    for kw in ({}, dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))):
        hits, approx = t.search(qs, 1000, 0.0, **kw)
        for i in range(len(qs)):
            one, ap1 = t.search(qs[i], 1000, 0.0, **kw)
            assert int(approx[i]) == int(ap1[0]) == n
            assert_hits_equal(hits[i], one[0], "W=%d q=%d %r" % (W, i, sorted(kw)))
        for i in range(70):
            assert int(hits[i]["row"][0]) == O.query_row(i, n) and hits[i]["score"][0] == 1.0
        results[tuple(sorted(kw))] = hits
    t.close()
    # ... and a subset of the batch directly against the ORACLE on the whole table (regenerated on the host)
    db = _host_table(0x5EED0001, n, W)
    for kw in ({}, dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))):
        hits = results[tuple(sorted(kw))]
        for i in (0, 13, 41, 69, 70, 79):
            want, wap = O.search(qs[i], db, 1000, 0.0, nthreads=ORACLE_THREADS, **kw)
            assert wap == n
            assert_hits_equal(hits[i], want, "oracle W=%d q=%d %r" % (W, i, sorted(kw)))
    # the single-query path with a cutoff on the same table: `approx` counts every row at or above it -- any chunk of
    # the table scanned twice or not at all would show
    t = capi.Table(W * 32)
    t.generate(0x5EED0001, capi.SYNTH_SPARSE, 0, n, 0)
    for i, (k, cutoff) in ((3, (1000, 0.06)), (75, (50, 0.11))):
        got, ap = t.search(qs[i], k, cutoff)
        want, wap = O.search(qs[i], db, k, cutoff, nthreads=ORACLE_THREADS)
        assert int(ap[0]) == wap, "approx W=%d q=%d cutoff=%g" % (W, i, cutoff)
        assert_hits_equal(got[0], want, "oracle, cutoff W=%d q=%d" % (W, i))
    t.close()


@pytest.mark.parametrize("W,n,nq,k", [(32, 50_000, 70, 8192), (64, 30_000, 130, 8192), (64, 300, 256, 300),
                                      (32, 5, 64, 10)])
def test_matrix_core_pass_large_k_and_tiny_tables(W, n, nq, k):
    """k at the multi-query select capacity (8192 finalists per query: queries with more ties fall
    back) and tables smaller than one row tile."""
    db = O.synth_rows(0xAD0C + W + n, 0, 0, n, W)
    t = make_table(db)
    qs = _mixed_queries(db, 0, nq, W)
    batch_check(t, db, qs, k, 0.0, ctx="mfma W=%d n=%d nq=%d k=%d" % (W, n, nq, k))
    batch_check(t, db, qs[:5], k, 0.0, ctx="mfma small batch W=%d n=%d k=%d" % (W, n, k))
    t.close()


@pytest.mark.parametrize("W,n", [(32, 20_000_000), (64, 17_000_000), (16, 24_000_000), (8, 30_000_000)])
def test_matrix_core_pass_with_cutoff(W, n):
    """Batches with a cutoff on tables large enough for the matrix-core sample pass: a selective
    cutoff stays on the matrix cores with the rows at or above it counted on the exact path; a cutoff
    that keeps a large part of the table (flagged by the sample pass) runs on the dense-cutoff variant,
    which counts the kept rows from the accumulators (gsim_prefilter.h cutoff_band) and sends only the
    pairs inside the band and the top-k candidates through the exact path; with the variant switched off
    the VALU pass takes those.  Every way hits AND approximate counts equal the single-query pipeline's."""
    t = capi.Table(W * 32)
    t.generate(0x5EED0001, capi.SYNTH_SPARSE, 0, n, 0)
    own = [O.synth_rows(0x5EED0001, 0, O.query_row(i, n), 1, W)[0] for i in range(40)]
    fresh = [O.synth_rows(0x5EED0002, 0, 700 + i, 1, W)[0] for i in range(8)]
    qs = np.stack(own + fresh)
    tv = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
    cases = [(0.2, {}), (0.12, tv), (0.02, {}), (0.04, tv), (0.07, {}), (1.5, {})]
    dense, sparse = {0.02, 0.04, 0.07}, {0.2, 1.5} # (0.12 Tversky: dense on the 1024-bit table, not on the 2048-bit one)
    results = []
    for cutoff, kw in cases:
        before = t.timing()["batches_dense_cutoff"]
        hits, approx = t.search(qs, 100, np.float32(cutoff), **kw)
        took_dense = t.timing()["batches_dense_cutoff"] > before
        assert (took_dense or cutoff not in dense) and not (took_dense and cutoff in sparse), "cutoff %g: route" % cutoff
        for i in range(len(qs)):
            one, ap1 = t.search(qs[i], 100, np.float32(cutoff), **kw)
            assert int(approx[i]) == int(ap1[0]), "W=%d cutoff=%g q=%d approx %d vs %d" % (W, cutoff, i, approx[i], ap1[0])
            assert_hits_equal(hits[i], one[0], "W=%d cutoff=%g q=%d" % (W, cutoff, i))
        results.append((hits, approx))
    # the same dense cutoff through the VALU pass (the route of weights without a band): a second handle on the same rows,
    # created with GSIM_BATCH_MFMA_DENSE=0 in the environment (the knobs are read per handle, by gsim_db_create)
    os.environ["GSIM_BATCH_MFMA_DENSE"] = "0"
    try:
        tv_off = capi.Table(W * 32)
    finally:
        del os.environ["GSIM_BATCH_MFMA_DENSE"]
    tv_off.generate(0x5EED0001, capi.SYNTH_SPARSE, 0, n, 0)
    before = tv_off.timing()["batches_dense_cutoff"]
    hits_v, approx_v = tv_off.search(qs, 100, np.float32(0.02))
    assert tv_off.timing()["batches_dense_cutoff"] == before, "the dense variant ran although it was switched off"
    tv_off.close()
    assert [int(x) for x in approx_v] == [int(x) for x in results[2][1]]
    for i in range(len(qs)):
        assert_hits_equal(hits_v[i], results[2][0][i], "VALU route q=%d" % i)
    t.close()
    # ... and a subset directly against the ORACLE on the whole table: hits and approximate counts
    db = _host_table(0x5EED0001, n, W)
    for (cutoff, kw), (hits, approx) in zip(cases, results):
        for i in (0, 17, 39, 40, 47):
            want, wap = O.search(qs[i], db, 100, np.float32(cutoff), nthreads=ORACLE_THREADS, **kw)
            assert int(approx[i]) == wap, "oracle W=%d cutoff=%g q=%d approx %d vs %d" % (W, cutoff, i, approx[i], wap)
            assert_hits_equal(hits[i], want, "oracle W=%d cutoff=%g q=%d" % (W, cutoff, i))


def test_matrix_core_dense_cutoff_full_batch():
    """256 queries (eight query tiles: every wave its own) through the dense-cutoff variant, 2048-bit rows, Tversky as
    BASELINE configs[4]: counts and hits equal the single-query pipeline's for every query, the oracle's for a few."""
    W, n = 64, 16_500_000
    t = capi.Table(W * 32)
    t.generate(0x5EED0003, capi.SYNTH_SPARSE, 0, n, 0)
    qs = np.stack([O.synth_rows(0x5EED0003, 0, O.query_row(i, n), 1, W)[0] for i in range(256)])
    kw = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
    cutoff = np.float32(0.06)
    before = t.timing()["batches_dense_cutoff"]
    hits, approx = t.search(qs, 50, cutoff, **kw)
    assert t.timing()["batches_dense_cutoff"] > before
    assert int(np.min(approx)) > n // 512  # (dense indeed)
    for i in range(256):
        one, ap1 = t.search(qs[i], 50, cutoff, **kw)
        assert int(approx[i]) == int(ap1[0]), "q=%d approx %d vs %d" % (i, approx[i], ap1[0])
        assert_hits_equal(hits[i], one[0], "q=%d" % i)
    t.close()
    db = _host_table(0x5EED0003, n, W)
    for i in (0, 100, 255):
        want, wap = O.search(qs[i], db, 50, cutoff, nthreads=ORACLE_THREADS, **kw)
        assert int(approx[i]) == wap
        assert_hits_equal(hits[i], want, "oracle q=%d" % i)


def test_bench_contract_lines():
    """bench.py prints exactly one JSON line with the contract's keys, in the headline mode (through
    the sharded code path as well) and in the batch mode (BASELINE configs[4] shape, small)."""
    import subprocess
    import sys
    common = ["--steps", "3", "--warmup", "2", "--rows-per-gpu", "400000", "--no-configs", "--queries-per-step", "4"]
    for extra in (["--no-cpu-baseline"], ["--no-cpu-baseline", "--force-sharded-path"],
                  ["--fp-bits", "2048", "--batch-queries", "64"],
                  ["--fp-bits", "2048", "--batch-queries", "64", "--force-sharded-path"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + extra, cwd=ROOT,
                           capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541"))
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout[-500:]
        assert len(lines[0]) < 12000, len(lines[0])
        d = json.loads(lines[0])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                    "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "detail"):
            assert key in d, key
        assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and "workload" in d["config"]
        assert d["roofline"]["bound"] in ("hbm", "mfma") and d["roofline"]["peak"] > 0
        assert d["collective"]["world"] == 1 and d["collective"]["ranks"][0]["rank"] == 0
        if "--batch-queries" in extra:
            continue
        if "--force-sharded-path" in extra:
            # the N > 1 line decomposes: this rank's phase times by HIP events, its shard's single-GPU twin
            pr = d["collective"]["per_rank"][0]
            assert set(pr) >= {"search_ms", "gather_us", "merge_us", "d2h_us"} and pr["search_ms"] > 0
            assert pr["twin_ms_per_query"] > 0 and d["collective"]["queries_in_flight"] == 8
            # ... and the full per-rank record is in the detail file the line names
            full = json.load(open(os.path.join(ROOT, d["detail"])))
            assert full["collective"]["per_rank"][0]["single_gpu_twin"]["ms_per_query"] > 0
        else:
            # SURVEY 8(d)'s latency next to the pipelined mean, and the PMC traffic of the dominant kernel in the line itself
            assert d["sync_ms_median"] > 0 and d["sync_ms_p95"] >= d["sync_ms_median"] and d["calls"] >= 50
            import shutil
            if shutil.which("rocprofv3"):
                rf = d["roofline"]
                assert rf["traffic"] is not None, rf["traffic_note"]
                # (400 k rows = 51 MB: the table fits the 256 MB Infinity Cache, whose hits the counter includes; the bound is loose)
                assert 0.5 * rf["algorithmic_bytes_per_launch"] <= rf["traffic"] <= 1.5 * rf["algorithmic_bytes_per_launch"], rf


def test_bench_default_command_line_parses():
    """The line of the command the DRIVER runs (`bench.py --gpus 1 --steps 20 --warmup 5`: every config, the widths, the
    server record, the CPU baseline) is one JSON line under 12 KB that carries `roofline` and `cpu_baseline`; everything
    else is in the detail file it names (round 5's 20.5 KB line could not be parsed by the driver)."""
    import subprocess
    import sys
    from gpusimilarity_amd import capi
    if capi.device_free_bytes(0) < 140 * 2**30:
        pytest.skip("the default run holds the 1 B-row table (128 GB): not enough free HBM on this device")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"], cwd=ROOT,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-500:]
    assert len(lines[0]) < 12000, len(lines[0])
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "summary", "detail"):
        assert key in d, key
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and "workload" in d["config"]
    rf, cb = d["roofline"], d["cpu_baseline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0.5 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port") and cb["sample"]
    # kernel <= step: the dominant kernel's average duration cannot exceed the mean time per query of the timed region
    assert rf["kernel_ms_avg"] <= d["ms_per_query"] * 1.001
    assert d["timed_region_s"] > 0 and abs(d["ms_per_step"] * d["steps"] / 1e3 - d["timed_region_s"]) < 0.01 * d["timed_region_s"]
    full = json.load(open(os.path.join(ROOT, d["detail"])))
    assert len(full["configs"]) >= 8 and full["widths"] and full["cpu_baseline"]["parts"]
    assert len(d["summary"]["configs"]) == len(full["configs"])
