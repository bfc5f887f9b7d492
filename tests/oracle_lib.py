"""ctypes binding of oracle/libgsim_oracle.so (+ optional oracle/_ref).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

METRIC_TANIMOTO = 0
METRIC_TVERSKY = 1
KIND_SPARSE = 0
KIND_DENSE = 1
KIND_MORGAN = 2

HIT_DTYPE = np.dtype([("row", "<u4"), ("score", "<f4"), ("common", "<u2"), ("popc_db", "<u2")])

_u32p = C.POINTER(C.c_uint32)


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "libgsim_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        L.gso_splitmix64.restype = C.c_uint64
        L.gso_splitmix64.argtypes = [C.c_uint64]
        L.gso_synth_word.restype = C.c_uint32
        L.gso_synth_word.argtypes = [C.c_uint64, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32]
        L.gso_synth_rows.restype = None
        L.gso_synth_rows.argtypes = [_u32p, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32]
        L.gso_query_row.restype = C.c_uint64
        L.gso_query_row.argtypes = [C.c_uint64, C.c_uint64]
        L.gso_tanimoto_raw.restype = None
        L.gso_tanimoto_raw.argtypes = [_u32p, _u32p, C.c_uint64, C.c_uint32, C.POINTER(C.c_float),
                                       C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]
        L.gso_score_one.restype = C.c_float
        L.gso_score_one.argtypes = [C.c_int, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32]
        L.gso_apply_cutoff.restype = C.c_float
        L.gso_apply_cutoff.argtypes = [C.c_float, C.c_float]
        L.gso_search.restype = C.c_int
        L.gso_search.argtypes = [_u32p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_float, C.c_int,
                                 C.c_float, C.c_float, C.c_uint32, C.c_int, C.c_void_p, _u32p,
                                 C.POINTER(C.c_uint64)]
        L.gso_merge_hits.restype = None
        L.gso_merge_hits.argtypes = [C.c_void_p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, _u32p]
        L.gso_bubble_sort.restype = None
        L.gso_bubble_sort.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int, C.c_int]
        L.gso_search_cpu.restype = C.c_int
        L.gso_search_cpu.argtypes = [_u32p, _u32p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_int),
                                     C.POINTER(C.c_float)]
        L.gso_fold.restype = None
        L.gso_fold.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.gso_fold_rows.restype = None
        L.gso_fold_rows.argtypes = [_u32p, C.c_uint64, C.c_uint32, C.c_int, _u32p]
        L.gso_effective_fold_factor.restype = C.c_int
        L.gso_effective_fold_factor.argtypes = [C.c_uint32, C.c_int]
        L.gso_search_folded.restype = C.c_int
        L.gso_search_folded.argtypes = [_u32p, _u32p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.c_float,
                                        C.c_uint32, C.c_void_p, _u32p, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def fold_rows(db, factor):
    db = np.ascontiguousarray(db, dtype=np.uint32)
    n, W = db.shape
    out = np.zeros((n, W // factor), dtype=np.uint32)
    lib().gso_fold_rows(_ptr(db, C.c_uint32), n, W, factor, _ptr(out, C.c_uint32))
    return out


def effective_fold_factor(W, requested):
    return int(lib().gso_effective_fold_factor(W, requested))


def search_folded(query, db, fold_factor, k, cutoff=0.0, row_base=0):
    db = np.ascontiguousarray(db, dtype=np.uint32)
    query = np.ascontiguousarray(query, dtype=np.uint32)
    n, W = db.shape
    hits = np.zeros(max(1, k), dtype=HIT_DTYPE)
    nh = C.c_uint32(0)
    ap = C.c_uint64(0)
    rc = lib().gso_search_folded(_ptr(query, C.c_uint32), _ptr(db, C.c_uint32), n, W, fold_factor, k, cutoff, row_base,
                                 hits.ctypes.data_as(C.c_void_p), C.byref(nh), C.byref(ap))
    assert rc == 0
    return hits[:nh.value].copy(), int(ap.value)


def synth_rows(seed, kind, first_row, nrows, W):
    out = np.empty((nrows, W), dtype=np.uint32)
    lib().gso_synth_rows(_ptr(out, C.c_uint32), seed, kind, first_row, nrows, W)
    return out


def synth_rows_mt(seed, kind, first_row, nrows, W, nthreads=None):
    """synth_rows on several host threads (ctypes releases the GIL): the tables of the full-size parity tests are
    10^8 .. 10^9 rows; every thread first-touches the pages of its own slice."""
    from concurrent.futures import ThreadPoolExecutor
    nthreads = max(1, min(nthreads or (os.cpu_count() or 1), 256, (nrows + 65535) // 65536))
    out = np.empty((nrows, W), dtype=np.uint32)
    per = (nrows + nthreads - 1) // nthreads
    L = lib()

    def work(t):
        lo, hi = t * per, min(nrows, (t + 1) * per)
        if lo < hi:
            L.gso_synth_rows(_ptr(out[lo:hi], C.c_uint32), seed, kind, first_row + lo, hi - lo, W)

    with ThreadPoolExecutor(nthreads) as ex:
        list(ex.map(work, range(nthreads)))
    return out


def query_row(q, nrows):
    return int(lib().gso_query_row(q, nrows))


def tanimoto_raw(query, db):
    db = np.ascontiguousarray(db, dtype=np.uint32)
    query = np.ascontiguousarray(query, dtype=np.uint32)
    n, W = db.shape
    sc = np.empty(n, dtype=np.float32)
    cm = np.empty(n, dtype=np.uint16)
    pc = np.empty(n, dtype=np.uint16)
    lib().gso_tanimoto_raw(_ptr(query, C.c_uint32), _ptr(db, C.c_uint32), n, W, _ptr(sc, C.c_float),
                           _ptr(cm, C.c_uint16), _ptr(pc, C.c_uint16))
    return sc, cm, pc


def search(query, db, k, cutoff=0.0, metric=METRIC_TANIMOTO, alpha=1.0, beta=1.0, row_base=0, nthreads=1):
    """Canonical search -> (hits structured array, approx)."""
    db = np.ascontiguousarray(db, dtype=np.uint32)
    query = np.ascontiguousarray(query, dtype=np.uint32)
    n, W = db.shape
    hits = np.zeros(max(1, min(k, n)), dtype=HIT_DTYPE)
    nh = C.c_uint32(0)
    ap = C.c_uint64(0)
    rc = lib().gso_search(_ptr(query, C.c_uint32), _ptr(db, C.c_uint32), n, W, k, cutoff, metric, alpha, beta,
                          row_base, nthreads, hits.ctypes.data_as(C.c_void_p), C.byref(nh), C.byref(ap))
    assert rc == 0
    return hits[:nh.value].copy(), int(ap.value)


def merge_hits(lists, k):
    """lists: list of HIT_DTYPE arrays (each canonical order)."""
    stride = max([len(x) for x in lists] + [1])
    buf = np.zeros((len(lists), stride), dtype=HIT_DTYPE)
    counts = np.zeros(len(lists), dtype=np.uint32)
    for i, x in enumerate(lists):
        buf[i, :len(x)] = x
        counts[i] = len(x)
    out = np.zeros(max(1, k), dtype=HIT_DTYPE)
    nout = C.c_uint32(0)
    lib().gso_merge_hits(buf.ctypes.data_as(C.c_void_p), _ptr(counts, C.c_uint32), len(lists), stride, k,
                         out.ctypes.data_as(C.c_void_p), C.byref(nout))
    return out[:nout.value].copy()


def bubble_sort(indices, scores, number_required):
    idx = np.ascontiguousarray(indices, dtype=np.int32).copy()
    sc = np.ascontiguousarray(scores, dtype=np.float32).copy()
    lib().gso_bubble_sort(_ptr(idx, C.c_int), _ptr(sc, C.c_float), len(idx), number_required)
    return idx, sc


def search_cpu(query, db, k):
    db = np.ascontiguousarray(db, dtype=np.uint32)
    query = np.ascontiguousarray(query, dtype=np.uint32)
    n, W = db.shape
    rows = np.zeros(k, dtype=np.int32)
    sc = np.zeros(k, dtype=np.float32)
    rc = lib().gso_search_cpu(_ptr(query, C.c_uint32), _ptr(db, C.c_uint32), n, W, k, _ptr(rows, C.c_int),
                              _ptr(sc, C.c_float))
    assert rc == 0
    return rows, sc


def fold(fp, factor):
    fp = np.ascontiguousarray(fp, dtype=np.int32)
    out = np.zeros(len(fp) // factor, dtype=np.int32)
    lib().gso_fold(_ptr(fp, C.c_int), len(fp), factor, _ptr(out, C.c_int))
    return out


# ---- optional: the reference's own functors (oracle/_ref) -------------------
_ref = None
_ref_sort = None


def ref_lib():
    """oracle/_ref/libgsim_ref.so or None when it has not been built."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libgsim_ref.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.gsref_table_create.restype = C.c_void_p
        L.gsref_table_create.argtypes = [C.POINTER(C.c_int), C.c_uint64, C.c_int]
        L.gsref_table_destroy.restype = None
        L.gsref_table_destroy.argtypes = [C.c_void_p]
        L.gsref_tanimoto_scan.restype = None
        L.gsref_tanimoto_scan.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_uint64, C.c_int]
        if hasattr(L, "gsref_search_topk"):
            L.gsref_search_topk.restype = C.c_int
            L.gsref_search_topk.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_uint64, C.c_int, C.c_int,
                                            C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L.gsref_fold.restype = None
        L.gsref_fold.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_int)]
        _ref = L
    return _ref


class RefTable:
    """A table held by the reference's TanimotoFunctorCPU driver."""

    def __init__(self, db):
        db = np.ascontiguousarray(db, dtype=np.uint32)
        self.n, self.W = db.shape
        self.h = ref_lib().gsref_table_create(db.view(np.int32).ctypes.data_as(C.POINTER(C.c_int)), self.n, self.W)

    def scan(self, query, nthreads=1):
        q = np.ascontiguousarray(query, dtype=np.uint32).view(np.int32)
        out = np.empty(self.n, dtype=np.float32)
        ref_lib().gsref_tanimoto_scan(self.h, q.ctypes.data_as(C.POINTER(C.c_int)), _ptr(out, C.c_float), self.n,
                                      nthreads)
        return out

    def search_topk(self, query, k, nthreads=1):
        """The reference functor on nthreads threads + a canonical top-k selection -> (rows, scores)."""
        q = np.ascontiguousarray(query, dtype=np.uint32).view(np.int32)
        rows = np.empty(max(k, 1), dtype=np.int32)
        scores = np.empty(max(k, 1), dtype=np.float32)
        n = ref_lib().gsref_search_topk(self.h, q.ctypes.data_as(C.POINTER(C.c_int)), self.n, nthreads, k,
                                        _ptr(rows, C.c_int), _ptr(scores, C.c_float))
        return rows[:n], scores[:n]

    def close(self):
        if self.h:
            ref_lib().gsref_table_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def ref_fold(fp, factor):
    fp = np.ascontiguousarray(fp, dtype=np.int32)
    out = np.zeros(len(fp) // factor, dtype=np.int32)
    ref_lib().gsref_fold(_ptr(fp, C.c_int), len(fp), factor, _ptr(out, C.c_int))
    return out


def ref_sort_lib():
    global _ref_sort
    if _ref_sort is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libgsim_ref_sort.so")
        if not os.path.exists(path):
            return None
        try:
            _ref_sort = C.CDLL(path)
        except OSError:
            return None
    return _ref_sort


def canonical_topk_from_scores(scores, k, cutoff):
    """Reference semantics applied to a raw score vector with numpy (used to turn
    reference-functor scores into golden top-k vectors)."""
    s = np.asarray(scores, dtype=np.float32).copy()
    with np.errstate(invalid="ignore"):
        s = np.where(s >= np.float32(cutoff), s, np.float32(0.0)).astype(np.float32)
    rows = np.arange(len(s), dtype=np.int64)
    if cutoff > 0:
        keep = s != 0
        rows, s = rows[keep], s[keep]
    order = np.lexsort((rows, -s.astype(np.float64)))
    order = order[:k]
    return rows[order], s[order], int(len(rows))
