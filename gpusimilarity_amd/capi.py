"""ctypes binding of the C ABI in ``include/gpusim_hip.h`` (libgsim_hip.so).

This is the only way Python reaches the GPU in this package: there is no
PyTorch/numpy fallback.  If the library is missing or no GPU is usable the calls
fail loudly (``GsimError``).
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSIM_LIB: alternative build of the same ABI (kernel ablation experiments only)
LIB_PATH = os.environ.get("GSIM_LIB") or os.path.join(_HERE, "libgsim_hip.so")

OK = 0
METRIC_TANIMOTO = 0
METRIC_TVERSKY = 1
SYNTH_SPARSE = 0
SYNTH_DENSE = 1
SYNTH_MORGAN = 2
SELECT_CAP = 8192

HIT_DTYPE = np.dtype([("row", "<u4"), ("score", "<f4"), ("common", "<u2"), ("popc_db", "<u2")])
HEADER_DTYPE = np.dtype([("count", "<u4"), ("flags", "<u4"), ("approx", "<u8")])


class GsimTiming(C.Structure):
    _fields_ = [("queries", C.c_uint64), ("scan_ms_sum", C.c_double), ("select_ms_sum", C.c_double),
                ("candidates_sum", C.c_uint64), ("finalists_sum", C.c_uint64), ("handed_back", C.c_uint64),
                ("batches", C.c_uint64), ("batch_kernel_ms_sum", C.c_double), ("handed_back_why", C.c_uint64), ("batches_dense_cutoff", C.c_uint64),
                ("collectives", C.c_uint64), ("gather_ms_sum", C.c_double), ("merge_ms_sum", C.c_double),
                ("blocks_rechecked", C.c_uint64), ("blocks_torn", C.c_uint64), ("batches_regrown", C.c_uint64),
                ("large_k_single_scan", C.c_uint64), ("rerun_own", C.c_uint64), ("rerun_publish", C.c_uint64),
                ("rerun_behind", C.c_uint64), ("rerun_torn", C.c_uint64), ("lane_queries", C.c_uint64), ("backoff_skips", C.c_uint64)]


class GsimError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gsim error %d: %s" % (code, msg))
        self.code = code


_lib = None

EXPORTS = [
    "gsim_device_count", "gsim_device_free_bytes", "gsim_available_device_bytes", "gsim_next_device",
    "gsim_db_create", "gsim_db_add_rows", "gsim_db_finalize", "gsim_db_set_fold_factor", "gsim_db_fold_factor", "gsim_db_set_fold_full_on_device",
    "gsim_fold_fingerprint", "gsim_db_generate", "gsim_db_generate_sharded", "gsim_synth_row", "gsim_db_attach_device_rows",
    "gsim_db_destroy", "gsim_db_count", "gsim_db_fp_bits", "gsim_db_data_bytes", "gsim_db_row",
    "gsim_db_shard_count", "gsim_db_shard_device", "gsim_db_search", "gsim_db_search_each", "gsim_db_search_timed", "gsim_db_search_cpu", "gsim_db_set_stream", "gsim_db_set_row_base",
    "gsim_result_block_bytes", "gsim_db_search_device", "gsim_merge_device", "gsim_db_search_batch_device",
    "gsim_merge_device_batch", "gsim_merge_host",
    "gsim_comm_create", "gsim_comm_destroy", "gsim_comm_size", "gsim_rccl_info", "gsim_db_set_comm", "gsim_db_set_comm_root",
    "gsim_db_enable_timing",
    "gsim_db_get_timing", "gsim_debug_query_flags", "gsim_debug_litmus", "gsim_debug_score_table", "gsim_debug_prefilter_constants", "gsim_debug_sort_desc", "gsim_last_error", "gsim_version",
]


def load():
    """Load libgsim_hip.so.  PyTorch (when importable) is imported FIRST so that the
    process holds exactly one HIP runtime: torch ships its own libamdhip64.so.7 and
    the dynamic linker then resolves our NEEDED entry to that same object."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GsimError(-100, "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C gpusimilarity_amd/csrc`" % LIB_PATH)
    if os.environ.get("GSIM_NO_TORCH", "") != "1" and "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:  # torch is plumbing, not a requirement of the ABI
            pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    u32p, u64p, vp = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_void_p
    sig = {
        "gsim_device_count": (C.c_int, [C.POINTER(C.c_int)]),
        "gsim_device_free_bytes": (C.c_int, [C.c_int, C.POINTER(C.c_size_t)]),
        "gsim_available_device_bytes": (C.c_int, [C.POINTER(C.c_size_t)]),
        "gsim_next_device": (C.c_int, [C.c_size_t, C.POINTER(C.c_int)]),
        "gsim_db_create": (C.c_int, [C.c_uint32, C.POINTER(vp)]),
        "gsim_db_add_rows": (C.c_int, [vp, u32p, C.c_uint64]),
        "gsim_db_finalize": (C.c_int, [vp, C.c_int, C.c_int]),
        "gsim_db_set_fold_factor": (C.c_int, [vp, C.c_uint32]),
        "gsim_db_fold_factor": (C.c_uint32, [vp]),
        "gsim_db_set_fold_full_on_device": (C.c_int, [vp, C.c_int]),
        "gsim_fold_fingerprint": (C.c_int, [u32p, C.c_uint32, C.c_uint32, u32p]),
        "gsim_db_generate": (C.c_int, [vp, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_int]),
        "gsim_db_generate_sharded": (C.c_int, [vp, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int]),
        "gsim_synth_row": (C.c_int, [C.c_uint64, C.c_int, C.c_uint64, C.c_uint32, u32p]),
        "gsim_db_attach_device_rows": (C.c_int, [vp, vp, C.c_uint64, C.c_int]),
        "gsim_db_destroy": (C.c_int, [vp]),
        "gsim_db_count": (C.c_uint64, [vp]),
        "gsim_db_fp_bits": (C.c_uint32, [vp]),
        "gsim_db_data_bytes": (C.c_size_t, [vp]),
        "gsim_db_row": (C.c_int, [vp, C.c_uint64, u32p]),
        "gsim_db_shard_count": (C.c_int, [vp]),
        "gsim_db_shard_device": (C.c_int, [vp, C.c_int]),
        "gsim_db_search": (C.c_int, [vp, u32p, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.c_float, C.c_float,
                                     vp, u32p, u64p]),
        "gsim_db_search_each": (C.c_int, [vp, u32p, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.c_float, C.c_float,
                                          vp, u32p, u64p]),
        "gsim_db_search_timed": (C.c_int, [vp, u32p, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.c_float, C.c_float,
                                           vp, u32p, u64p, C.POINTER(C.c_double)]),
        "gsim_db_search_cpu": (C.c_int, [vp, u32p, C.c_uint32, C.c_uint32, C.c_float, vp, u32p]),
        "gsim_db_set_stream": (C.c_int, [vp, vp]),
        "gsim_db_set_row_base": (C.c_int, [vp, C.c_uint32]),
        "gsim_result_block_bytes": (C.c_size_t, [C.c_uint32]),
        "gsim_db_search_device": (C.c_int, [vp, u32p, C.c_uint32, C.c_float, C.c_int, C.c_float, C.c_float, vp]),
        "gsim_merge_device": (C.c_int, [C.c_int, vp, vp, C.c_uint32, C.c_size_t, C.c_uint32, vp]),
        "gsim_db_search_batch_device": (C.c_int, [vp, u32p, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.c_float,
                                                  C.c_float, vp]),
        "gsim_merge_device_batch": (C.c_int, [C.c_int, vp, vp, C.c_uint32, C.c_uint32, C.c_size_t, C.c_uint32, vp]),
        "gsim_merge_host": (C.c_int, [vp, C.c_uint32, C.c_size_t, C.c_uint32, vp]),
        "gsim_comm_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]),
        "gsim_comm_destroy": (C.c_int, [vp]),
        "gsim_rccl_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
        "gsim_comm_size": (C.c_int, [vp]),
        "gsim_db_set_comm": (C.c_int, [vp, vp]),
        "gsim_db_set_comm_root": (C.c_int, [vp, C.c_int]),
        "gsim_db_enable_timing": (C.c_int, [vp, C.c_int]),
        "gsim_db_get_timing": (C.c_int, [vp, C.POINTER(GsimTiming)]),
        "gsim_debug_litmus": (C.c_int, [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_ulonglong)]),
        "gsim_debug_query_flags": (C.c_int, [vp, C.POINTER(C.c_uint8), C.c_uint32, C.POINTER(C.c_uint32)]),
        "gsim_debug_score_table": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint32, C.c_uint32,
                                             C.c_uint32, C.POINTER(C.c_float)]),
        "gsim_debug_sort_desc": (C.c_int, [C.c_int, C.c_void_p, C.c_uint32]),
        "gsim_debug_prefilter_constants": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint32, C.c_int, C.c_float,
                                                      C.POINTER(C.c_float)]),
        "gsim_last_error": (C.c_char_p, []),
        "gsim_version": (C.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise GsimError(rc, load().gsim_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    n = C.c_int(0)
    check(load().gsim_device_count(C.byref(n)))
    return n.value


def device_free_bytes(device: int) -> int:
    v = C.c_size_t(0)
    check(load().gsim_device_free_bytes(device, C.byref(v)))
    return v.value


def available_device_bytes() -> int:
    v = C.c_size_t(0)
    check(load().gsim_available_device_bytes(C.byref(v)))
    return v.value


def next_device(required_bytes: int) -> int:
    d = C.c_int(-1)
    check(load().gsim_next_device(required_bytes, C.byref(d)))
    return d.value


def synth_row(seed: int, kind: int, row: int, fp_bits: int) -> np.ndarray:
    """Row `row` of the synthetic table gsim_db_generate makes (host twin of the device generator)."""
    out = np.empty(fp_bits // 32, dtype=np.uint32)
    check(load().gsim_synth_row(seed, kind, row, fp_bits, _u32(out)))
    return out


def result_block_bytes(k: int) -> int:
    return int(load().gsim_result_block_bytes(k))


def _u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class Table:
    """A fingerprint table on the GPU(s): thin object wrapper over ``gsim_db``."""

    def __init__(self, fp_bits: int):
        self._L = load()
        h = C.c_void_p()
        check(self._L.gsim_db_create(fp_bits, C.byref(h)))
        self._h = h
        self.fp_bits = fp_bits
        self.W = fp_bits // 32

    # -- lifecycle ---------------------------------------------------------
    def add_rows(self, rows: np.ndarray):
        rows = np.ascontiguousarray(rows, dtype=np.uint32).reshape(-1, self.W)
        check(self._L.gsim_db_add_rows(self._h, _u32(rows), rows.shape[0]))
        return self

    def set_fold_factor(self, fold_factor: int):
        check(self._L.gsim_db_set_fold_factor(self._h, fold_factor))
        return self

    def set_fold_full_on_device(self, allow: bool):
        check(self._L.gsim_db_set_fold_full_on_device(self._h, 1 if allow else 0))
        return self

    def fold_factor(self) -> int:
        return int(self._L.gsim_db_fold_factor(self._h))

    def finalize(self, device: int = 0, ndevices: int = 1):
        check(self._L.gsim_db_finalize(self._h, device, ndevices))
        return self

    def generate(self, seed: int, kind: int, first_row: int, nrows: int, device: int = 0, ndevices: int = 1):
        """The synthetic table in HBM; ndevices > 1: split over that many GPUs like finalize(device, ndevices)."""
        if ndevices > 1:
            check(self._L.gsim_db_generate_sharded(self._h, seed, kind, first_row, nrows, device, ndevices))
        else:
            check(self._L.gsim_db_generate(self._h, seed, kind, first_row, nrows, device))
        return self

    def attach_device_rows(self, ptr: int, nrows: int, device: int = 0):
        check(self._L.gsim_db_attach_device_rows(self._h, C.c_void_p(ptr), nrows, device))
        return self

    def close(self):
        if getattr(self, "_h", None):
            self._L.gsim_db_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- accessors -----------------------------------------------------------
    def count(self) -> int:
        return int(self._L.gsim_db_count(self._h))

    def data_bytes(self) -> int:
        return int(self._L.gsim_db_data_bytes(self._h))

    def shard_count(self) -> int:
        return int(self._L.gsim_db_shard_count(self._h))

    def shard_devices(self):
        return [int(self._L.gsim_db_shard_device(self._h, i)) for i in range(self.shard_count())]

    def row(self, i: int) -> np.ndarray:
        out = np.empty(self.W, dtype=np.uint32)
        check(self._L.gsim_db_row(self._h, i, _u32(out)))
        return out

    # -- search ----------------------------------------------------------------
    def search(self, queries, k, cutoff=0.0, metric=METRIC_TANIMOTO, alpha=1.0, beta=1.0):
        """-> (list of HIT_DTYPE arrays, one per query; approx uint64 array)"""
        q = np.ascontiguousarray(queries, dtype=np.uint32).reshape(-1, self.W)
        nq = q.shape[0]
        hits = np.zeros((nq, max(k, 1)), dtype=HIT_DTYPE)
        counts = np.zeros(nq, dtype=np.uint32)
        approx = np.zeros(nq, dtype=np.uint64)
        check(self._L.gsim_db_search(self._h, _u32(q), nq, k, cutoff, metric, alpha, beta,
                                     hits.ctypes.data_as(C.c_void_p), _u32(counts),
                                     approx.ctypes.data_as(C.POINTER(C.c_uint64))))
        return [hits[i, :counts[i]].copy() for i in range(nq)], approx

    def make_search_buffers(self, nq, k):
        """Preallocated outputs for :meth:`search_into` (latency-sensitive callers)."""
        return (np.zeros((nq, max(k, 1)), dtype=HIT_DTYPE), np.zeros(nq, dtype=np.uint32),
                np.zeros(nq, dtype=np.uint64))

    def search_into(self, queries, k, bufs, cutoff=0.0, metric=METRIC_TANIMOTO, alpha=1.0, beta=1.0):
        """gsim_db_search into caller-owned buffers; queries must be a C-contiguous uint32 array."""
        hits, counts, approx = bufs
        check(self._L.gsim_db_search(self._h, _u32(queries), hits.shape[0], k, cutoff, metric, alpha, beta,
                                     hits.ctypes.data_as(C.c_void_p), _u32(counts),
                                     approx.ctypes.data_as(C.POINTER(C.c_uint64))))

    def search_each_into(self, queries, k, bufs, cutoff=0.0, metric=METRIC_TANIMOTO, alpha=1.0, beta=1.0):
        """gsim_db_search_each into caller-owned buffers: the queries one after the other through the single-query path."""
        hits, counts, approx = bufs
        check(self._L.gsim_db_search_each(self._h, _u32(queries), hits.shape[0], k, cutoff, metric, alpha, beta,
                                          hits.ctypes.data_as(C.c_void_p), _u32(counts),
                                          approx.ctypes.data_as(C.POINTER(C.c_uint64))))

    def search_timed_into(self, queries, k, bufs, cutoff=0.0, metric=METRIC_TANIMOTO, alpha=1.0, beta=1.0) -> np.ndarray:
        """gsim_db_search_timed: the queries one at a time, nothing enqueued ahead; -> seconds per query (measured in the library)."""
        hits, counts, approx = bufs
        sec = np.zeros(hits.shape[0], dtype=np.float64)
        check(self._L.gsim_db_search_timed(self._h, _u32(queries), hits.shape[0], k, cutoff, metric, alpha, beta,
                                           hits.ctypes.data_as(C.c_void_p), _u32(counts),
                                           approx.ctypes.data_as(C.POINTER(C.c_uint64)), sec.ctypes.data_as(C.POINTER(C.c_double))))
        return sec

    def search_cpu(self, queries, k, cutoff=0.0):
        q = np.ascontiguousarray(queries, dtype=np.uint32).reshape(-1, self.W)
        nq = q.shape[0]
        hits = np.zeros((nq, max(k, 1)), dtype=HIT_DTYPE)
        counts = np.zeros(nq, dtype=np.uint32)
        check(self._L.gsim_db_search_cpu(self._h, _u32(q), nq, k, cutoff, hits.ctypes.data_as(C.c_void_p),
                                         _u32(counts)))
        return [hits[i, :counts[i]].copy() for i in range(nq)]

    def set_stream(self, stream_ptr: int):
        check(self._L.gsim_db_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_row_base(self, base: int):
        check(self._L.gsim_db_set_row_base(self._h, base))

    def search_device(self, query, k, d_result_ptr, cutoff=0.0, metric=METRIC_TANIMOTO, alpha=1.0, beta=1.0):
        q = np.ascontiguousarray(query, dtype=np.uint32).reshape(self.W)
        check(self._L.gsim_db_search_device(self._h, _u32(q), k, cutoff, metric, alpha, beta,
                                            C.c_void_p(d_result_ptr)))

    def search_batch_device(self, queries, k, d_results_ptr, cutoff=0.0, metric=METRIC_TANIMOTO, alpha=1.0, beta=1.0):
        """nq queries; result block q lands at d_results_ptr + q * result_block_bytes(k) (device)."""
        q = np.ascontiguousarray(queries, dtype=np.uint32).reshape(-1, self.W)
        check(self._L.gsim_db_search_batch_device(self._h, _u32(q), len(q), k, cutoff, metric, alpha, beta,
                                                  C.c_void_p(d_results_ptr)))

    def set_comm(self, comm):
        """Route multi-shard searches through `comm` (a :class:`Comm`; None: back to the host merge)."""
        check(self._L.gsim_db_set_comm(self._h, comm._h if comm is not None else None))
        self._comm = comm  # (keeps it alive)

    def set_comm_root(self, shard: int):
        """The shard whose device merges the gathered blocks (every device holds them all after the all-gather)."""
        check(self._L.gsim_db_set_comm_root(self._h, shard))

    def enable_timing(self, enable=True):
        check(self._L.gsim_db_enable_timing(self._h, 1 if enable else 0))

    def timing(self) -> dict:
        t = GsimTiming()
        check(self._L.gsim_db_get_timing(self._h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in GsimTiming._fields_}

    def query_flags(self, n: int) -> np.ndarray:
        """One byte per query of the last search call made with timing enabled (gsim_debug_query_flags): 1 handed back by its own
        single launch, 2 re-run behind a launch that did not close, 4 torn block, 8 routed around the single launch by the
        back-off, 16 / 32 the same two for the large-k publishing route."""
        out = np.zeros(n, dtype=np.uint8)
        w = C.c_uint32(0)
        check(self._L.gsim_debug_query_flags(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8)), n, C.byref(w)))
        return out[:w.value]


def litmus(test: int, workgroups: int = 256, iterations: int = 100000, device: int = 0) -> dict:
    """gsim_debug_litmus -> {loads, torn, headers_seen, stale_first_read, never_landed, rereads, timed_out, stores}"""
    st = (C.c_ulonglong * 8)()
    check(load().gsim_debug_litmus(device, test, workgroups, iterations, st))
    return dict(zip(("loads", "torn", "headers_seen", "stale_first_read", "never_landed", "rereads", "timed_out", "stores"), [int(x) for x in st]))


def rccl_info() -> dict:
    """gsim_rccl_info: the RCCL header version the library was built with, the version and file of the librccl.so bound to this process."""
    L = load()
    h, r = C.c_int(0), C.c_int(0)
    buf = C.create_string_buffer(1024)
    check(L.gsim_rccl_info(C.byref(h), C.byref(r), buf, len(buf)))
    return {"header_version": h.value, "runtime_version": r.value, "path": buf.value.decode()}


class Comm:
    """``gsim_comm``: the RCCL communicator of an in-process multi-device handle (ncclCommInitAll)."""

    def __init__(self, devices):
        self._L = load()
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        check(self._L.gsim_comm_create(devs, len(devices), C.byref(h)))
        self._h = h

    def size(self) -> int:
        return int(self._L.gsim_comm_size(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.gsim_comm_destroy(self._h)
            self._h = None


def fold_fingerprint(fp, fold_factor: int) -> np.ndarray:
    fp = np.ascontiguousarray(fp, dtype=np.uint32)
    out = np.zeros(len(fp) // fold_factor, dtype=np.uint32)
    check(load().gsim_fold_fingerprint(_u32(fp), len(fp), fold_factor, _u32(out)))
    return out


def merge_device(device, stream_ptr, d_blocks_ptr, nblocks, block_bytes, k, d_result_ptr):
    check(load().gsim_merge_device(device, C.c_void_p(stream_ptr), C.c_void_p(d_blocks_ptr), nblocks, block_bytes, k,
                                   C.c_void_p(d_result_ptr)))


def merge_device_batch(device, stream_ptr, d_blocks_ptr, nranks, nq, block_bytes, k, d_results_ptr):
    check(load().gsim_merge_device_batch(device, C.c_void_p(stream_ptr), C.c_void_p(d_blocks_ptr), nranks, nq,
                                         block_bytes, k, C.c_void_p(d_results_ptr)))


def merge_host(blocks: bytes, nblocks: int, block_bytes: int, k: int) -> bytes:
    """gsim_merge_host on a bytes object holding nblocks result blocks."""
    src = (C.c_ubyte * len(blocks)).from_buffer_copy(blocks)
    out = (C.c_ubyte * result_block_bytes(k))()
    check(load().gsim_merge_host(C.cast(src, C.c_void_p), nblocks, block_bytes, k, C.cast(out, C.c_void_p)))
    return bytes(out)


def make_result_block(hits: np.ndarray, approx: int, k: int, flags: int = 0) -> bytes:
    """Serialise (hits, approx) as one result block of capacity k."""
    hdr = np.zeros(1, dtype=HEADER_DTYPE)
    hdr["count"], hdr["flags"], hdr["approx"] = len(hits), flags, approx
    body = np.zeros(k, dtype=HIT_DTYPE)
    body[:len(hits)] = hits
    raw = hdr.tobytes() + body.tobytes()
    return raw + b"\0" * (result_block_bytes(k) - len(raw))


def parse_result_block(buf: bytes, k: int):
    """bytes of one result block -> (HIT_DTYPE array, approx, flags)"""
    hdr = np.frombuffer(buf, dtype=HEADER_DTYPE, count=1)[0]
    hits = np.frombuffer(buf, dtype=HIT_DTYPE, count=int(hdr["count"]), offset=HEADER_DTYPE.itemsize)
    return hits.copy(), int(hdr["approx"]), int(hdr["flags"])


def debug_score_table(metric, alpha, beta, a, max_b, max_c, device=0) -> np.ndarray:
    out = np.empty((max_c + 1, max_b + 1), dtype=np.float32)
    check(load().gsim_debug_score_table(device, metric, alpha, beta, a, max_b, max_c,
                                        out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def debug_sort_desc(keys: np.ndarray, device=0) -> np.ndarray:
    """The device sort of the large-k and folded paths (launch_sort_desc) on a uint64 array of 2^i keys, descending."""
    out = np.ascontiguousarray(keys, dtype=np.uint64).copy()
    check(load().gsim_debug_sort_desc(device, out.ctypes.data_as(C.c_void_p), len(out)))
    return out


def debug_prefilter_constants(metric, alpha, beta, max_qa, cutoff=None, device=0) -> np.ndarray:
    """Pre-filter constants of the matrix-core pass, [max_qa+1, 512 (or 1 with a cutoff), 4]; device < 0: host twin."""
    nlev = 1 if cutoff is not None else 512
    out = np.empty((max_qa + 1, nlev, 4), dtype=np.float32)
    check(load().gsim_debug_prefilter_constants(device, metric, alpha, beta, max_qa, 1 if cutoff is not None else 0,
                                                0.0 if cutoff is None else cutoff, out.ctypes.data_as(C.POINTER(C.c_float))))
    return out
