#!/usr/bin/env python3
"""Benchmark of the hot path: Tanimoto top-1000 scan of synthetic 1024-bit tables.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` (N > 1) without a launcher starts its own N ranks (one process per GPU, 127.0.0.1
rendezvous); under torch.distributed.run the ranks are taken from the environment.

A "step" is one pass of the hot path over one batch of synthetic input: --queries-per-step
(default 128: 20 steps keep the GPU busy for ~5 s at 100 M rows) single queries, one after the other -- for each, the packed table resident in HBM
is scanned once, the exact top-k is selected, (N > 1: the per-GPU top-k blocks are all-gathered
over RCCL/xGMI and merged), and the k hits land in host memory.  Weak scaling: every rank holds
--rows-per-gpu rows (default 100 M at N = 1 = BASELINE.json configs[2], the HBM-bound roofline
run; 125 M at N > 1 so that 8 GPUs hold the 1 B-row table of configs[3]).
value = rows of the whole table * queries / max-over-ranks time.

Prints ONE JSON line (rank 0).  `roofline` is the dominant kernel's algorithmic bytes (128 B per
fingerprint per pass) over its HIP-event duration on the stream it runs on.  At N = 1 the line
also carries `configs`: every single-GPU BASELINE config measured in this run (configs[1]: 1 M
rows; configs[2]: the headline; configs[4]'s per-GPU shape: 125 M x 2048-bit, Tversky, 256-query
batches) plus the 1 M / 100 M-row tables again with MORGAN-SHAPED rows (clustered, tie-heavy: what the
reference's own numbers are quoted on), and `cpu_baseline`: the reference's host functor path + top-k on this box's cores
(oracle/_ref when present, else the oracle port) on bounded samples.
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
MFMA_FP4_PEAK_TFLOPS = 10000.0  # dense MX-FP4, MI355X_MICROARCH.md (measured 9099)
DB_SEED = 0x5EED0001
GOLDEN = 0x9E3779B97F4A7C15
M64 = (1 << 64) - 1
SYNTH_SPARSE, SYNTH_DENSE, SYNTH_MORGAN = 0, 1, 2


def _splitmix64(x):
    z = (x + GOLDEN) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def synth_row(seed, kind, row, W):
    """One row of the synthetic table, from the generator's own host twin in the product library
    (gsim_synth_row) -- the query fingerprints are rows of the table: guaranteed score-1.0 self hit."""
    from gpusimilarity_amd import capi
    return capi.synth_row(seed, kind, row, W * 32)


def query_row(q, nrows):
    return _splitmix64((0xC0FFEE + q) & M64) % nrows


# ---------------------------------------------------------------------------------------------
# self-spawn: `python bench.py --gpus N` without a launcher
# ---------------------------------------------------------------------------------------------

def spawn_ranks(n):
    """Start n ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, the
    contract of torch.distributed.run) and wait for them.  Rank 0 inherits stdout (the JSON line)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GSIM_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


# ---------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------

def _timed(run, budget_s, max_reps=1000):
    run()
    reps, t0 = 0, time.perf_counter()
    while True:
        run()
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= max_reps:
            return reps, el


def cpu_baseline(fp_bits, k, kind):
    """The reference's host path on this box's cores, bounded samples:
    TanimotoFunctorCPU on all hardware threads + the canonical top-k by selection, on configs[1]
    (1 M rows) and an 8 M-row slice; the reference's own top_results_bubble_sort
    (fingerprintdb_cuda.cpp:92-103, what search_cpu really runs) at 1 M rows for k = 10 and k;
    and configs[0] (test/small.fsim, top-10)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    W = fp_bits // 32
    cores = os.cpu_count() or 1
    use_ref = O.ref_lib() is not None and hasattr(O.ref_lib(), "gsref_search_topk")
    parts = []

    def search_rate(n, budget):
        db = O.synth_rows(DB_SEED, kind, 0, n, W)
        q = db[query_row(0, n)]
        nt = min(cores, max(1, n // 4096))
        if use_ref:
            tab = O.RefTable(db)
            run = lambda: tab.search_topk(q, k, nthreads=nt)  # noqa: E731
        else:
            run = lambda: O.search(q, db, k, 0.0, nthreads=nt)  # noqa: E731
        reps, el = _timed(run, budget)
        return {"rows": n, "threads": nt, "queries": reps, "seconds": round(el, 2), "ms_per_query": 1e3 * el / reps,
                "fingerprints_per_s": n * reps / el, "GB_per_s": n * (fp_bits // 8) * reps / el / 1e9}, db, q

    big, _, _ = search_rate(8_000_000, 8.0)  # 1 GB of fingerprints: larger than the host's last-level caches
    parts.append(dict(big, what="scan + top-%d, 8 M-row slice" % k))
    one, db1, q1 = search_rate(1_000_000, 3.0)
    parts.append(dict(one, what="scan + top-%d, configs[1] (1 M rows)" % k))
    # the reference's own partial bubble sort over the 1 M scores (O(k N)): what FingerprintDB::search_cpu pays
    sort = O.ref_sort_lib()
    scores = O.tanimoto_raw(q1, db1)[0]
    for kk in (10, k):
        idx = np.arange(len(scores), dtype=np.int32)
        if sort is not None:
            import ctypes as C
            sort.gsref_bubble_sort.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int, C.c_int]
            i2, s2 = idx.copy(), scores.copy()
            t0 = time.perf_counter()
            sort.gsref_bubble_sort(i2.ctypes.data_as(C.POINTER(C.c_int)), s2.ctypes.data_as(C.POINTER(C.c_float)), len(idx), kk)
            el = time.perf_counter() - t0
            src = "reference top_results_bubble_sort (fingerprintdb_cuda.cpp compiled in place)"
        else:
            t0 = time.perf_counter()
            O.bubble_sort(idx, scores, kk)
            el = time.perf_counter() - t0
            src = "oracle port of top_results_bubble_sort"
        parts.append({"what": "%s, 1 M scores, k = %d, 1 thread" % (src, kk), "rows": len(idx), "threads": 1,
                      "seconds": round(el, 3), "ms_per_query": 1e3 * el})
    try:  # configs[0]: the reference's own fixture, CPU path, top-10
        from gpusimilarity_amd.fsim import read_fsim
        f = read_fsim(os.path.join(ROOT, "tests", "golden", "small.fsim"))
        small = np.ascontiguousarray(np.concatenate(f.fp_blocks), dtype=np.uint32)
        reps, el = _timed(lambda: O.search_cpu(small[0], small, 10), 0.5, 20000)
        parts.append({"what": "configs[0]: test/small.fsim (%d rows), search_cpu top-10 (oracle port), 1 thread" % len(small),
                      "rows": int(len(small)), "threads": 1, "queries": reps, "ms_per_query": 1e3 * el / reps})
    except Exception as e:  # a report, never a reason to fail
        parts.append({"what": "configs[0] small.fsim not timed: %r" % (e,)})
    return {"value": big["fingerprints_per_s"], "unit": "fingerprints/s", "cores": big["threads"],
            "kind": "reference" if use_ref else "port",
            "sample": "%d queries over a %d-row x %d-bit synthetic slice, %d threads, %.1f s; %s" % (
                big["queries"], big["rows"], fp_bits, big["threads"], big["seconds"],
                "reference TanimotoFunctorCPU (calculation_functors.cpp compiled in place, oracle/_ref) + canonical top-%d "
                "by per-thread selection and merge" % k if use_ref else "oracle port gso_search (scan + heap top-%d)" % k),
            "parts": parts}


# ---------------------------------------------------------------------------------------------
# HBM traffic of the dominant kernel from the PMC counters (MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots")
# ---------------------------------------------------------------------------------------------

def pmc_child(rows, fp_bits, kind, k):
    """`bench.py --pmc-child`: the table, six single queries through the synchronous C ABI, nothing else -- the run
    rocprofv3 counts (one counter pass per process: FETCH_SIZE and WRITE_SIZE do not fit one pass)."""
    from gpusimilarity_amd import capi
    t = capi.Table(fp_bits)
    t.generate(DB_SEED, kind, 0, rows, 0)
    bufs = t.make_search_buffers(1, k)
    for i in range(6):
        t.search_into(np.ascontiguousarray(synth_row(DB_SEED, kind, query_row(i, rows), fp_bits // 32)).reshape(1, -1), k, bufs)
    t.close()


def pmc_traffic(rows, fp_bits, kind_name, k, kernel_substr):
    """-> (bytes read from HBM per launch, bytes written, note).  Separate `rocprofv3 --kernel-trace --pmc` passes of a
    child of this script; FETCH_SIZE is reported in KiB and, on gfx950, counts a 16 B/lane streaming read at half its
    bytes (128-B requests tallied at 64 B): doubled, as the guide prescribes.  WRITE_SIZE is uncalibrated (reported as is)."""
    import csv
    import glob
    if shutil.which("rocprofv3") is None:
        return None, None, "rocprofv3 is not on PATH"
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="gsim_pmc_", dir="/tmp")
        try:
            cmd = ["rocprofv3", "--output-format", "csv", "--kernel-trace", "--pmc", counter, "-d", d, "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-child", "--rows-per-gpu", str(rows), "--fp-bits", str(fp_bits), "--kind", kind_name,
                   "--k", str(k)]
            p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, timeout=600)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        vals.append(float(r["Counter_Value"]))
            if p.returncode != 0 or not vals:
                return None, None, "rocprofv3 --pmc %s failed (rc %d, %d launches seen): %s" % (
                    counter, p.returncode, len(vals), p.stderr.decode("utf-8", "replace")[-300:])
            res[counter] = (sum(vals) / len(vals), len(vals))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch = res["FETCH_SIZE"][0] * 1024.0 * 2.0
    write = res["WRITE_SIZE"][0] * 1024.0
    note = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes of a child process (%d launches each); "
            "FETCH_SIZE [KiB] x 1024 x 2 (gfx950 tallies the 128-B requests of a 16 B/lane streaming read at 64 B); "
            "WRITE_SIZE [KiB] x 1024, uncalibrated" % res["FETCH_SIZE"][1])
    return fetch, write, note



# ---------------------------------------------------------------------------------------------
# output: ONE short JSON line (the contract) + bench_detail.json beside this script (everything else)
# ---------------------------------------------------------------------------------------------

LINE_CAP = 12000  # bytes; the driver parses the line from a bounded tail of stdout (round 5's 20.5 KB line was not parsed)
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline", "timed_region_s", "queries_per_step", "ms_per_query",
                 "whole_path_hbm_frac", "sync_ms_median", "sync_ms_p95", "calls", "queries_per_s")


def _sig(x, digits=6):
    """floats to `digits` significant digits, recursively (a line of 17-digit doubles is twice as long as it needs to be)"""
    if isinstance(x, float):
        return float("%.*g" % (digits, x)) if x == x and x not in (float("inf"), float("-inf")) else None
    if isinstance(x, dict):
        return {k_: _sig(v, digits) for k_, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact_line(out, detail_path):
    """The contract line: the contract's keys first and short; `configs`, `widths`, `server_latency`, `cpu_baseline.parts`,
    the full `collective.per_rank`, the full `routes` and the notes live in the detail file only."""
    line = {k_: out[k_] for k_ in CONTRACT_KEYS if k_ in out}
    if "config" in line:
        line["config"] = {k_: v for k_, v in out["config"].items() if k_ != "query_execution"}
    if out.get("roofline"):
        keep = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "kernel_ms_avg",
                "algorithmic_bytes_per_launch", "queries_handed_back", "issue")
        line["roofline"] = {k_: out["roofline"].get(k_) for k_ in keep if k_ in out["roofline"] or k_ == "traffic"}
        if isinstance(line["roofline"].get("issue"), dict):
            line["roofline"]["issue"] = {k_: v for k_, v in line["roofline"]["issue"].items() if k_ != "source"}
    if out.get("cpu_baseline"):
        line["cpu_baseline"] = {k_: out["cpu_baseline"].get(k_) for k_ in ("value", "unit", "cores", "kind", "sample")}
    if out.get("collective"):
        c = out["collective"]
        cc = {k_: c.get(k_) for k_ in ("backend", "world", "launcher", "data_path_collective", "queries_in_flight", "shared_gpu_test_mode", "rccl", "identical_hits_across_routes")
              if c.get(k_) is not None or k_ in ("backend", "world")}
        cc["ranks"] = [{"rank": m.get("rank"), "device": m.get("device"), "pci_bus_id": m.get("pci_bus_id")} for m in c.get("ranks", [])]
        if c.get("per_rank"):
            pr = []
            for m in c["per_rank"]:
                e = {"rank": m.get("rank")}
                e.update(m.get("phases_per_query") or {})
                tw = m.get("single_gpu_twin") or {}
                e["twin_ms_per_query"], e["twin_kernel_ms"] = tw.get("ms_per_query"), tw.get("kernel_ms_avg")
                pr.append(e)
            cc["per_rank"] = pr
        line["collective"] = cc
    if out.get("routes"):
        keep = ("ms_per_query", "fingerprints_per_s", "kernel_ms_avg_first_shard", "gather_us_avg", "merge_us_avg", "collectives", "transport",
                "identical_to_host_merge", "backend", "queries_handed_back", "error")
        line["routes"] = {name: {k_: r.get(k_) for k_ in keep if r.get(k_) is not None} for name, r in out["routes"].items()}
    if out.get("summary"):
        line["summary"] = out["summary"]
    line["detail"] = detail_path
    line = _sig(line)
    text = json.dumps(line, separators=(",", ":"))
    while len(text) >= LINE_CAP and line.get("summary", {}).get("configs"):
        line["summary"]["configs"].pop()  # never reached with today's entries (the line is ~4 KB); a cap is a cap
        line["summary"]["truncated"] = True
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_CAP, len(text)
    return text


def emit(out, json_fd, detail_name="bench_detail.json"):
    """Write the full record to bench_detail.json (beside bench.py; and under gpurun_out/ when that directory exists, so that a
    gpurun call brings it home) and the contract line to stdout."""
    paths = [os.path.join(ROOT, detail_name)]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", detail_name))
    written = None
    for p_ in paths:
        try:
            with open(p_, "w") as f:
                json.dump(out, f, indent=1)
                f.write("\n")
            written = written or os.path.relpath(p_, ROOT)
        except OSError:
            pass
    os.write(json_fd, (compact_line(out, written) + "\n").encode())

# ---------------------------------------------------------------------------------------------
# GPU runs
# ---------------------------------------------------------------------------------------------

def time_queries(ctx, table, total_rows, R, fp_bits, kind, k, steps, warmup, qps, sharded, barrier=True, first_row=0):
    """steps x qps single queries (after warmup x qps), barrier + synchronize on both sides, max over ranks."""
    import torch
    import torch.distributed as dist
    from gpusimilarity_amd.sharded import ShardedSearch
    W = fp_bits // 32
    nq = (warmup + steps) * qps
    distinct = min(nq, 64)  # distinct queries, cycled
    queries = [synth_row(DB_SEED, kind, first_row + query_row(i, total_rows), W) for i in range(distinct)]
    bufs = table.make_search_buffers(qps, k)
    # eight gather+merge objects on the one stream, as many as the N = 1 path keeps enqueued: while the host waits for
    # query i's hits, the local search, all-gather and merge of queries i + 1 ... i + 7 are already enqueued behind it
    # (each object has its own blocks and completion event)
    depth = 8
    pair = [ShardedSearch(table, k, ctx["dev"], stream=ctx["stream"]) for _ in range(depth)] if sharded else None
    ss = pair[0] if sharded else None
    last = [ss]
    steps_q = [np.ascontiguousarray(np.stack([queries[(s_ * qps + j) % distinct] for j in range(qps)])) for s_ in range(warmup + steps)]

    def one_step(s_):
        if not sharded:
            # ONE call of the C ABI per step: gsim_db_search_each answers the step's queries strictly one after the
            # other through the synchronous single-query path (FingerprintDB::search called qps times, as the
            # reference's C++ server does) -- each query's kernel writes its hits into pinned host memory and the
            # next one is launched when they are there; no Python between the queries
            table.search_each_into(steps_q[s_], k, bufs)
            return
        pending = []
        for j in range(qps):
            cur = pair[j % depth]
            if len(pending) == depth:
                pending.pop(0).synchronize()  # a query is done when its k hits are in host memory
            cur.enqueue(steps_q[s_][j])  # local top-k -> all-gather (k*12+16 B per GPU) -> rank merge -> D2H
            pending.append(cur)
        for cur in pending:
            cur.synchronize()
        last[0] = pending[-1]

    for s_ in range(warmup):
        one_step(s_)
    if warmup:
        if sharded:
            hits, approx, _ = last[0].result()
        else:
            hits, approx = bufs[0][qps - 1, :bufs[1][qps - 1]], int(bufs[2][qps - 1])
        want_row = first_row + query_row((warmup * qps - 1) % distinct, total_rows)
        # (Morgan-shaped tables hold exact duplicates: the query's own row is one of the rows scoring 1.0)
        assert len(hits) == min(k, total_rows) and hits["score"][0] == 1.0 and want_row in hits["row"][hits["score"] == 1.0], \
            "self hit missing: %r" % (hits[:3],)
        assert approx == total_rows
    table.enable_timing(True)
    if sharded:
        for o in pair:
            o.enable_phase_timing(True)
    torch.cuda.synchronize()
    if dist.is_initialized() and barrier:
        dist.barrier()
    t0 = time.perf_counter()
    for s_ in range(warmup, warmup + steps):
        one_step(s_)
    torch.cuda.synchronize()
    if dist.is_initialized() and barrier:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if ctx["world"] > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctx["dev"] if ctx["backend"] == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tm = table.timing()
    table.enable_timing(False)
    phases = None
    if sharded:  # this rank's mean stream time per query and phase (HIP events around each step of ShardedSearch.enqueue)
        ph = [o.phases() for o in pair if o.phases()]
        ntot = sum(p_["n"] for p_ in ph)
        if ntot:
            phases = {key: sum(p_[key] * p_["n"] for p_ in ph) / ntot for key in ("search_ms", "gather_us", "merge_us", "d2h_us")}
            phases["queries"] = ntot
        for o in pair:
            o.enable_phase_timing(False)
    # SURVEY 8(d)'s latency: wall time from query-on-host to top-k-on-host, ONE query at a time through gsim_db_search
    # (nothing enqueued ahead, unlike the timed region above), median and p95 of >= 200 calls
    sync = None
    if not sharded:
        nlat = 200 if R * fp_bits <= 100_000_000 * 1024 else 50
        qlat = np.ascontiguousarray(np.stack([queries[j % distinct] for j in range(nlat + 20)]))
        blat = table.make_search_buffers(nlat + 20, k)
        lat = sorted(table.search_timed_into(qlat, k, blat)[20:])  # gsim_db_search_timed: measured inside the library
        # ... and what a Python caller sees around one ctypes call per query (its binding overhead included)
        b1 = table.make_search_buffers(1, k)
        plat = []
        for j in range(60):
            q1 = queries[j % distinct].reshape(1, -1)
            t1 = time.perf_counter()
            table.search_into(q1, k, b1)
            if j >= 10:
                plat.append(time.perf_counter() - t1)
        plat.sort()
        sync = {"calls": len(lat), "sync_ms_median": 1e3 * lat[len(lat) // 2], "sync_ms_p95": 1e3 * lat[int(len(lat) * 0.95)],
                "sync_ms_min": 1e3 * lat[0], "sync_ms_median_through_python_ctypes": 1e3 * plat[len(plat) // 2]}
    n = max(1, tm["queries"])  # queries timed with HIP events (the first 1024 of the timed region)
    nall = max(1, steps * qps)  # queries the device-side totals cover (all of the timed region)
    kernel_ms = tm["scan_ms_sum"] / n
    # Small tables, round 6: gsim_db_search_each alternates consecutive queries between two half-grid lanes (own stream each),
    # so TWO launches of the dominant kernel run side by side and each takes longer than a whole-grid launch would: the time the
    # GPU spends per launch is kernel_ms / 2 (gsim_timing.lane_queries says whether the timed region ran that way)
    lanes = 2 if tm.get("lane_queries", 0) > 0 and not sharded else 1
    kernel_ms_each = kernel_ms
    kernel_ms = kernel_ms / lanes
    algo = R * (fp_bits // 8)  # bytes per launch of the dominant kernel on one GPU
    achieved = algo / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    res = {
        "seconds": elapsed, "queries": steps * qps, "ms_per_query": 1e3 * elapsed / (steps * qps), "sync_latency": sync, "phases": phases,
        "fingerprints_per_s": total_rows * steps * qps / elapsed,
        "whole_path_hbm_frac": (total_rows * (fp_bits // 8) / (elapsed / (steps * qps))) / (HBM_PEAK_GBS * 1e9 * ctx["world"]),
        "roofline": {
            "kernel": kernel_label(W, table_uses_fused(k, tm)) + (" [publishes only; fused_binsort_kernel + binrank_emit_kernel rank]"
                                                                 if tm.get("large_k_single_scan", 0) > 0 else "") +
                      (" [two half-grid launches side by side: kernel_ms_avg = a launch's HIP-event duration / 2]" if lanes > 1 else ""),
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "traffic_note": "not collected for this entry (the headline entry runs the rocprofv3 --pmc passes)",
            "kernel_ms_avg": kernel_ms, "other_kernels_ms_avg": tm["select_ms_sum"] / n,
            "launches_side_by_side": lanes, "kernel_ms_avg_each_launch": kernel_ms_each,
            "algorithmic_bytes_per_launch": algo,
            "candidates_per_query": tm["candidates_sum"] / nall, "finalists_per_query": tm["finalists_sum"] / nall,
            "queries_handed_back": tm["handed_back"], "timed_with_hip_events": tm["queries"],
        },
    }
    return res, ss


def kernel_label(W, fused):
    """The dominant kernel's name for rows of W words (csrc/gsim_fused.hip launch_fused, gsim_scan.hip launch_scan)."""
    L = W // 4
    if W % 4 == 0 and L & (L - 1) == 0:
        return ("fused_kernel<%d,8> (single launch: scan + publish + select)" % L) if fused else "scan_kernel<%d,8>" % L
    odd = W
    while odd % 2 == 0:
        odd //= 2
    if W % 4 == 0 and odd <= 15:
        c = {3: 3, 5: 2}.get(odd, 1)
        return ("fused_kernel<-%d,%d> (single launch, rows of %d sixteen-byte units streamed through registers)" % (odd, c, L)) if fused \
            else "scan_ragged_kernel<%d,%d>" % (odd, c)
    if W in (3, 5, 7, 6, 10, 14, 9, 11, 18, 22) and fused:
        return "fused_kernel<-%d,%d,words> (single launch, rows of %d words streamed at word granularity)" % (W, {3: 3, 6: 3, 5: 2, 10: 2}.get(W, 1), W)
    return "scan_generic_kernel (LDS-staged rows, four-kernel pipeline)"


def table_uses_fused(k, tm):
    """the single launch scanned (and ranked, or -- gsim_timing.large_k_single_scan -- published for the kernels that rank)"""
    return (k <= 8192 or tm.get("large_k_single_scan", 0) > 0) and os.environ.get("GSIM_FUSED", "1") != "0" and tm["handed_back"] < max(1, tm["queries"])


# The matrix-core batch kernel against its OTHER roofline ("report it against both"): instruction issue.  Two waves per
# SIMD share one issue port with the MFMAs; scripts/mfma_fp4_probe.hip (profiles/r01_mfma_fp4_probe.txt) measured the cycles per
# MFMA and SIMD (at the nominal 2.4 GHz the 10 PF peak is quoted on: 32 = the peak) for n extra vector instructions per MFMA:
MFMA_ISSUE_PROBE = ((0, 38.2), (4, 38.1), (8, 57.6), (12, 69.7))
# ... and the shipped 2048-bit kernel's vector instructions per MFMA, SQ_INSTS_VALU / SQ_INSTS_MFMA
# (profiles/r05_batch_mfma_pmc_raw.txt; 5 of them are the operand expansion, round 5 could not take them out: DESIGN.md section 3)
BATCH_VALU_PER_MFMA = {64: 7.6}
# ... and of its dense-cutoff variant (batch_mfma_kernel<64,1,1,4096,2,true>: one row tile per wave -- with two it spills 104 registers --
# and the band classification of every pair: profiles/r06_batch_cutoff_pmc_raw.txt, 12.80 G vector instructions per 1.000 G MFMAs)
BATCH_VALU_PER_MFMA_DENSE = {64: 12.8}


def issue_roofline(W, frac, dense=False):
    n = (BATCH_VALU_PER_MFMA_DENSE if dense else BATCH_VALU_PER_MFMA).get(W)
    if n is None or not frac:
        return None
    pts = MFMA_ISSUE_PROBE
    (n0, c0), (n1, c1) = pts[-2], pts[-1]
    cyc = c1 + (c1 - c0) * (n - n1) / (n1 - n0)  # beyond the probe's last point: its last slope (3.0 cycles per instruction)
    for (n0, c0), (n1, c1) in zip(pts, pts[1:]):
        if n0 <= n <= n1:
            cyc = c0 + (c1 - c0) * (n - n0) / (n1 - n0)
            break
    ceiling = 32.0 / cyc
    out = {"valu_per_mfma": n, "issue_ceiling_frac_of_peak": ceiling, "issue_frac": frac / ceiling,
           "source": "profiles/r01_mfma_fp4_probe.txt (cycles per MFMA at n vector instructions, two waves per SIMD) and "
                     + ("profiles/r06_batch_cutoff_pmc_raw.txt" if dense else "profiles/r05_batch_mfma_pmc_raw.txt") + " (SQ_INSTS_VALU / SQ_INSTS_MFMA of this kernel)"}
    if dense:
        out["note"] = ("below its issue ceiling: the dense variant runs ONE row tile per wave (a single dependent MFMA chain; with two the "
                       "kernel spills 104 registers and takes 65.8 ms) -- DESIGN.md section 3")
    return out


def time_batches(ctx, table, total_rows, R, fp_bits, kind, k, Q, steps, warmup, sharded, cutoff=0.0):
    """BASELINE configs[4]: Tversky(0.3, 0.7), Q-query batches, top-k per query; a step = one batch.
    Rows shard over the ranks, every rank scores all Q queries against its shard (the matrix-core
    pass), ONE all-gather of Q result blocks per step, one merge launch."""
    import torch
    import torch.distributed as dist
    from gpusimilarity_amd import capi
    from gpusimilarity_amd.sharded import ShardedBatchSearch
    W = fp_bits // 32
    kw = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
    nb = warmup + steps
    batches = [np.ascontiguousarray(np.stack([synth_row(DB_SEED, kind, query_row(b * Q + i, total_rows), W)
                                              for i in range(Q)]), dtype=np.uint32)
               for b in range(min(nb, 3))]  # a few distinct batches, cycled
    sb = ShardedBatchSearch(table, k, Q, ctx["dev"], stream=ctx["stream"], search_kwargs=kw) if sharded else None
    bufs = table.make_search_buffers(Q, k)  # caller-owned outputs of the synchronous C-ABI call

    def one_batch(qs):
        if not sharded:
            table.search_into(qs, k, bufs, cutoff, **kw)
            return
        sb.enqueue(qs)
        sb.synchronize()

    for b in range(warmup):
        one_batch(batches[b % len(batches)])
    if warmup:
        b = (warmup - 1) % len(batches)
        first = sb.results()[0][0] if sharded else bufs[0][0, :bufs[1][0]]
        assert int(first["row"][0]) == query_row(b * Q, total_rows) and first["score"][0] == 1.0, "self hit missing"
    table.enable_timing(True)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    for b in range(warmup, nb):
        one_batch(batches[b % len(batches)])
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if ctx["world"] > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctx["dev"] if ctx["backend"] == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tm = table.timing()
    table.enable_timing(False)
    per = elapsed / steps
    kms = tm["batch_kernel_ms_sum"] / max(1, tm["batches"])
    flops = 2.0 * Q * R * fp_bits  # one GPU, one batch: the 0/1 multiply-adds of the contraction
    tf_kernel = flops / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
    return {
        "seconds": elapsed, "batches": steps, "ms_per_batch": 1e3 * per, "pairs_per_s": Q * total_rows / per,
        "queries_per_s": Q / per,
        "roofline": {"kernel": "batch_mfma_kernel<%d> (MX-FP4 contraction of the packed bits, f32 accumulate: exact)" % W,
                     "bound": "mfma", "achieved": tf_kernel, "peak": MFMA_FP4_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": tf_kernel / MFMA_FP4_PEAK_TFLOPS, "traffic": None, "kernel_ms_avg": kms,
                     "issue": issue_roofline(W, tf_kernel / MFMA_FP4_PEAK_TFLOPS, dense=bool(cutoff)),
                     "whole_step_tflops": flops / per / 1e12,
                     "timed_with_hip_events": tm["batches"],
                     "table_bytes_read_once_per_batch_GBs": R * fp_bits / 8 / per / 1e9},
    }


def in_process(args):
    """`bench.py --gpus N --in-process`: the multi-GPU routes ONE process drives (FingerprintDB::search's fan-out and merge,
    fingerprintdb_cuda.cu:356-380; `gpusimserver --gpus N [--merge rccl]`).  One handle over N devices, the table generated shard
    by shard in HBM; per route the same queries (gsim_db_search_each: single queries, nothing shares a table pass), identical hits
    asserted between the routes:
      host_merge   every shard's kernels write their block into pinned host memory, k-way merge on the host (the default),
                   eight queries enqueued ahead on every shard;
      gsim_comm    blocks stay in HBM, one grouped ncclAllGather over the shards' streams, merge_kernel on the root's device;
      twin         the first shard's rows alone on its device (what one GPU of the N does on its own);
      torch_route  this script's one-process-per-GPU mode (RCCL through torch.distributed) as a child process, its own line.
    On a one-GPU box the test-hooks build presents N logical devices (GSIM_LIB=gpusimilarity_amd/testhooks/libgsim_hip.so
    GSIM_TEST_ALIAS_DEVICES=N): the code path is the product's, the devices are one GPU, the numbers are not a scaling curve."""
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch  # noqa: F401  (first: one HIP runtime in the process, see capi.load)
    from gpusimilarity_amd import capi
    N = args.gpus
    if capi.device_count() < N:
        raise SystemExit("--gpus %d --in-process but only %d device(s) visible" % (N, capi.device_count()))
    aliased = bool(os.environ.get("GSIM_TEST_ALIAS_DEVICES"))
    R = args.rows_per_gpu or (100_000_000 if N == 1 else 125_000_000)
    total = R * N
    kind = {"sparse": capi.SYNTH_SPARSE, "dense": capi.SYNTH_DENSE, "morgan": capi.SYNTH_MORGAN}[args.kind]
    k, W = args.k, args.fp_bits // 32
    qps = max(1, args.queries_per_step)
    steps, warmup = args.steps, args.warmup
    distinct = min((warmup + steps) * qps, 64)
    queries = [synth_row(DB_SEED, kind, query_row(i, total), W) for i in range(distinct)]
    steps_q = [np.ascontiguousarray(np.stack([queries[(s_ * qps + j) % distinct] for j in range(qps)])) for s_ in range(warmup + steps)]

    def run(table, rows):
        bufs = table.make_search_buffers(qps, k)
        for s_ in range(warmup):
            table.search_each_into(steps_q[s_], k, bufs)
        table.enable_timing(True)
        t0 = time.perf_counter()
        for s_ in range(warmup, warmup + steps):
            table.search_each_into(steps_q[s_], k, bufs)
        el = time.perf_counter() - t0
        tm = table.timing()
        table.enable_timing(False)
        nq = steps * qps
        r = {"ms_per_query": 1e3 * el / nq, "fingerprints_per_s": rows * nq / el, "timed_region_s": el, "queries": nq,
             "whole_path_hbm_frac": (rows * (args.fp_bits // 8) / (el / nq)) / (HBM_PEAK_GBS * 1e9 * max(1, table.shard_count())),
             "kernel_ms_avg_first_shard": tm["scan_ms_sum"] / max(1, tm["queries"]), "queries_handed_back": tm["handed_back"]}
        if tm["collectives"]:
            r["gather_us_avg"] = 1e3 * tm["gather_ms_sum"] / tm["collectives"]
            r["merge_us_avg"] = 1e3 * tm["merge_ms_sum"] / tm["collectives"]
            r["collectives"] = tm["collectives"]
        last = (bufs[0][:, :].copy(), bufs[1].copy(), bufs[2].copy())
        return r, last

    table = capi.Table(args.fp_bits).generate(DB_SEED, kind, 0, total, 0, ndevices=N)
    routes = {}
    routes["host_merge"], ref = run(table, total)
    assert int(ref[1][-1]) == min(k, total) and int(ref[2][-1]) == total and ref[0][-1]["score"][0] == 1.0, "self hit missing"
    comm = None
    try:
        comm = capi.Comm(list(range(table.shard_count())))
        table.set_comm(comm)
        routes["gsim_comm"], got = run(table, total)
        routes["gsim_comm"]["transport"] = "loop-back copies (aliased devices)" if aliased and N > 1 else "RCCL ncclAllGather, world %d" % table.shard_count()
        assert got[0].tobytes() == ref[0].tobytes() and (got[1] == ref[1]).all() and (got[2] == ref[2]).all(), "gsim_comm route differs from the host merge"
        routes["gsim_comm"]["identical_to_host_merge"] = True
        table.set_comm(None)
    except capi.GsimError as e:
        routes["gsim_comm"] = {"error": str(e)}
    shards = table.shard_count()
    shard_devices = table.shard_devices()
    table.close()
    if comm is not None:
        comm.close()
    # the first shard's rows alone (row indices 0 .. R-1: the queries that are rows of other shards are fresh fingerprints to it)
    twin = capi.Table(args.fp_bits).generate(DB_SEED, kind, 0, R, 0)
    routes["twin"], _ = run(twin, R)
    twin.close()
    if not args.no_torch_route:
        env = dict(os.environ)
        if aliased and N > 1:
            env["GSIM_BENCH_SHARE_GPU"] = "1"  # (one physical GPU: the ranks share cuda:0 over gloo -- test mode, see main())
            for name in ("GSIM_TEST_ALIAS_DEVICES", "GSIM_LIB"):
                env.pop(name, None)
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(N), "--steps", str(steps), "--warmup", str(warmup), "--queries-per-step", str(qps),
               "--rows-per-gpu", str(R), "--k", str(k), "--fp-bits", str(args.fp_bits), "--kind", args.kind, "--no-configs", "--no-cpu-baseline",
               "--no-server-latency", "--no-pmc"] + (["--force-sharded-path"] if N == 1 else [])
        try:
            p = subprocess.run(cmd, env=env, capture_output=True, timeout=1800)
            line = json.loads(p.stdout.decode().strip().splitlines()[-1])
            routes["torch_route"] = {"ms_per_query": line["ms_per_query"], "fingerprints_per_s": line["value"], "timed_region_s": line["timed_region_s"],
                                     "backend": line["collective"]["backend"], "per_rank": line["collective"].get("per_rank"),
                                     "shared_gpu_test_mode": line["collective"].get("shared_gpu_test_mode", False)}
        except Exception as e:  # a report, never a reason to lose the in-process numbers
            routes["torch_route"] = {"error": repr(e)}
    hm = routes["host_merge"]
    out = {
        "timed_region_s": hm["timed_region_s"],
        "metric": "fingerprints scanned/sec (1024-bit Tanimoto top-1000)", "value": hm["fingerprints_per_s"], "unit": "fingerprints/s",
        "n_gpus": N, "steps": steps, "warmup": warmup, "ms_per_step": hm["ms_per_query"] * qps, "queries_per_step": qps,
        "ms_per_query": hm["ms_per_query"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "%d x %d-bit %s synthetic fingerprints per GPU x %d GPU(s) in ONE process (gsim_db_generate_sharded), Tanimoto top-%d, "
                               "cutoff 0, %d single queries per step; value = the host-merge route (the product's default)" % (R, args.fp_bits, args.kind, N, k, qps),
                   "rows_per_gpu": R, "fp_bits": args.fp_bits, "k": k, "shards": shards,
                   "parallelism": "row shards on %d devices of one process, host merge | gsim_comm all-gather + merge_kernel" % N,
                   "devices": "aliased: %d logical devices on one physical GPU (test hook) -- not a scaling measurement" % N if aliased else "physical"},
        "whole_path_hbm_frac": hm["whole_path_hbm_frac"],
        "roofline": {"kernel": "fused_kernel (per shard)", "bound": "hbm", "achieved": R * (args.fp_bits // 8) / (hm["kernel_ms_avg_first_shard"] * 1e-3) / 1e9 if hm["kernel_ms_avg_first_shard"] else None,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (R * (args.fp_bits // 8) / (hm["kernel_ms_avg_first_shard"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if hm["kernel_ms_avg_first_shard"] else None,
                     "traffic": None, "kernel_ms_avg": hm["kernel_ms_avg_first_shard"]},
        "routes": routes,
        # who is in the job (the one-process route's "ranks" are the handle's shards, one per device) and what crosses xGMI
        "collective": {"backend": routes.get("gsim_comm", {}).get("transport", "gsim_comm failed"), "world": shards,
                       "ranks": [{"rank": i, "device": "hip:%d" % d, "pci_bus_id": None} for i, d in enumerate(shard_devices)],
                       "launcher": "one process, %d devices" % N,
                       "data_path_collective": "gsim_comm route: one grouped ncclAllGather of the shards' result blocks per query, %d B per shard, "
                                               "then merge_kernel on the root's device; host_merge route: none (pinned blocks, k-way merge on the host)"
                                               % capi.result_block_bytes(k),
                       "rccl": capi.rccl_info(), "identical_hits_across_routes": bool(routes.get("gsim_comm", {}).get("identical_to_host_merge"))},
    }
    emit(out, json_fd, "bench_detail_inprocess.json")


def main():
    # stdout carries exactly ONE JSON line: route everything else that may write to fd 1
    # (RCCL prints a version banner from C) to stderr
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--queries-per-step", type=int, default=128,
                    help="single queries per step (a step is one pass of the hot path over one batch of input)")
    ap.add_argument("--rows-per-gpu", type=int, default=0)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--fp-bits", type=int, default=1024)
    ap.add_argument("--kind", choices=["sparse", "dense", "morgan"], default="sparse")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the other single-GPU BASELINE configs (N = 1 only)")
    ap.add_argument("--no-server-latency", action="store_true",
                    help="skip the end-to-end gpusimserver record (scripts/server_latency.py, N = 1 only)")
    ap.add_argument("--batch-queries", type=int, default=0,
                    help="BASELINE configs[4] instead of the headline run: Tversky(0.3, 0.7) batches of this many "
                         "queries per step (use with --fp-bits 2048); a step is one batch")
    ap.add_argument("--cutoff", type=float, default=0.0, help="--batch-queries: similarity cutoff of the batch (0.1 keeps 4.5 %% of the synthetic table per query: "
                                                               "the dense-cutoff variant of the contraction kernel counts them)")
    ap.add_argument("--pmc-child", action="store_true", help="internal: the process rocprofv3 counts (pmc_traffic)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--in-process", action="store_true",
                    help="the one-process multi-GPU routes of the product (gpusimserver --gpus N): ONE handle sharded over --gpus "
                         "devices -- host merge and the C-ABI collective (gsim_comm: RCCL all-gather + merge_kernel) -- each next to the "
                         "first shard's single-GPU twin, and the one-process-per-GPU torch route as a child process")
    ap.add_argument("--no-torch-route", action="store_true", help="--in-process: skip the torch.distributed child run")
    ap.add_argument("--force-sharded-path", action="store_true",
                    help="run the N>1 code path (device result blocks, all-gather, device merge) even at N=1")
    args = ap.parse_args()

    if args.pmc_child:
        pmc_child(args.rows_per_gpu or 100_000_000, args.fp_bits, {"sparse": 0, "dense": 1, "morgan": 2}[args.kind], args.k)
        return
    if args.in_process:
        in_process(args)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))  # no launcher: start the ranks ourselves

    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch  # (first: one HIP runtime in the process, see capi.load)
    import torch.distributed as dist
    from gpusimilarity_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available() or capi.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # test-only: all ranks on cuda:0 over gloo (a one-GPU box cannot give every rank its own device and RCCL
    # refuses two ranks per device); the line says so and its value is not a scaling number
    share = os.environ.get("GSIM_BENCH_SHARE_GPU", "") == "1" and world > 1
    if not share and world > capi.device_count():
        raise SystemExit("--gpus %d but only %d GPU(s) visible" % (world, capi.device_count()))
    device_index = 0 if share else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    backend = None
    if world > 1 or args.force_sharded_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        backend = "gloo" if share else "nccl"
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = {"dev": dev, "world": world, "rank": rank, "backend": backend, "stream": torch.cuda.Stream(device=dev)}

    R = args.rows_per_gpu or (100_000_000 if world == 1 else 125_000_000)
    total_rows = R * world
    kind = {"sparse": capi.SYNTH_SPARSE, "dense": capi.SYNTH_DENSE, "morgan": capi.SYNTH_MORGAN}[args.kind]
    k = args.k
    sharded = world > 1 or args.force_sharded_path

    def make_table(rows, bits, first_row, kind=kind):
        t = capi.Table(bits)
        t.generate(DB_SEED, kind, first_row, rows, device_index)  # this rank's contiguous shard, made in HBM
        t.set_row_base(first_row)
        return t

    # who is in the job: every rank's device, gathered through the process group the data path uses
    me = {"rank": rank, "device": str(dev), "pci_bus_id": getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None),
          "device_name": torch.cuda.get_device_properties(dev).name}
    members = [me]
    if dist.is_initialized():
        members = [None] * world
        dist.all_gather_object(members, me)
    collective = {"backend": ("rccl (torch 'nccl')" if backend == "nccl" else backend), "world": dist.get_world_size() if dist.is_initialized() else 1,
                  "ranks": members, "launcher": "self-spawned" if os.environ.get("GSIM_BENCH_SPAWNED") else
                  ("torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ or "WORLD_SIZE" in os.environ else "single process"),
                  "data_path_collective": ("one all_gather_into_tensor of the per-rank result blocks per query (batch mode: per batch), "
                                           "%d B per rank" % capi.result_block_bytes(k)) if world > 1 else None}
    if share:
        collective["shared_gpu_test_mode"] = True
    if dist.is_initialized():  # (the C-ABI library's own view; the torch route's collectives run on torch's librccl.so either way)
        try:
            collective["rccl"] = dict(capi.rccl_info(), torch_nccl_version=".".join(str(x) for x in torch.cuda.nccl.version()))
        except Exception as e:
            collective["rccl"] = {"error": repr(e)}

    table = make_table(R, args.fp_bits, rank * R)
    if args.batch_queries:
        Q = args.batch_queries
        res = time_batches(ctx, table, total_rows, R, args.fp_bits, kind, k, Q, args.steps, args.warmup, sharded, cutoff=np.float32(args.cutoff))
        out = {
            "metric": "(query, fingerprint) pairs scored/sec (%d-bit Tversky(0.3,0.7), %d-query batches, top-%d)" % (args.fp_bits, Q, k),
            "value": res["pairs_per_s"], "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_batch"], "queries_per_s": res["queries_per_s"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32 bits as MX-FP4 {0,1} operands, f32 accumulate (exact)",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[4] shape: %d x %d-bit rows per GPU x %d GPU(s), Tversky a=0.3 b=0.7, "
                                   "%d-query batch, top-%d" % (R, args.fp_bits, world, Q, k),
                       "rows_per_gpu": R, "fp_bits": args.fp_bits, "k": k, "batch": Q,
                       "parallelism": "row shards, 1 process/GPU" + (", RCCL all_gather of Q result blocks" if world > 1 else "")},
            "roofline": res["roofline"], "collective": collective,
        }
        table.close()
        if rank == 0:
            emit(out, json_fd, "bench_detail_batch.json")
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return

    qps = max(1, args.queries_per_step)
    res, _ss = time_queries(ctx, table, total_rows, R, args.fp_bits, kind, k, args.steps, args.warmup, qps, sharded)
    if sharded:
        # what each rank's own phases cost (HIP events on its stream) and -- the N = 1 twin -- what the same shard answers
        # alone through the synchronous single-GPU path: the 1 -> N curve then decomposes into scan / gather / merge / copy
        table.set_stream(0)  # back to the handle's own stream
        table.set_row_base(rank * R)
        tw, _ = time_queries(dict(ctx, world=1), table, R, R, args.fp_bits, kind, k, 2, 1, min(qps, 64), False, barrier=False, first_row=rank * R)
        mine = {"rank": rank, "phases_per_query": res["phases"],
                "single_gpu_twin": {"rows": R, "ms_per_query": tw["ms_per_query"], "fingerprints_per_s": tw["fingerprints_per_s"],
                                    "kernel_ms_avg": tw["roofline"]["kernel_ms_avg"]}}
        per_rank = [mine]
        if dist.is_initialized() and world > 1:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
        collective["per_rank"] = per_rank
        collective["queries_in_flight"] = 8
    table.close()
    if world == 1 and not sharded and not args.no_pmc and res["roofline"]["kernel"].startswith("fused_kernel"):
        try:
            fetch, write, note = pmc_traffic(R, args.fp_bits, args.kind, k, "fused_kernel")
        except Exception as e:  # a report, never a reason to lose the headline
            fetch, write, note = None, None, "pmc passes failed: %r" % (e,)
        res["roofline"]["traffic"] = fetch
        res["roofline"]["traffic_written"] = write
        res["roofline"]["traffic_note"] = note
        if fetch:
            res["roofline"]["traffic_over_algorithmic"] = fetch / res["roofline"]["algorithmic_bytes_per_launch"]
    headline_cfg = {
        "name": ("BASELINE configs[2]: 100M x 1024-bit, Tanimoto top-1000, 1 MI355X" if world == 1 and R == 100_000_000 and
                 args.fp_bits == 1024 and k == 1000 else "headline"),
        "rows_per_gpu": R, "fp_bits": args.fp_bits, "k": k, "ms_per_query": res["ms_per_query"],
        "ms_per_step": res["ms_per_query"] * qps, "value": res["fingerprints_per_s"], "unit": "fingerprints/s",
        "timed_region_s": res["seconds"], "whole_path_hbm_frac": res["whole_path_hbm_frac"], "roofline": res["roofline"]}
    if res["sync_latency"]:
        headline_cfg.update(res["sync_latency"])
    out = {
        "timed_region_s": res["seconds"],
        "metric": "fingerprints scanned/sec (1024-bit Tanimoto top-1000)",
        "value": res["fingerprints_per_s"], "unit": "fingerprints/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_query"] * qps, "queries_per_step": qps, "ms_per_query": res["ms_per_query"],
        "ms_per_query_note": "mean over the timed region, up to 8 queries enqueued ahead of the one being waited for; sync_ms_* = one "
                             "query at a time, query on the host -> hits on the host, timed inside the library (gsim_db_search_timed; SURVEY 8d's latency)",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {
            "workload": ("%d x %d-bit %s synthetic fingerprints per GPU x %d GPU(s) = %d rows, Tanimoto top-%d, "
                         "cutoff 0, %d single queries per step; %s" % (
                             R, args.fp_bits, args.kind, world, total_rows, k, qps,
                             "BASELINE configs[2] (100M x 1024-bit, 1 MI355X, HBM-bound roofline run)"
                             if world == 1 and R == 100_000_000 else
                             "BASELINE configs[3] shape (1B x 1024-bit over 8 GPUs = 125M rows/GPU), "
                             "per-GPU top-k + RCCL all-gather + merge")),
            "rows_per_gpu": R, "fp_bits": args.fp_bits, "k": k, "metric": "tanimoto",
            "parallelism": "row shards, 1 process/GPU" + (", RCCL all_gather of top-k blocks" if world > 1 else ""),
            "query_execution": ("one query at a time on the GPU; gsim_db_search_each keeps up to 8 of the step's queries "
                                "enqueued ahead (own result block each), no query shares a table pass with another"
                                if not sharded else
                                "one query at a time on every GPU: local search, all-gather, merge, D2H; up to 8 queries' steps are "
                                "enqueued ahead of the one the host waits for, as at N = 1 (which holds 100 M rows per GPU, this line "
                                "%d M: collective.per_rank has each rank's phase times and its shard's single-GPU twin)" % (R // 1_000_000)),
        },
        "whole_path_hbm_frac": res["whole_path_hbm_frac"],
        "roofline": res["roofline"],
        "collective": collective,
    }
    if res["sync_latency"]:
        out.update(res["sync_latency"])
    if world == 1 and not args.no_configs and not sharded:
        # the other single-GPU BASELINE configs, measured in this run, each with its own dominant-kernel roofline
        cfgs = []

        def single_cfg(name, rows, kind_, qps_, note=None, k_=1000):
            t = make_table(rows, 1024, 0, kind_)
            r, _ = time_queries(ctx, t, rows, rows, 1024, kind_, k_, max(args.steps, 20), args.warmup, qps_, False)
            t.close()
            c = {"name": name, "rows_per_gpu": rows, "fp_bits": 1024, "k": k_, "ms_per_query": r["ms_per_query"],
                 "ms_per_step": r["ms_per_query"] * qps_, "queries_per_step": qps_, "value": r["fingerprints_per_s"],
                 "unit": "fingerprints/s", "timed_region_s": r["seconds"], "whole_path_hbm_frac": r["whole_path_hbm_frac"],
                 "roofline": r["roofline"]}
            if r["sync_latency"]:
                c.update(r["sync_latency"])
            if note:
                c["note"] = note
            return c

        cfgs.append(single_cfg("BASELINE configs[1]: 1M x 1024-bit, Tanimoto top-1000, 1 MI355X", 1_000_000, kind, 512,
                               "latency-bound: the 128 MB table streams in 16 us at 8 TB/s; the rest is launch, the threshold "
                               "exchange and the select inside the single launch"))
        cfgs.append(headline_cfg)
        # the same two tables with Morgan-shaped rows (GSIM_SYNTH_MORGAN: popcount 20..53, frequent bits, series of analogs,
        # duplicates -- coarse, tie-heavy scores): what the reference's published numbers are quoted on
        try:
            cfgs.append(single_cfg("configs[1] shape, Morgan-like rows: 1M x 1024-bit, Tanimoto top-1000", 1_000_000,
                                   capi.SYNTH_MORGAN, 512))
            cfgs.append(single_cfg("configs[2] shape, Morgan-like rows: 100M x 1024-bit, Tanimoto top-1000", 100_000_000,
                                   capi.SYNTH_MORGAN, 32))
        except Exception as e:  # never lose the headline over it
            cfgs.append({"name": "Morgan-like tables", "error": repr(e)})
        # large k (not a BASELINE config; the reference sorts every row, any k: fingerprintdb_cuda.cu:280-290): the single launch
        # publishes, two small kernels rank by score bin
        try:
            cfgs.append(single_cfg("large k: 100M x 1024-bit, Tanimoto top-20000 (single launch publishes, bin-ranked emission)", 100_000_000,
                                   kind, 32, k_=20000))
            cfgs.append(single_cfg("large k: 100M x 1024-bit, Tanimoto top-50000 (single launch publishes, bin-ranked emission)", 100_000_000,
                                   kind, 16, k_=50000))
            cfgs.append(single_cfg("large k: 1M x 1024-bit, Tanimoto top-8192", 1_000_000, kind, 128, k_=8192))
        except Exception as e:
            cfgs.append({"name": "large k", "error": repr(e)})
        try:
            t4 = make_table(125_000_000, 2048, 0)
            r4 = time_batches(ctx, t4, 125_000_000, 125_000_000, 2048, kind, 1000, 256, 8, 2, False)
            cfgs.append({"name": "BASELINE configs[4], per-GPU shape: 125M x 2048-bit, Tversky(0.3,0.7), 256-query batch, top-1000",
                         "rows_per_gpu": 125_000_000, "fp_bits": 2048, "k": 1000, "batch": 256,
                         "ms_per_step": r4["ms_per_batch"], "value": r4["pairs_per_s"], "unit": "pairs/s",
                         "queries_per_s": r4["queries_per_s"], "timed_region_s": r4["seconds"], "roofline": r4["roofline"]})
            # the same batches with a cutoff that keeps 4.5 % of the table per query (5.7 M rows): `approx` is their exact
            # count -- counted from the accumulators by the dense-cutoff variant of the contraction kernel
            r5 = time_batches(ctx, t4, 125_000_000, 125_000_000, 2048, kind, 1000, 256, 6, 2, False, cutoff=np.float32(0.1))
            dense = t4.timing()["batches_dense_cutoff"]
            t4.close()
            cfgs.append({"name": "configs[4] shape with cutoff 0.1 (5.7 M rows at or above it per query, counted exactly)",
                         "rows_per_gpu": 125_000_000, "fp_bits": 2048, "k": 1000, "batch": 256, "cutoff": 0.1,
                         "ms_per_step": r5["ms_per_batch"], "value": r5["pairs_per_s"], "unit": "pairs/s",
                         "batches_counted_on_the_matrix_cores": int(dense),
                         "queries_per_s": r5["queries_per_s"], "timed_region_s": r5["seconds"], "roofline": r5["roofline"]})
        except Exception as e:  # never lose the headline over it
            cfgs.append({"name": "BASELINE configs[4] per-GPU shape", "error": repr(e)})
        # other row widths (the reference's --gpu_bitcount folds 1024-bit rows to 512 / 256 / 128 bits; MACCS-sized keys
        # are 166 -> 192 bits): whole-query fraction of the HBM peak per width at EQUAL TABLE BYTES (6.4 GB each, top-1000) -- at
        # equal rows a table of narrow rows is simply smaller and shows the query's fixed ~25 us instead of the width's rate
        widths = []
        for bits in (128, 256, 512, 2048, 160, 192, 896, 1152):
            rows = int(6_400_000_000 // (bits // 8))
            try:
                tw = capi.Table(bits)
                tw.generate(DB_SEED, kind, 0, rows, device_index)
                rw, _ = time_queries(ctx, tw, rows, rows, bits, kind, 1000, 4, 1, 32, False)
                tw.close()
                widths.append({"fp_bits": bits, "rows": rows, "table_bytes": rows * (bits // 8), "ms_per_query": rw["ms_per_query"], "whole_path_hbm_frac": rw["whole_path_hbm_frac"],
                               "kernel": rw["roofline"]["kernel"], "kernel_ms_avg": rw["roofline"]["kernel_ms_avg"], "kernel_hbm_frac": rw["roofline"]["frac"],
                               "queries_handed_back": rw["roofline"]["queries_handed_back"],
                               "sync_ms_median": rw["sync_latency"]["sync_ms_median"] if rw["sync_latency"] else None})
            except Exception as e:  # never lose the headline over it
                widths.append({"fp_bits": bits, "error": repr(e)})
        out["widths"] = widths
        # north_star's target sentence names THIS table: 1 B x 1024-bit rows (128 GB of the 288) on one GPU, the single launch
        try:
            t1b = make_table(1_000_000_000, 1024, 0)
            r1b, _ = time_queries(ctx, t1b, 1_000_000_000, 1_000_000_000, 1024, kind, 1000, 3, 1, 4, False)
            t1b.close()
            c1b = {"name": "1B x 1024-bit on ONE MI355X (north_star's target table), Tanimoto top-1000, single launch", "rows_per_gpu": 1_000_000_000,
                   "fp_bits": 1024, "k": 1000, "ms_per_query": r1b["ms_per_query"], "queries_per_step": 4, "ms_per_step": 4 * r1b["ms_per_query"],
                   "value": r1b["fingerprints_per_s"], "unit": "fingerprints/s", "timed_region_s": r1b["seconds"],
                   "whole_path_hbm_frac": r1b["whole_path_hbm_frac"], "roofline": r1b["roofline"]}
            if r1b["sync_latency"]:
                c1b.update(r1b["sync_latency"])
            cfgs.append(c1b)
        except Exception as e:  # never lose the headline over it
            cfgs.append({"name": "1B x 1024-bit on one GPU", "error": repr(e)})
        out["configs"] = cfgs
        out["configs_note"] = ("configs[0] (small.fsim, CPU path) is timed under cpu_baseline.parts; configs[3] (1B rows over 8 GPUs) "
                               "is this script at --gpus 8")
    if world == 1 and not sharded and not args.no_server_latency and not args.no_configs:
        # the reference's own published metric: server-side search latency through the socket backend, at two of its
        # slide-12/13 table sizes (ChEMBL 1 618 358 rows, Enamine 56 667 620 rows), Morgan-shaped rows, k = 20 and 1000
        try:
            env = dict(os.environ, SL_REQUESTS="30")
            p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "server_latency.py"), "1618358", "56667620"],
                               env=env, capture_output=True, timeout=300)
            out["server_latency"] = json.loads(p.stdout.decode().strip().splitlines()[-1]) if p.returncode == 0 else \
                {"error": p.stderr.decode("utf-8", "replace")[-500:]}
        except Exception as e:  # a report, never a reason to lose the headline
            out["server_latency"] = {"error": repr(e)}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.fp_bits, k, kind)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "fingerprints/s", "cores": os.cpu_count(),
                                       "kind": "port", "sample": "failed: %r" % (e,)}
        # LAST in the line, and short: the driver keeps the tail of stdout (8 KB), and round 4's record lost configs[1] to it
        if "configs" in out:
            short = []
            for c in out["configs"]:
                if "error" in c:
                    short.append({"name": c["name"][:60], "error": c["error"][:80]})
                    continue
                rf = c.get("roofline") or {}
                e = {"name": c["name"][:72], "ms": round(c.get("ms_per_query", c.get("ms_per_step", 0.0)), 5), "per": "query" if "ms_per_query" in c else "batch",
                     "kernel_ms": round(rf.get("kernel_ms_avg") or 0.0, 5), "frac": round(rf.get("frac") or 0.0, 4), "bound": rf.get("bound")}
                if c.get("sync_ms_median") is not None:
                    e["sync_ms_median"] = round(c["sync_ms_median"], 5)
                if rf.get("issue"):
                    e["issue_frac"] = round(rf["issue"]["issue_frac"], 3)
                if rf.get("queries_handed_back") is not None:
                    e["handed_back"] = rf["queries_handed_back"]
                short.append(e)
            out["summary"] = {"configs": short, "traffic_over_algorithmic": out["roofline"].get("traffic_over_algorithmic"),
                              "cpu_baseline_fp_per_s": (out.get("cpu_baseline") or {}).get("value")}
        emit(out, json_fd)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
