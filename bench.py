#!/usr/bin/env python3
"""Benchmark of the hot path: Tanimoto top-1000 scan of synthetic 1024-bit tables.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one query: the packed table resident in HBM is scanned once, the exact
top-k is selected, (N > 1: per-GPU top-k blocks are all-gathered over RCCL/xGMI and
merged), and the k hits land in host memory.  Weak scaling: every rank holds
--rows-per-gpu rows (default 100 M at N = 1 = BASELINE.json configs[2], the
HBM-bound roofline run; 125 M at N > 1 so that 8 GPUs hold the 1 B-row table of
configs[3]).  value = rows of the whole table * steps / max-over-ranks time.

Prints ONE JSON line (rank 0).  The `roofline` object is the scan kernel's
algorithmic bytes (128 B per fingerprint per pass) over its HIP-event duration on
the stream it runs on; `cpu_baseline` times the reference's host functor path
(oracle/_ref when present, else the oracle port) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (first: one HIP runtime in the process, see capi.load)
import torch.distributed as dist  # noqa: E402

from gpusimilarity_amd import capi  # noqa: E402
from gpusimilarity_amd.sharded import ShardedBatchSearch, ShardedSearch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
DB_SEED = 0x5EED0001
GOLDEN = 0x9E3779B97F4A7C15
M64 = (1 << 64) - 1


def _splitmix64(x):
    z = (x + GOLDEN) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def synth_row(seed, kind, row, W):
    """Host twin of the device generator for one row -- used to make the
    query fingerprints (queries are rows of the table: guaranteed score-1.0 self hit)."""
    out = np.zeros(W, dtype=np.uint32)
    for j in range(W):
        ctr = row * W + j
        if kind == capi.SYNTH_DENSE:
            out[j] = _splitmix64((seed + ctr * GOLDEN) & M64) & 0xFFFFFFFF
        else:
            h0 = _splitmix64((seed + (2 * ctr) * GOLDEN) & M64)
            h1 = _splitmix64((seed + (2 * ctr + 1) * GOLDEN) & M64)
            out[j] = (h0 & 0xFFFFFFFF) & (h0 >> 32) & (h1 & 0xFFFFFFFF) & (h1 >> 32)
    return out


def query_row(q, nrows):
    return _splitmix64((0xC0FFEE + q) & M64) % nrows


def cpu_baseline(fp_bits, k, kind, budget_s=12.0):
    """The reference's host functor path on this box's cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    W = fp_bits // 32
    cores = os.cpu_count() or 1
    n = 8_000_000  # 1 GB of fingerprints: larger than the host's last-level caches
    db = O.synth_rows(DB_SEED, kind, 0, n, W)
    q = db[query_row(0, n)]
    use_ref = O.ref_lib() is not None
    if use_ref:
        tab = O.RefTable(db)
        run = lambda: tab.scan(q, nthreads=cores)  # noqa: E731
        what = "reference TanimotoFunctorCPU (calculation_functors.cpp compiled in place, oracle/_ref), scoring only"
    else:
        run = lambda: O.search(q, db, k, 0.0, nthreads=cores)  # noqa: E731
        what = "oracle port gso_search (scan + heap top-%d)" % k
    run()
    reps, t0 = 0, time.perf_counter()
    while True:
        run()
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 1000:
            break
    return {"value": n * reps / el, "unit": "fingerprints/s", "cores": cores,
            "kind": "reference" if use_ref else "port",
            "sample": "%d queries over a %d-row x %d-bit synthetic slice, %d threads, %.1f s; %s" %
                      (reps, n, fp_bits, cores, el, what)}


MFMA_FP4_PEAK_TFLOPS = 10000.0  # dense MX-FP4, MI355X_MICROARCH.md (measured 9099)


def run_batches(args, table, stream, dev, world, rank, R, total_rows, kind, W, k, sharded_path, json_fd):
    """BASELINE configs[4]: Tversky(0.3, 0.7), Q-query batches, top-k per query; a step = one batch.
    Rows shard over the ranks, every rank scores all Q queries against its shard (the matrix-core
    pass), ONE all-gather of Q result blocks per step, one merge launch."""
    Q = args.batch_queries
    kw = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
    nb = args.warmup + args.steps
    batches = [np.ascontiguousarray(np.stack([synth_row(DB_SEED, kind, query_row(b * Q + i, total_rows), W)
                                              for i in range(Q)]), dtype=np.uint32)
               for b in range(min(nb, 4))]  # a few distinct batches, cycled
    sb = None
    if sharded_path:
        sb = ShardedBatchSearch(table, k, Q, dev, stream=stream, search_kwargs=kw)
    last = {}
    bufs = table.make_search_buffers(Q, k)  # caller-owned outputs of the synchronous C-ABI call

    def one_batch(qs):
        if not sharded_path:
            table.search_into(qs, k, bufs, 0.0, **kw)
            last["hits"] = [bufs[0][0, :bufs[1][0]]]
            return
        sb.enqueue(qs)
        sb.synchronize()

    for b in range(args.warmup):
        one_batch(batches[b % len(batches)])
    if args.warmup:
        b = (args.warmup - 1) % len(batches)
        first = sb.results()[0][0] if sharded_path else last["hits"][0]
        assert int(first["row"][0]) == query_row(b * Q, total_rows) and first["score"][0] == 1.0, "self hit missing"
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    for b in range(args.warmup, nb):
        one_batch(batches[b % len(batches)])
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        steps = args.steps
        per = elapsed / steps
        pairs = Q * total_rows / per
        tflops = 2.0 * Q * R * args.fp_bits / per / 1e12  # one GPU: 0/1 multiply-adds of the contraction
        out = {
            "metric": "(query, fingerprint) pairs scored/sec (%d-bit Tversky(0.3,0.7), %d-query batches, top-%d)" % (
                args.fp_bits, Q, k),
            "value": pairs, "unit": "pairs/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * per, "queries_per_s": Q / per, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 bits as MX-FP4 {0,1} operands, f32 accumulate (exact)",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[4] shape: %d x %d-bit rows per GPU x %d GPU(s), Tversky a=0.3 b=0.7, "
                                   "%d-query batch, top-%d" % (R, args.fp_bits, world, Q, k),
                       "rows_per_gpu": R, "fp_bits": args.fp_bits, "k": k, "batch": Q,
                       "parallelism": "row shards, 1 process/GPU" + (", RCCL all_gather of Q result blocks" if world > 1 else "")},
            "roofline": {"kernel": "batch_mfma_kernel", "bound": "mfma",
                         "achieved": tflops, "peak": MFMA_FP4_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / MFMA_FP4_PEAK_TFLOPS,
                         "traffic": None,
                         "note": "whole step (sample passes, contraction, compaction, select, host) per GPU; "
                                 "table bytes read once per batch: %.1f GB/s effective" % (R * args.fp_bits / 8 / per / 1e9)},
        }
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main():
    # stdout carries exactly ONE JSON line: route everything else that may write to fd 1
    # (RCCL prints a version banner from C) to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows-per-gpu", type=int, default=0)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--fp-bits", type=int, default=1024)
    ap.add_argument("--kind", choices=["sparse", "dense"], default="sparse")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch-queries", type=int, default=0,
                    help="BASELINE configs[4] instead of the headline run: Tversky(0.3, 0.7) batches of this many "
                         "queries per step (use with --fp-bits 2048); a step is one batch")
    ap.add_argument("--force-sharded-path", action="store_true",
                    help="run the N>1 code path (device result blocks, all-gather, device merge) even at N=1")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available() or capi.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_sharded_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    R = args.rows_per_gpu or (100_000_000 if world == 1 else 125_000_000)
    total_rows = R * world
    kind = capi.SYNTH_SPARSE if args.kind == "sparse" else capi.SYNTH_DENSE
    W = args.fp_bits // 32
    k = args.k

    table = capi.Table(args.fp_bits)
    table.generate(DB_SEED, kind, rank * R, R, local_rank)  # this rank's contiguous shard, made in HBM
    table.set_row_base(rank * R)
    stream = torch.cuda.Stream(device=dev)
    table.set_stream(stream.cuda_stream)

    sharded_path = world > 1 or args.force_sharded_path
    if args.batch_queries:
        run_batches(args, table, stream, dev, world, rank, R, total_rows, kind, W, k, sharded_path, json_fd)
        return
    nq = args.warmup + args.steps
    queries = [synth_row(DB_SEED, kind, query_row(i, total_rows), W) for i in range(nq)]
    bufs = table.make_search_buffers(1, k)
    ss = None
    if sharded_path:
        ss = ShardedSearch(table, k, dev, stream=stream)  # local result block stays in HBM

    def one_query(q):
        if not sharded_path:
            # single GPU: the C ABI's synchronous entry point (FingerprintDB::search): scan ->
            # compact -> select, the select kernel writes the hits into pinned host memory,
            # the call returns when they are there
            table.search_into(q, k, bufs)
            return
        ss.enqueue(q)  # local top-k -> RCCL all-gather (k*12+16 B per GPU) -> rank merge -> D2H
        ss.synchronize()  # the query is done when its k hits are in host memory

    def last_result():
        if not sharded_path:
            return bufs[0][0, :bufs[1][0]], int(bufs[2][0])
        hits, approx, _ = ss.result()
        return hits, approx

    for i in range(args.warmup):
        one_query(queries[i])
    # sanity on the last warm-up query: the self hit leads the result
    if args.warmup:
        hits, approx = last_result()
        want_row = query_row(args.warmup - 1, total_rows)
        assert len(hits) == min(k, total_rows) and int(hits["row"][0]) == want_row and hits["score"][0] == 1.0, \
            "self hit missing: %r" % (hits[:3],)
        assert approx == total_rows

    table.enable_timing(True)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, nq):
        one_query(queries[i])
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tm = table.timing()

    # HBM traffic of the scan kernel from the committed PMC pass (rocprofv3 cannot run inside bench.py):
    # measured bytes / algorithmic bytes, applied to this run's algorithmic bytes
    traffic_ratio, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")) as f:
            pm = json.load(f)
        if pm["fp_bits"] == args.fp_bits:
            traffic_ratio, traffic_src = pm["ratio"], pm["source"]
    except Exception:
        pass

    if rank == 0:
        steps = args.steps
        ms = 1e3 * elapsed / steps
        scan_ms = tm["scan_ms_sum"] / max(1, tm["queries"])
        algo_bytes = R * (args.fp_bits // 8)  # per scan-kernel launch on one GPU
        achieved = algo_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        out = {
            "metric": "fingerprints scanned/sec (1024-bit Tanimoto top-1000)",
            "value": total_rows * steps / elapsed,
            "unit": "fingerprints/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": ms, "ms_per_query": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {
                "workload": ("%d x %d-bit %s synthetic fingerprints per GPU x %d GPU(s) = %d rows, Tanimoto top-%d, "
                             "cutoff 0; %s" % (R, args.fp_bits, args.kind, world, total_rows, k,
                                               "BASELINE configs[2] (100M x 1024-bit, 1 MI355X, HBM-bound roofline run)"
                                               if world == 1 and R == 100_000_000 else
                                               "BASELINE configs[3] shape (1B x 1024-bit over 8 GPUs = 125M rows/GPU), "
                                               "per-GPU top-k + RCCL all-gather + merge")),
                "rows_per_gpu": R, "fp_bits": args.fp_bits, "k": k, "metric": "tanimoto",
                "parallelism": "row shards, 1 process/GPU" + (", RCCL all_gather of top-k blocks" if world > 1 else ""),
            },
            "whole_path_hbm_frac": (total_rows * (args.fp_bits // 8) / (elapsed / steps)) / (HBM_PEAK_GBS * 1e9 * world),
            "roofline": {
                "kernel": "scan_kernel<8,8>" if args.fp_bits == 1024 else "scan_kernel",
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": (traffic_ratio * algo_bytes) if traffic_ratio else None, "traffic_unit": "bytes/launch",
                "traffic_source": traffic_src,
                "scan_ms_avg": scan_ms, "select_ms_avg": tm["select_ms_sum"] / max(1, tm["queries"]),
                "algorithmic_bytes_per_launch": algo_bytes,
                "candidates_per_query": tm["candidates_sum"] / max(1, tm["queries"]),
                "finalists_per_query": tm["finalists_sum"] / max(1, tm["queries"]),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            table.close()
            try:
                out["cpu_baseline"] = cpu_baseline(args.fp_bits, k, kind)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "fingerprints/s", "cores": os.cpu_count(),
                                       "kind": "port", "sample": "failed: %r" % (e,)}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
