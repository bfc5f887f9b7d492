"""BASELINE configs[3] and configs[4] at FULL size on ONE MI355X (288 GB hold them):

* configs[3]: 1 B x 1024-bit rows.  One 1 B-row handle answers the queries; then the table is
  rebuilt as EIGHT handles of 125 M rows (the eight GPUs' shards: gsim_db_generate with
  first_row = g * 125 M, gsim_db_set_row_base), every handle leaves its top-k block in device
  memory (gsim_db_search_device) and gsim_merge_device merges the eight blocks -- what eight ranks
  and one all-gather produce.  The merged block must equal the single handle's answer bit for bit.
* configs[4]: 8 x 100 M x 2048-bit rows, Tversky(0.3, 0.7), 256-query batches:
  gsim_db_search_batch_device on every handle, gsim_merge_device_batch, against one 800 M-row handle.
* bench.py --gpus 8 with eight self-spawned ranks sharing cuda:0 (125 M rows each, gloo).

Round 4: the billion rows are ALSO generated on the host (the oracle's generator on all host
threads, 128 GB) and every query's full top-k -- rows, score bits, popcounts, approx -- is compared
with the oracle's scan of all 10^9 rows; one shard's own block (row_base != 0, the per-GPU shape of
configs[3]) is compared with the oracle's scan of that shard's rows.  A host without ~140 GB of
free memory falls back to the properties the earlier rounds checked (equality of the two routes,
returned rows regenerated and rescored by the oracle, self hit / canonical order / prefix).
(fingerprintdb_cuda.cu:356-380 is the reference's fan-out + merge; its slices are ~8.4 M rows,
:111-126.)
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from gpusimilarity_amd import capi

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x5EED0001
GIB = 1 << 30


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def canonical_sorted(h):
    s, r = h["score"], h["row"].astype(np.int64)
    return bool(np.all((s[:-1] > s[1:]) | ((s[:-1] == s[1:]) & (r[:-1] < r[1:]))))


def rescored_by_oracle(hits, q, kind, W, metric=0, alpha=0.0, beta=0.0):
    """Every given hit's row is regenerated on the CPU; popcounts and the score's bit pattern must match."""
    a = int(np.unpackbits(q.view(np.uint8)).sum())
    L = O.lib()
    for h in hits:
        row = O.synth_rows(SEED, kind, int(h["row"]), 1, W)[0]
        c = int(np.unpackbits((row & q).view(np.uint8)).sum())
        b = int(np.unpackbits(row.view(np.uint8)).sum())
        assert (c, b) == (int(h["common"]), int(h["popc_db"]))
        assert bits(np.float32(L.gso_score_one(metric, alpha, beta, a, b, c))) == bits(h["score"])


def host_free_bytes():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 0


def assert_block_equal(got, want, ctx):
    assert len(got) == len(want), "%s: %d hits, oracle %d" % (ctx, len(got), len(want))
    assert (got["row"] == want["row"]).all(), ctx
    assert (bits(got["score"]) == bits(want["score"])).all(), ctx
    assert (got["common"] == want["common"]).all() and (got["popc_db"] == want["popc_db"]).all(), ctx


def need_free(nbytes):
    free = capi.device_free_bytes(0)
    if free < nbytes:
        pytest.skip("needs %.0f GiB of free HBM, %.0f GiB free" % (nbytes / GIB, free / GIB))


def oracle_level_or_skip(level, host_bytes, test):
    """The two parametrisations of a full-size test: `oracle_all_rows` needs the table on the host as well (the oracle
    scans every row), `properties_only` is what remains when the host has no room.  Exactly one of them runs; the other
    one is SKIPPED with the reason, so that the record shows which parity level the box reached."""
    have = host_free_bytes()
    enough = have > host_bytes
    if level == "oracle_all_rows" and not enough:
        pytest.skip("host has %.0f GiB free, the oracle's copy of the table needs %.0f GiB: see [properties_only]" % (have / GIB, host_bytes / GIB))
    if level == "properties_only" and enough:
        pytest.skip("covered by [oracle_all_rows] (the host has room for the oracle's copy of the table)")
    from conftest import record_parity
    record_parity(test, level, "host free %.0f GiB" % (have / GIB))
    return level == "oracle_all_rows"


@pytest.mark.parametrize("level", ["oracle_all_rows", "properties_only"])
@pytest.mark.parametrize("kind", [capi.SYNTH_SPARSE, capi.SYNTH_MORGAN])
def test_configs3_one_billion_rows_eight_shards_equal_one_handle(kind, level):
    import torch
    W, k, G, per = 32, 1000, 8, 125_000_000
    total = G * per
    need_free(total * W * 4 + 36 * GIB)
    with_oracle = oracle_level_or_skip(level, total * W * 4 + 16 * GIB, "configs[3] 1 B x 1024-bit, kind %d" % kind)
    queries = [capi.synth_row(SEED, kind, O.query_row(i, total), W * 32) for i in range(3)]
    queries.append(O.synth_rows(0x5EED0002, 0, 77, 1, W)[0])  # a fresh fingerprint, not a row of the table
    cases = [(q, kk, cut) for q in queries for kk, cut in ((k, 0.0),)] + [(queries[0], 10, 0.0), (queries[1], k, 0.3)]
    # ---- the oracle's copy of the table and its answers over ALL rows
    nt = os.cpu_count() or 1
    host = None
    if with_oracle:
        host = O.synth_rows_mt(SEED, kind, 0, total, W, nt)
        oracle = [O.search(q, host, kk, cut, nthreads=nt) for q, kk, cut in cases]
    # ---- one handle holds the whole table
    whole = capi.Table(W * 32).generate(SEED, kind, 0, total, 0)
    want = []
    for i, (q, kk, cut) in enumerate(cases):
        h, ap = whole.search(q, kk, cut)
        want.append((h[0], int(ap[0])))
        if host is not None:
            assert int(ap[0]) == oracle[i][1], "case %d" % i
            assert_block_equal(h[0], oracle[i][0], "1 B rows, one handle, case %d" % i)
    h0, ap0 = want[0]
    assert ap0 == total and len(h0) == k
    # self hit (a Morgan-shaped table holds exact duplicates: the query's row is ONE of the rows scoring 1.0)
    assert h0["score"][0] == np.float32(1.0) and O.query_row(0, total) in h0["row"][h0["score"] == np.float32(1.0)]
    assert canonical_sorted(h0)
    rescored_by_oracle(np.concatenate([h0[:48], h0[-48:]]), queries[0], kind, W)
    hp, _ = whole.search(queries[0], 100, 0.0)
    assert (hp[0]["row"] == h0["row"][:100]).all()  # prefix property
    whole.close()
    # ---- the same rows as eight shards, each with its own handle, state and scratch
    stream = torch.cuda.Stream()
    shards = []
    for g in range(G):
        t = capi.Table(W * 32).generate(SEED, kind, g * per, per, 0)
        t.set_row_base(g * per)
        t.set_stream(stream.cuda_stream)
        shards.append(t)
    for (q, kk, cut), (wh, wap) in zip(cases, want):
        blk = capi.result_block_bytes(kk)
        gathered = torch.zeros(blk * G, dtype=torch.uint8, device="cuda")
        merged = torch.zeros(blk, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            for g, t in enumerate(shards):
                t.search_device(q, kk, gathered.data_ptr() + g * blk, cutoff=cut)
            capi.merge_device(0, stream.cuda_stream, gathered.data_ptr(), G, blk, kk, merged.data_ptr())
        stream.synchronize()
        hits, approx, _ = capi.parse_result_block(merged.cpu().numpy().tobytes(), kk)
        assert approx == wap
        assert len(hits) == len(wh) and (hits["row"] == wh["row"]).all() and (bits(hits["score"]) == bits(wh["score"])).all()
        assert (hits["common"] == wh["common"]).all() and (hits["popc_db"] == wh["popc_db"]).all()
        # every shard's own block: rows inside its range, canonical order
        raw = gathered.cpu().numpy().tobytes()
        for g in range(G):
            hg, apg, _ = capi.parse_result_block(raw[g * blk:(g + 1) * blk], kk)
            assert len(hg) == 0 or (int(hg["row"].min()) >= g * per and int(hg["row"].max()) < (g + 1) * per)
            assert canonical_sorted(hg)
            if host is not None and g in (3, 7):  # a shard's own answer (row_base != 0) against the oracle on ITS rows
                ws, wsap = O.search(q, host[g * per:(g + 1) * per], kk, cut, row_base=g * per, nthreads=nt)
                assert apg == wsap
                assert_block_equal(hg, ws, "shard %d of 8" % g)
    for t in shards:
        t.close()
    del host


@pytest.mark.parametrize("level", ["oracle_all_rows", "properties_only"])
def test_configs4_batches_over_eight_shards_equal_one_handle(level):
    import torch
    W, k, G, per, Q = 64, 1000, 8, 100_000_000, 256
    total = G * per
    need_free(total * W * 4 + 40 * GIB)
    with_oracle = oracle_level_or_skip(level, total * W * 4 + 16 * GIB, "configs[4] 800 M x 2048-bit, 256-query batch")
    kind = capi.SYNTH_SPARSE
    kw = dict(metric=capi.METRIC_TVERSKY, alpha=np.float32(0.3), beta=np.float32(0.7))
    qs = np.ascontiguousarray(np.stack([capi.synth_row(SEED, kind, O.query_row(i, total), W * 32) for i in range(Q)]))
    whole = capi.Table(W * 32).generate(SEED, kind, 0, total, 0)
    wh, wap = whole.search(qs, k, 0.0, **kw)
    assert all(int(a) == total for a in wap)
    for i in (0, 100, 255):
        assert int(wh[i]["row"][0]) == O.query_row(i, total) and wh[i]["score"][0] == np.float32(1.0)
        assert canonical_sorted(wh[i])
    rescored_by_oracle(np.concatenate([wh[7][:24], wh[7][-24:]]), qs[7], kind, W, 1, np.float32(0.3), np.float32(0.7))
    # three of the batch's queries against the oracle's scan of ALL 800 M rows (205 GB on the host)
    if with_oracle:
        nt = os.cpu_count() or 1
        host = O.synth_rows_mt(SEED, kind, 0, total, W, nt)
        for i in (0, 131, 255):
            ws, wsap = O.search(qs[i], host, k, 0.0, nthreads=nt, **kw)
            assert int(wap[i]) == wsap
            assert_block_equal(wh[i], ws, "800 M x 2048-bit, batch query %d" % i)
        del host
    # the shared pass against the single-query path on the same handle
    one, _ = whole.search(qs[200], k, 0.0, **kw)
    assert (one[0]["row"] == wh[200]["row"]).all() and (bits(one[0]["score"]) == bits(wh[200]["score"])).all()
    whole.close()
    stream = torch.cuda.Stream()
    blk = capi.result_block_bytes(k)
    gathered = torch.zeros(blk * Q * G, dtype=torch.uint8, device="cuda")
    merged = torch.zeros(blk * Q, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    shards = []
    for g in range(G):
        t = capi.Table(W * 32).generate(SEED, kind, g * per, per, 0)
        t.set_row_base(g * per)
        t.set_stream(stream.cuda_stream)
        shards.append(t)
    with torch.cuda.stream(stream):
        for g, t in enumerate(shards):  # rank-major: rank g's Q blocks
            t.search_batch_device(qs, k, gathered.data_ptr() + g * Q * blk, **kw)
        capi.merge_device_batch(0, stream.cuda_stream, gathered.data_ptr(), G, Q, blk, k, merged.data_ptr())
    stream.synchronize()
    raw = merged.cpu().numpy().tobytes()
    for i in range(Q):
        hits, approx, _ = capi.parse_result_block(raw[i * blk:(i + 1) * blk], k)
        assert approx == total
        assert (hits["row"] == wh[i]["row"]).all() and (bits(hits["score"]) == bits(wh[i]["score"])).all(), i
        assert (hits["common"] == wh[i]["common"]).all() and (hits["popc_db"] == wh[i]["popc_db"]).all()
    for t in shards:
        t.close()


def test_bench_eight_ranks_sharing_the_gpu():
    """`python bench.py --gpus 8`: eight self-spawned ranks x 125 M rows on cuda:0 (gloo gather, test mode) --
    the N = 8 code path of configs[3] end to end, every step checked by bench.py's own self-hit assertion."""
    need_free(8 * 125_000_000 * 128 + 24 * GIB)
    env = dict(os.environ, GSIM_BENCH_SHARE_GPU="1")
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                          "--queries-per-step", "4"], env=env, capture_output=True, timeout=1800)
    assert out.returncode == 0, out.stderr.decode("utf-8", "replace")[-4000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["collective"]["world"] == 8
    assert rec["config"]["rows_per_gpu"] == 125_000_000 and rec["value"] > 0
    assert sorted(r["rank"] for r in rec["collective"]["ranks"]) == list(range(8))
