#!/usr/bin/env python3
"""End-to-end latency of the `gpusimserver` backend, the reference's own published metric: its slides quote the server's
"Search completed, time elapsed" (gpusim.cpp:420-429, the timer around searchDatabases) for Tanimoto searches returning
20 / 1000 results on tables of 1.6 M ... 113 M fingerprints.

This script starts bin/gpusimserver on a synthetic Morgan-shaped table generated in HBM ("synthetic:<rows>:morgan"),
replays requests in the reference's wire format (Appendix B of SURVEY.md: what python/gpusim_search.py:36-47 writes)
over /tmp/gpusimilarity and reports, per table size and k:
  * client_ms: send of the request frame -> last byte of the reply (socket, QDataStream decode, search, SMILES / ID
    gather and ";:;" folding, reply encode, socket) -- what a caller of the backend sees;
  * server_ms: the server's own "time elapsed" line (the reference's published figure measures exactly this).
Every reply is checked: the query is a row of the table, so the first hit scores 1.0 and the row's own SMILES is
among the hits that do.

    python scripts/server_latency.py [rows ...]   (env: SL_REQUESTS, SL_KIND, SL_GPUS)
Prints one JSON document.
"""
import json
import os
import socket
import struct
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SOCK = "/tmp/gpusimilarity"
SEED = 0x5EED0001
KINDS = {"sparse": 0, "dense": 1, "morgan": 2}


def cstr(b):
    return struct.pack(">I", len(b) + 1) + b + b"\0"


def request(dbname, dbkey, request_num, k, cutoff, fp_bytes):
    return (struct.pack(">i", 1) + cstr(dbname) + cstr(dbkey) + struct.pack(">iid", request_num, k, cutoff) +
            struct.pack(">I", len(fp_bytes)) + fp_bytes)


def read_reply(s):
    buf = b""

    def need(n):
        nonlocal buf
        while len(buf) < n:
            chunk = s.recv(1 << 20)
            if not chunk:
                raise RuntimeError("server closed the connection")
            buf += chunk

    need(16)
    req, n, approx = struct.unpack(">iiQ", buf[:16])
    off = 16
    strings = []
    for _ in range(2 * n):
        need(off + 4)
        ln = struct.unpack(">I", buf[off:off + 4])[0]
        need(off + 4 + ln)
        strings.append(buf[off + 4:off + 4 + ln - 1])
        off += 4 + ln
    need(off + 8 * n)
    scores = struct.unpack(">%dd" % n, buf[off:off + 8 * n])
    return req, approx, strings[:n], strings[n:], scores


def measure(rows, kind, ks, nreq, gpus):
    from bench import query_row
    from gpusimilarity_amd import capi
    if os.path.exists(SOCK):
        os.unlink(SOCK)
    log = open("/tmp/gpusimserver_latency.log", "w+")
    t0 = time.perf_counter()
    cmd = [os.path.join(ROOT, "gpusimilarity_amd", "bin", "gpusimserver")] + (["--gpus", str(gpus)] if gpus != 1 else []) + \
        ["synthetic:%d:%s" % (rows, kind)]
    p = subprocess.Popen(cmd, stderr=log)
    try:
        while True:
            log.seek(0)
            if "Ready for searches." in log.read():
                break
            if p.poll() is not None:
                log.seek(0)
                raise RuntimeError("gpusimserver exited: " + log.read()[-2000:])
            time.sleep(0.05)
        startup_s = time.perf_counter() - t0
        out = []
        queries = [capi.synth_row(SEED, KINDS[kind], query_row(i, rows), 1024).tobytes() for i in range(16)]
        for k in ks:
            client, n_hits = [], 0
            s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            s.connect(SOCK)
            log.seek(0, 2)
            mark = log.tell()
            for i in range(5 + nreq):
                frame = request(b"synthetic", b"pass", 1000 + i, k, 0.0, queries[i % 16])
                t1 = time.perf_counter()
                s.sendall(frame)
                req, approx, smiles, ids, scores = read_reply(s)
                el = time.perf_counter() - t1
                assert req == 1000 + i and approx == rows, (req, approx)
                assert len(scores) > 0 and scores[0] == 1.0, scores[:3]
                own = b"S%010d" % query_row(i % 16, rows)
                assert own in [sm for sm, sc in zip(smiles, scores) if sc == 1.0], (own, smiles[:3])
                n_hits = len(scores)
                if i >= 5:
                    client.append(1e3 * el)
            s.close()
            time.sleep(0.05)
            log.seek(mark)
            server = [1e3 * float(ln.rsplit(":", 1)[1]) for ln in log.read().splitlines() if "time elapsed" in ln][5:]
            client.sort()
            server.sort()
            out.append({"rows": rows, "kind": kind, "k": k, "hits_returned": n_hits, "requests": nreq, "gpus": gpus,
                        "client_ms_median": client[len(client) // 2], "client_ms_mean": sum(client) / len(client),
                        "client_ms_p95": client[int(0.95 * (len(client) - 1))],
                        "server_ms_median": server[len(server) // 2] if server else None,
                        "server_ms_mean": sum(server) / len(server) if server else None,
                        "startup_s": round(startup_s, 2)})
        return out
    finally:
        p.terminate()
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
        log.close()


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [1_600_000, 16_400_000, 56_700_000, 113_000_000]
    nreq = int(os.environ.get("SL_REQUESTS", "50"))
    kind = os.environ.get("SL_KIND", "morgan")
    gpus = int(os.environ.get("SL_GPUS", "1"))
    rec = {"what": "gpusimserver end to end: request frame on /tmp/gpusimilarity -> reply frame (reference wire format), "
                   "Tanimoto, cutoff 0, synthetic %s-shaped 1024-bit table generated in HBM" % kind,
           "reference_metric": "server-side search latency, gpusim.cpp:420-429 (BASELINE.md section 1: 37.57 ms at 56.7 M rows "
                               "on 4 x V100, k = 20 ... )",
           "results": []}
    for n in sizes:
        rec["results"] += measure(n, kind, (20, 1000), nreq, gpus)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
