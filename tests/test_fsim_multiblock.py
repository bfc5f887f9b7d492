"""Multi-block .fsim databases (gpusim.cpp:202-236: a thread-pool qUncompress of N FP blocks -> N
storages; writer roll-over python/gpusim_createdb.py:56-69): a 3-FP-block / 2-SMILES-block /
3-ID-block file through the Python reader, the C++ extractData inside `gpusimserver` (CPU route,
unfolded GPU route = one table, folded GPU route = three storages), against the oracle; and the
negative cases (truncated file, corrupt block, count mismatch: gpusim.cpp:186-189,
fingerprintdb_cuda.cu:153-156)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from gpusimilarity_amd.fsim import read_fsim, write_fsim
from test_host_cpp import BIN, Server

BLOCKS = (700, 1300, 1001)
W = 32


@pytest.fixture(scope="module")
def multi(tmp_path_factory):
    d = tmp_path_factory.mktemp("multi")
    n = sum(BLOCKS)
    rows = O.synth_rows(0xF51A, 0, 0, n, W)
    parts, pos = [], 0
    for b in BLOCKS:
        parts.append(rows[pos:pos + b])
        pos += b
    smiles = [b"C" * (1 + i % 7) + str(i).encode() for i in range(n)]  # all distinct
    ids = [b"ID%07d" % i for i in range(n)]
    path = str(d / "multi.fsim")
    write_fsim(path, "sesame", W * 32, parts, smiles, ids, smiles_blocks=2, id_blocks=3)
    return path, rows, smiles, ids


def cstr(b):
    return struct.pack(">I", len(b) + 1) + b + b"\0"


def ask(srv, name, key, reqnum, k, cutoff, fp):
    req = (struct.pack(">i", 1) + cstr(name) + cstr(key) + struct.pack(">iid", reqnum, k, cutoff) + struct.pack(">I", len(fp)) + fp)
    reply = srv.ask(req)
    rq, n, approx = struct.unpack(">iiQ", reply[:16])
    assert rq == reqnum
    off, strings = 16, []
    for _ in range(2 * n):
        ln = struct.unpack(">I", reply[off:off + 4])[0]
        strings.append(reply[off + 4:off + 4 + ln - 1])
        off += 4 + ln
    scores = [np.float32(s) for s in struct.unpack(">%dd" % n, reply[off:off + 8 * n])]
    return strings[:n], strings[n:], scores, approx


def test_python_reader_round_trip(multi):
    path, rows, smiles, ids = multi
    f = read_fsim(path)
    assert f.dbkey == "sesame" and f.fp_bitcount == W * 32 and f.fp_count == len(rows)
    assert [b.shape[0] for b in f.fp_blocks] == list(BLOCKS)
    assert (f.rows() == rows).all() and f.smiles == smiles and f.ids == ids


def test_server_cpu_route_reads_every_block(multi):
    """extractData (C++) inflates all blocks; the CPU route scores the concatenation: hits = the oracle's
    search_cpu (bubble-sort order), strings from the right SMILES / ID blocks."""
    path, rows, smiles, ids = multi
    srv = Server(["--cpu_only", path])
    try:
        for qrow, k in ((5, 10), (1999, 25), (2999, 3)):  # queries from each FP block
            got_smiles, got_ids, scores, _ = ask(srv, b"multi", b"sesame", 77 + qrow, k, 0.0, rows[qrow].tobytes())
            wrows, wscores = O.search_cpu(rows[qrow], rows, k)
            assert got_ids == [ids[int(r)] for r in wrows] and got_smiles == [smiles[int(r)] for r in wrows]
            assert scores == list(wscores)
        # two requests in ONE write: both are answered, in order (the second must not be dropped)
        import socket
        from test_host_cpp import SOCK
        fp = rows[42].tobytes()
        one = lambda num: (struct.pack(">i", 1) + cstr(b"multi") + cstr(b"sesame") + struct.pack(">iid", num, 1, 0.0) +  # noqa: E731
                           struct.pack(">I", len(fp)) + fp)
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        s.connect(SOCK)
        s.settimeout(30)
        s.sendall(one(1001) + one(1002))
        buf = b""
        while buf.count(ids[42]) < 2:
            chunk = s.recv(65536)
            assert chunk, "connection closed after %d bytes" % len(buf)
            buf += chunk
        s.close()
        first = 16 + (4 + len(smiles[42]) + 1) + (4 + len(ids[42]) + 1) + 8
        assert struct.unpack(">i", buf[:4])[0] == 1001 and struct.unpack(">i", buf[first:first + 4])[0] == 1002
    finally:
        srv.close()


@pytest.mark.gpu
def test_server_gpu_routes_one_table_and_three_folded_storages(multi):
    path, rows, smiles, ids = multi
    srv = Server([path])  # unfolded: the three blocks are one table on the GPU
    try:
        for qrow, k, cutoff in ((5, 10, 0.0), (1999, 1000, 0.0), (2999, 20, 0.09)):
            _, got_ids, scores, approx = ask(srv, b"multi", b"sesame", 1, k, cutoff, rows[qrow].tobytes())
            want, wap = O.search(rows[qrow], rows, k, cutoff, nthreads=4)
            assert approx == wap and got_ids == [ids[int(r)] for r in want["row"]] and scores == list(want["score"])
        # a huge and a non-positive count from the socket: clamped / answered with nothing, the server survives
        _, got_ids, _, _ = ask(srv, b"multi", b"sesame", 2, 2_000_000_000, 0.0, rows[5].tobytes())
        assert len(got_ids) == len(rows)
        assert ask(srv, b"multi", b"sesame", 3, -5, 0.0, rows[5].tobytes())[1] == []
    finally:
        srv.close()
    srv = Server(["--gpu_bitcount", "256", path])  # folded by 4: every FP block is its own storage (3 shards)
    try:
        for qrow, k, cutoff in ((5, 10, 0.0), (1999, 7, 0.2)):
            _, got_ids, scores, approx = ask(srv, b"multi", b"sesame", 4, k, cutoff, rows[qrow].tobytes())
            # fingerprintdb_cuda.cu:228-339 per storage, merged as :363-380
            parts, wap, base = [], 0, 0
            for b in BLOCKS:
                h, ap = O.search_folded(rows[qrow], rows[base:base + b], 4, k, cutoff, row_base=base)
                parts.append(h)
                wap += ap
                base += b
            want = O.merge_hits(parts, k)
            assert approx == wap and got_ids == [ids[int(r)] for r in want["row"]] and scores == list(want["score"])
    finally:
        srv.close()


def test_corrupt_and_truncated_files_are_refused(multi, tmp_path):
    path = multi[0]
    raw = open(path, "rb").read()
    exe = os.path.join(BIN, "gpusimserver")

    def refused(data, text):
        p = str(tmp_path / "bad.fsim")
        open(p, "wb").write(data)
        r = subprocess.run([exe, "--cpu_only", p], capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and text in r.stderr, r.stderr[-500:]
        with pytest.raises(Exception):
            read_fsim(p)

    refused(raw[:len(raw) // 2], "gpusimserver:")                      # truncated in the middle of a block
    refused(struct.pack(">i", 2) + raw[4:], "Database version incompatible")  # gpusim.cpp:186-189
    bad = bytearray(raw)
    bad[200:260] = bytes(60)                                             # garbage inside the first zlib stream
    refused(bytes(bad), "gpusimserver:")
    off = 4 + 4 + len(b"sesame") + 1 + 4                                 # version, cstr dbkey, fp_bitcount
    refused(raw[:off] + struct.pack(">i", sum(BLOCKS) + 1) + raw[off + 4:], "Mismatch between FP count and data")
    # a wild block count (the i32 behind fp_count): refused because the rest of the file cannot hold that many blocks,
    # before anything is sized by it; and a negative one
    refused(raw[:off + 4] + struct.pack(">i", 0x7FFFFFF0) + raw[off + 8:], "gpusimserver:")
    refused(raw[:off + 4] + struct.pack(">i", -3) + raw[off + 8:], "gpusimserver:")
    refused(b"", "gpusimserver:")                                        # an empty file


def test_more_blocks_than_workers(tmp_path):
    """100 FP blocks (more than kMaxExtractThreads = 32 workers; the reader used to start one thread per block): every
    block is read, in file order, and the CPU route answers from all of them."""
    nb, per = 100, 37
    n = nb * per
    rows = O.synth_rows(0xB10C, 0, 0, n, W)
    parts = [rows[i * per:(i + 1) * per] for i in range(nb)]
    smiles = [b"N%d" % i for i in range(n)]
    ids = [b"I%d" % i for i in range(n)]
    path = str(tmp_path / "many.fsim")
    write_fsim(path, "k", W * 32, parts, smiles, ids, smiles_blocks=40, id_blocks=50)
    f = read_fsim(path)
    assert len(f.fp_blocks) == nb
    srv = Server(["--cpu_only", path])
    try:
        for q in (0, n // 2, n - 1):  # rows of the first, a middle and the last block: each finds itself first
            fp = rows[q].tobytes()
            smi, idv, scores, _ = ask(srv, b"many", b"k", q, 3, 0.0, fp)
            assert smi[0] == smiles[q] and idv[0] == ids[q] and scores[0] == np.float32(1.0)
    finally:
        srv.close()
