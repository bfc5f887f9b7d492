#!/opt/conda/bin/python3.9
"""Emit golden socket-protocol frames with the REAL Qt QDataStream (PyQt5 5.9, only
present under /opt/conda in the dev container) -> tests/golden/protocol_frames.json.

Requests are written exactly as the reference client writes them
(python/gpusim_search.py:36-47: writeInt, writeString x2 per db, writeInt,
writeInt, writeFloat, `<< QByteArray`); expected replies are written the way
GPUSimServer::incomingSearchRequest serialises them (gpusim.cpp:432-453: ints,
n x char*, n x char*, n x float-as-double).  Expected result CONTENT comes from
tests/golden/small_fsim_topk.json (reference functor scores) and the strings in
small.fsim; the multi-db fold follows gpusim.cpp:339-373 (ids joined by ';:;').
"""
import json
import os
import struct
import sys
import zlib

from PyQt5 import QtCore

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def read_small():
    b = open(os.path.join(GOLD, "small.fsim"), "rb").read()
    off = [0]

    def u32():
        v = struct.unpack_from(">I", b, off[0])[0]
        off[0] += 4
        return v

    def take(n):
        s = b[off[0]:off[0] + n]
        off[0] += n
        return s

    u32()
    take(u32())
    bits, cnt = u32(), u32()

    def blocks():
        out = []
        for _ in range(u32()):
            d = take(u32())
            out.append(zlib.decompress(d[4:]))
        return out

    fp, smi, ids = blocks(), blocks(), blocks()

    def strs(payload):
        o, res = 0, []
        while o < len(payload):
            n = struct.unpack_from(">I", payload, o)[0]
            res.append(payload[o + 4:o + 4 + n - 1])
            o += 4 + n
        return res

    return fp[0], strs(smi[0]), strs(ids[0]), bits // 8


def request(dbs, request_num, k, cutoff, fp_bytes):
    qba = QtCore.QByteArray()
    qds = QtCore.QDataStream(qba, QtCore.QIODevice.WriteOnly)
    qds.writeInt(len(dbs))
    for name, key in dbs:
        qds.writeString(name.encode())
        qds.writeString(key.encode())
    qds.writeInt(request_num)
    qds.writeInt(k)
    qds.writeFloat(cutoff)
    qds << QtCore.QByteArray(fp_bytes)
    return bytes(qba)


def reply(request_num, approx, smiles, ids, scores):
    qba = QtCore.QByteArray()
    qds = QtCore.QDataStream(qba, QtCore.QIODevice.WriteOnly)
    qds.writeInt(request_num)
    qds.writeInt(len(smiles))
    qds.writeUInt64(approx)
    for s in smiles:
        qds.writeString(s)
    for s in ids:
        qds.writeString(s)
    for f in scores:
        qds.writeFloat(f)
    return bytes(qba)


def f32(bits_hex):
    return struct.unpack(">f", bytes.fromhex(bits_hex))[0]


def main():
    fp, smiles, ids, row_bytes = read_small()
    gold = json.load(open(os.path.join(GOLD, "small_fsim_topk.json")))
    frames = []
    for qe in gold["queries"]:
        qrow = qe["query_row"]
        q = fp[qrow * row_bytes:(qrow + 1) * row_bytes]
        cases = {(c["k"], c["cutoff"]): c for c in qe["cases"]}
        for (k, cutoff) in ((10, 0.0), (10, 0.3), (15, 0.1)):
            c = cases[(k, cutoff)]
            req_num = 0x1000 + qrow * 16 + k
            rows = c["rows"]
            sc = [f32(b) for b in c["score_bits"]]
            frames.append({
                "name": "single q%d k%d cutoff%g" % (qrow, k, cutoff), "mode": "gpu",
                "request": request([("small", "pass")], req_num, k, cutoff, q).hex(),
                "reply": reply(req_num, c["approx"], [smiles[r] for r in rows], [ids[r] for r in rows], sc).hex()})
        # CPU route (--cpu_only): cutoff ignored, approx not produced (0), k results
        c = cases[(10, 0.0)]
        rows, sc = c["rows"], [f32(b) for b in c["score_bits"]]
        frames.append({"name": "cpu q%d k10 cutoff0.3(ignored)" % qrow, "mode": "cpu",
                       "request": request([("small", "pass")], 77 + qrow, 10, 0.3, q).hex(),
                       "reply": reply(77 + qrow, 0, [smiles[r] for r in rows], [ids[r] for r in rows], sc).hex()})
        # two databases holding the same molecules (test_gpusim.cpp:71-99): every SMILES appears
        # twice with the same score; gpusim.cpp:339-373 folds them, joining ids with ';:;'.  The
        # id map stops growing once it holds k distinct SMILES, so the k-th result has seen only
        # its first copy.
        k = 10
        rows, sc = c["rows"][:k], [f32(b) for b in c["score_bits"]][:k]
        joined = [ids[r] + b";:;" + ids[r] for r in rows[:-1]] + [ids[rows[-1]]]
        for mode, approx in (("gpu", 200), ("cpu", 0)):
            frames.append({"name": "two dbs q%d k10 %s" % (qrow, mode), "mode": mode,
                           "request": request([("small", "pass"), ("small_copy", "pass")], 900 + qrow, k, 0.0, q).hex(),
                           "reply": reply(900 + qrow, approx, [smiles[r] for r in rows], joined, sc).hex()})
    # wrong key / unknown database: empty reply
    q = fp[:row_bytes]
    frames.append({"name": "wrong key", "mode": "both",
                   "request": request([("small", "nope")], 5, 10, 0.0, q).hex(), "reply": reply(5, 0, [], [], []).hex()})
    frames.append({"name": "unknown db", "mode": "both",
                   "request": request([("missing", "pass")], 6, 10, 0.0, q).hex(), "reply": reply(6, 0, [], [], []).hex()})
    out = {"source": "PyQt5 %s QDataStream, stream objects at their default version as in the reference client (ints, char*, double and QByteArray encode identically in every Qt 5 stream version)" % QtCore.QT_VERSION_STR,
           "frames": frames}
    with open(os.path.join(GOLD, "protocol_frames.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote %d frames" % len(frames))


if __name__ == "__main__":
    main()
