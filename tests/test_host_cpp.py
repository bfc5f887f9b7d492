"""The C++ host twin (gpusimilarity_amd/csrc/host): the reference's Boost tests
restated in tests/cpp/test_gpusim_host.cpp, and the socket protocol of the
`gpusimserver` backend against golden frames made with the real Qt QDataStream."""
import json
import os
import shutil
import socket
import subprocess
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.environ.get("GSIM_TEST_BIN", os.path.join(ROOT, "gpusimilarity_amd", "bin"))
GOLD = os.path.join(ROOT, "tests", "golden")
SOCK = "/tmp/gpusimilarity"


@pytest.fixture(scope="module")
def fsim_pair(tmp_path_factory):
    d = tmp_path_factory.mktemp("fsim")
    a, b = str(d / "small.fsim"), str(d / "small_copy.fsim")
    shutil.copy(os.path.join(GOLD, "small.fsim"), a)
    shutil.copy(os.path.join(GOLD, "small.fsim"), b)  # test/CMakeLists.txt:11-12 does the same
    return a, b


def run_host_tests(mode, pair):
    exe = os.path.join(BIN, "test_gpusim_host")
    assert os.path.exists(exe), "build first: make -C gpusimilarity_amd/csrc"
    p = subprocess.run([exe, mode, pair[0], pair[1]], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]


def test_reference_unit_tests_cpu_cases(fsim_pair):
    """CPUSort (test_gpusim.cpp:134-146), .fsim load, codec KAT, TestSearchMultiple on the CPU route."""
    run_host_tests("cpu", fsim_pair)


@pytest.mark.gpu
def test_reference_unit_tests_gpu_cases(fsim_pair):
    """CompareGPUtoCPU, TestSearchMultiple, TestSimilarityCutoff, getNextGPU (test_gpusim.cpp)."""
    run_host_tests("gpu", fsim_pair)


class Server:
    def __init__(self, args, exe=None, env=None):
        if os.path.exists(SOCK):
            os.unlink(SOCK)
        self.p = subprocess.Popen([exe or os.path.join(BIN, "gpusimserver")] + args, stderr=subprocess.PIPE, text=True, env=env)
        for _ in range(600):
            if os.path.exists(SOCK):
                break
            if self.p.poll() is not None:
                raise RuntimeError("gpusimserver exited: " + self.p.stderr.read()[-2000:])
            time.sleep(0.05)
        else:
            raise RuntimeError("gpusimserver did not open " + SOCK)

    def ask(self, request: bytes, split=False) -> bytes:
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        s.connect(SOCK)
        s.settimeout(30)
        if split:  # the frame arrives in two pieces: the server must buffer it
            s.sendall(request[:7])
            time.sleep(0.05)
            s.sendall(request[7:])
        else:
            s.sendall(request)
        # reply = i32 req, i32 n, u64 approx, n cstr, n cstr, n x 8 bytes
        buf = b""

        def complete(b):
            if len(b) < 16:
                return False
            n = int.from_bytes(b[4:8], "big")
            off = 16
            for _ in range(2 * n):
                if len(b) < off + 4:
                    return False
                off += 4 + int.from_bytes(b[off:off + 4], "big")
            return len(b) >= off + 8 * n

        while not complete(buf):
            chunk = s.recv(65536)
            if not chunk:
                break
            buf += chunk
        s.close()
        return buf

    def close(self):
        self.p.terminate()
        try:
            self.p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            self.p.kill()


def check_frames(server, mode, may_split=True, approx_undefined=False):
    frames = json.load(open(os.path.join(GOLD, "protocol_frames.json")))["frames"]
    n = 0
    for i, f in enumerate(frames):
        if f["mode"] not in (mode, "both"):
            continue
        reply = server.ask(bytes.fromhex(f["request"]), split=(may_split and i % 2 == 1))
        if approx_undefined:  # bytes 8..15 = u64 approx
            assert len(reply) >= 16 and reply[:8].hex() + reply[16:].hex() == f["reply"][:16] + f["reply"][32:], f["name"]
            n += 1
            continue
        assert reply.hex() == f["reply"], f["name"]
        n += 1
    assert n >= 6


def test_server_protocol_cpu_only(fsim_pair):
    """`gpusimserver --cpu_only`: byte-identical replies to the golden frames (no GPU needed)."""
    srv = Server(["--cpu_only", fsim_pair[0], fsim_pair[1]])
    try:
        check_frames(srv, "cpu")
    finally:
        srv.close()
    assert not os.path.exists(SOCK)  # removed on shutdown


def test_seam_b_links_and_serves_the_golden_frames(fsim_pair, tmp_path):
    """INTEGRATION.md seam B, linked: the REFERENCE's own gpusim.cpp + main.cpp + fingerprintdb_cuda.cpp +
    calculation_functors.cpp, compiled where they lie, around docs/fingerprintdb_hip.cpp + libgsim_hip.so in place of
    fingerprintdb_cuda.cu (reference CMakeLists.txt:60-67) -- a scratch binary in the test's temporary directory, dev
    container only.  Started --cpu_only on the reference's own fixture it answers the golden request frames (made with the
    real Qt QDataStream) byte for byte: the adapter's constructor, storage bookkeeping, getFingerprint and accessors carry
    the reference's server; its search() is exercised on the GPU box through the same C ABI by the Qt-free twin."""
    if not os.path.isdir("/root/reference") or not os.path.isdir("/opt/conda/include/qt/QtCore"):
        pytest.skip("dev container only: needs /root/reference and the image's Qt 5.9")
    out = subprocess.run(["bash", os.path.join(ROOT, "scripts", "check_seam_b.sh"), "--link", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    exe = str(tmp_path / "refserver_hip")
    srv = Server(["--cpu_only", fsim_pair[0], fsim_pair[1]], exe=exe, env=dict(os.environ, QT_LOGGING_RULES="*.debug=false"))
    try:
        # (the reference reads a request with ONE readAll(), gpusim.cpp:380: whole frames only; and on the CPU route its reply's
        # `approx` is an uninitialised local -- search_cpu never writes it, gpusim.cpp:324-330 adds it up anyway -- so those
        # eight bytes are whatever the stack held: every other byte of every reply is compared; the product's server sends 0)
        check_frames(srv, "cpu", may_split=False, approx_undefined=True)
    finally:
        srv.close()
        os.unlink(exe)


@pytest.mark.gpu
def test_server_protocol_gpu(fsim_pair):
    srv = Server([fsim_pair[0], fsim_pair[1]])
    try:
        check_frames(srv, "gpu")
    finally:
        srv.close()


def test_server_cli_errors(fsim_pair):
    exe = os.path.join(BIN, "gpusimserver")
    assert subprocess.run([exe, "--cpu_only", "--gpu_bitcount", "512", fsim_pair[0]], capture_output=True).returncode == 1
    assert subprocess.run([exe, "--gpu_bitcount", "abc", fsim_pair[0]], capture_output=True).returncode == 1
    assert subprocess.run([exe, "/nonexistent/x.fsim"], capture_output=True).returncode == 1


PYQT = "/opt/conda/bin/python3.9"


@pytest.mark.skipif(not os.path.exists(PYQT), reason="PyQt5 (real QLocalSocket client) only in the dev container")
def test_real_qt_client_against_cpu_server(fsim_pair):
    """The reference client's exact Qt calls (python/gpusim_search.py:27-71: QLocalSocket
    'gpusimilarity', QDataStream writes/reads) against our server."""
    client = r'''
import sys, json, struct, zlib
from PyQt5 import QtCore, QtNetwork
app = QtCore.QCoreApplication([])
sock = QtNetwork.QLocalSocket(app)
sock.connectToServer('gpusimilarity')
assert sock.waitForConnected(5000), sock.errorString()
fp = bytes.fromhex(sys.argv[1])
out = QtCore.QByteArray(); qds = QtCore.QDataStream(out, QtCore.QIODevice.WriteOnly)
qds.writeInt(1); qds.writeString(b'small'); qds.writeString(b'pass')
qds.writeInt(424242); qds.writeInt(20); qds.writeFloat(0); qds << QtCore.QByteArray(fp)
sock.write(out); sock.flush(); sock.waitForReadyRead(30000)
data = sock.readAll(); rd = QtCore.QDataStream(data)
assert rd.readInt() == 424242
n = rd.readInt(); approx = rd.readUInt64()
smiles = [rd.readString() for _ in range(n)]; ids = [rd.readString() for _ in range(n)]
scores = [rd.readFloat() for _ in range(n)]
print(json.dumps({"n": n, "ids": [i.decode() for i in ids], "scores": scores}))
'''
    from gpusimilarity_amd.fsim import read_fsim
    fs = read_fsim(fsim_pair[0])
    srv = Server(["--cpu_only", fsim_pair[0]])
    try:
        p = subprocess.run([PYQT, "-c", client, fs.rows()[3].tobytes().hex()], capture_output=True, text=True, timeout=60)
        assert p.returncode == 0, p.stderr[-2000:]
        got = json.loads(p.stdout.strip().splitlines()[-1])
    finally:
        srv.close()
    assert got["n"] == 20 and got["ids"][0] == "ZINC00000022" and got["scores"][0] == 1.0
    assert got["ids"][1] == "ZINC00000323"  # SURVEY Appendix C, query = row 3


@pytest.mark.gpu
def test_server_gpu_bitcount_folds_like_the_reference(fsim_pair):
    """`gpusimserver --gpu_bitcount 512` (gpusim.cpp:144-163): tables are folded by 1024/512 = 2 and
    searched the reference's approximate way; replies match the oracle's folded search."""
    import struct
    import numpy as np
    import oracle_lib as O
    from gpusimilarity_amd.fsim import read_fsim
    fs = read_fsim(fsim_pair[0])
    rows = fs.rows()

    def cstr(b):
        return struct.pack(">I", len(b) + 1) + b + b"\0"

    srv = Server(["--gpu_bitcount", "512", fsim_pair[0]])
    try:
        for qrow, k, cutoff in ((3, 10, 0.0), (0, 5, 0.25)):
            fp = rows[qrow].tobytes()
            req = (struct.pack(">i", 1) + cstr(b"small") + cstr(b"pass") + struct.pack(">iid", 4242 + qrow, k, cutoff) +
                   struct.pack(">I", len(fp)) + fp)
            reply = srv.ask(req)
            reqnum, n, approx = struct.unpack(">iiQ", reply[:16])
            want, wap = O.search_folded(rows[qrow], rows, 2, k, cutoff)
            assert reqnum == 4242 + qrow and n == len(want) and approx == wap
            off, ids = 16, []
            for _ in range(2 * n):
                ln = struct.unpack(">I", reply[off:off + 4])[0]
                ids.append(reply[off + 4:off + 4 + ln - 1])
                off += 4 + ln
            assert ids[n:] == [fs.ids[int(r)] for r in want["row"]]
            scores = struct.unpack(">%dd" % n, reply[off:off + 8 * n])
            assert [np.float32(s) for s in scores] == [s for s in want["score"]]
    finally:
        srv.close()
