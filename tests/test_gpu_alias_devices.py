"""The in-process multi-device path on a one-GPU box: GSIM_TEST_ALIAS_DEVICES=4 makes the library
present four logical devices on GPU 0 (a test hook, csrc/capi_lifecycle.cpp alias_devices), so
gsim_db_finalize(db, dev, n > 1), the per-shard fan-out + host merge of single queries, batches and
folded tables (fingerprintdb_cuda.cu:356-380), gsim_next_device's round robin (:54-68) and
`gpusimserver --gpus 4` all run and are compared with the oracle / the golden protocol frames."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_in_process_multi_device_shards_on_aliased_devices():
    env = dict(os.environ, GSIM_TEST_ALIAS_DEVICES="4")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "alias_worker.py")], env=env, capture_output=True,
                         timeout=1500)
    assert out.returncode == 0, (out.stdout.decode("utf-8", "replace")[-2000:] + out.stderr.decode("utf-8", "replace")[-4000:])
    assert b"alias worker ok" in out.stdout


@pytest.mark.parametrize("gpus", ["4", "0"])
def test_server_shards_over_aliased_devices(gpus, monkeypatch, tmp_path):
    """`gpusimserver --gpus N`: every table sharded over N (0 = all) devices, replies byte-identical to the golden frames."""
    import test_host_cpp as H
    monkeypatch.setenv("GSIM_TEST_ALIAS_DEVICES", "4")
    import shutil
    pair = (str(tmp_path / "small.fsim"), str(tmp_path / "small_copy.fsim"))
    for f in pair:
        shutil.copy(os.path.join(ROOT, "tests", "golden", "small.fsim"), f)
    srv = H.Server(["--gpus", gpus, pair[0], pair[1]])
    try:
        H.check_frames(srv, "gpu")
    finally:
        srv.close()
