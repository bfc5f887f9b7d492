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
    from conftest import hooks_env
    env = hooks_env(GSIM_TEST_ALIAS_DEVICES="4")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "alias_worker.py")], env=env, capture_output=True,
                         timeout=1500)
    assert out.returncode == 0, (out.stdout.decode("utf-8", "replace")[-2000:] + out.stderr.decode("utf-8", "replace")[-4000:])
    assert b"alias worker ok" in out.stdout


def test_c_abi_collective_world_one_rccl():
    """gsim_comm_create on the lease's one GPU is a real ncclCommInitAll (world 1): gsim_db_search through the collective
    route -- block in HBM, ncclAllGather in place, merge_kernel into pinned memory -- equals the host route and the oracle,
    single queries and a 70-query batch.  (World > 1 needs more GPUs than a lease has; the gather's slot layout and the
    merge run at world 3 and 4 in the alias worker above through the loop-back gather.)"""
    import numpy as np
    import oracle_lib as O
    from gpusimilarity_amd import capi
    n, W = 2_000_003, 32
    db = O.synth_rows_mt(0xC011EC7, 2, 0, n, W)
    t = capi.Table(W * 32).add_rows(db).finalize(0, 1)
    comm = capi.Comm([0])
    assert comm.size() == 1
    t.set_comm(comm)
    t.enable_timing(True)
    qs = np.stack([db[O.query_row(i, n)] for i in range(70)])
    for qi, (k, cutoff) in enumerate(((1000, 0.0), (10, 0.0), (1000, 0.25), (9000, 0.0))):
        hits, approx = t.search(qs[qi], k, cutoff)
        want, wap = O.search(qs[qi], db, k, cutoff, nthreads=8)
        assert int(approx[0]) == wap and (hits[0]["row"] == want["row"]).all()
        assert (hits[0]["score"].view(np.uint32) == want["score"].view(np.uint32)).all()
        assert (hits[0]["common"] == want["common"]).all() and (hits[0]["popc_db"] == want["popc_db"]).all()
    bh, bap = t.search(qs, 100, 0.0)
    for i in (0, 33, 69):
        want, wap = O.search(qs[i], db, 100, 0.0, nthreads=8)
        assert int(bap[i]) == wap and (bh[i]["row"] == want["row"]).all()
        assert (bh[i]["score"].view(np.uint32) == want["score"].view(np.uint32)).all()
    tm = t.timing()
    assert tm["collectives"] == 5, tm
    t.set_comm(None)
    t.close()
    comm.close()
    with pytest.raises(capi.GsimError):
        capi.Comm([0, 0])  # a device twice
    with pytest.raises(capi.GsimError):
        capi.Comm([5])     # not present on a one-GPU lease


@pytest.mark.parametrize("gpus,merge", [("4", "host"), ("0", "host"), ("4", "rccl"), ("1", "rccl")])
def test_server_shards_over_aliased_devices(gpus, merge, monkeypatch, tmp_path):
    """`gpusimserver --gpus N [--merge rccl]`: every table sharded over N (0 = all) devices, the shards' results merged on
    the host or through the C ABI's collective (aliased devices: its loop-back gather; one device: a real world-1 RCCL
    communicator) -- replies byte-identical to the golden frames."""
    import test_host_cpp as H
    from conftest import hooks_env
    for name, value in hooks_env(GSIM_TEST_ALIAS_DEVICES="4").items():  # (the server inherits the environment)
        if name in ("GSIM_TEST_ALIAS_DEVICES", "LD_LIBRARY_PATH", "GSIM_LIB"):
            monkeypatch.setenv(name, value)
    import shutil
    pair = (str(tmp_path / "small.fsim"), str(tmp_path / "small_copy.fsim"))
    for f in pair:
        shutil.copy(os.path.join(ROOT, "tests", "golden", "small.fsim"), f)
    if gpus == "1":
        monkeypatch.delenv("GSIM_TEST_ALIAS_DEVICES")
    srv = H.Server(["--gpus", gpus, "--merge", merge, pair[0], pair[1]])
    try:
        H.check_frames(srv, "gpu")
    finally:
        srv.close()


def test_bench_in_process_on_four_aliased_devices_names_its_collective():
    """the first contact with a real multi-GPU node must not be able to fail on plumbing.  `bench.py --gpus 4
    --in-process` on four aliased devices (test-hooks build): ONE line under 12 KB, `collective.world == 4` with all four ranks
    listed and the data-path collective named, the RCCL the library is bound to reported, identical hits between the host merge
    and the gsim_comm route asserted by the script itself and stated in the line, every route timed."""
    import json
    from conftest import hooks_env
    env = hooks_env(GSIM_TEST_ALIAS_DEVICES="4")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--in-process", "--steps", "2", "--warmup", "1",
                          "--queries-per-step", "16", "--rows-per-gpu", "2000000", "--no-torch-route"], env=env, capture_output=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr.decode("utf-8", "replace")[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 12000, (len(lines), len(lines[0]) if lines else 0)
    d = json.loads(lines[0])
    c = d["collective"]
    assert d["n_gpus"] == 4 and c["world"] == 4 and [m["rank"] for m in c["ranks"]] == [0, 1, 2, 3]
    assert len({m["device"] for m in c["ranks"]}) == 4 and "ncclAllGather" in c["data_path_collective"]
    assert c["identical_hits_across_routes"] is True and d["routes"]["gsim_comm"]["identical_to_host_merge"] is True
    assert c["rccl"]["runtime_version"] // 10000 == c["rccl"]["header_version"] // 10000 and os.path.exists(c["rccl"]["path"])
    for name in ("host_merge", "gsim_comm", "twin"):
        assert d["routes"][name]["ms_per_query"] > 0, name
    assert d["routes"]["gsim_comm"]["collectives"] == 2 * 16 and d["routes"]["gsim_comm"]["gather_us_avg"] > 0
    assert d["config"]["shards"] == 4 and d["value"] > 0 and "roofline" in d
