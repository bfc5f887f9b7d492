"""The N > 1 data path with the REAL local search: several processes, each owning a row
shard on cuda:0, gather of the per-rank result blocks, device merge -- equal to the oracle
on the whole table (fingerprintdb_cuda.cu:356-380 is the reference's fan-out + merge).
Also: bench.py launches its own ranks when started without torch.distributed.run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_match_the_oracle(world):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker_gpu.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out.decode("utf-8", "replace")[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: the parent starts two ranks itself.  On a
    one-GPU box the ranks share cuda:0 over gloo (GSIM_BENCH_SHARE_GPU=1, test-only); the line
    must report world 2 and both ranks."""
    env = dict(os.environ, GSIM_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                          "--rows-per-gpu", "2000000", "--no-cpu-baseline", "--no-configs"], env=env, capture_output=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr.decode("utf-8", "replace")[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["collective"]["world"] == 2
    assert sorted(r["rank"] for r in rec["collective"]["ranks"]) == [0, 1]
    assert rec["value"] > 0
