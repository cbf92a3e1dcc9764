"""N>1 path of the batched solvers (BASELINE config 3): a fixed batch split over ranks, no collective
during the solve, one all-gather of the results.  world_size-2/3 gloo runs: CPU protocol test, one-GPU rehearsal."""
import os
import socket
import subprocess
import sys

import pytest

from tests._util import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(mode, nproc, count, m, n):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_shard_worker.py"), mode, str(count), str(m), str(n)]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count(" ok") == nproc


def test_shard_ranges_partition_the_batch():
    import importlib.util

    spec = importlib.util.spec_from_file_location("lpx_batched_src", os.path.join(ROOT, "linear-programming_amd", "batched.py"))
    src = open(spec.origin).read()
    ns = {}
    exec(src[src.index("def shard("):src.index("class ResultGather")], ns)  # the pure function, without the GPU package
    for count in (1, 7, 64, 4096, 4099):
        for world in (1, 2, 3, 8):
            edges = [ns["shard"](count, g, world) for g in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == count
            assert all(edges[g][1] == edges[g + 1][0] for g in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("nproc,count", [(2, 10), (3, 11)])
def test_protocol_cpu_gloo(nproc, count):
    _run("cpu", nproc, count, 12, 30)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,count,m,n", [(2, 64, 64, 128), (3, 50, 24, 50)])
def test_product_path_sharded_on_one_gpu(nproc, count, m, n):
    _run("gpu", nproc, count, m, n)
