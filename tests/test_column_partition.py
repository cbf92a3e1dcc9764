"""N>1 path of the column-partitioned tableau: world_size-2/3 gloo runs (CPU protocol test, GPU rehearsal)."""
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


def _run(mode, nproc, m, n, seed, maxiters, **extra_env):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_colpart_worker.py"), mode, str(m), str(n), str(seed), str(maxiters)]
    env = dict(os.environ, OMP_NUM_THREADS="2", **extra_env)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count(" ok") == nproc


@pytest.mark.parametrize("nproc,m,n", [(2, 16, 32), (3, 24, 50)])
def test_protocol_cpu_gloo(nproc, m, n):
    _run("cpu", nproc, m, n, 1, 100000)


@pytest.mark.parametrize("nproc,m,n", [(2, 16, 32), (3, 24, 50)])
def test_protocol_deferred_updates_cpu_gloo(nproc, m, n):
    """The exchange with deferred updates (lpx_tableau_block_*), NumPy model on CPU: proposals and pivot rows formed from
    the stale slice plus the pending updates, flush every 5 pivots -- the oracle's pivot sequence on every rank."""
    _run("cpu-deferred", nproc, m, n, 1, 100000)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,m,n", [(2, 64, 128), (3, 50, 121)])
def test_product_path_two_ranks_one_gpu(nproc, m, n):
    _run("gpu", nproc, m, n, 2, 100000)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"LPX_TABLEAU_SMALL": "0"}, {"LPX_TABLEAU_LOOKAHEAD": "0"},
                                 {"LPX_TABLEAU_SMALL": "0", "LPX_TABLEAU_LOOKAHEAD": "0"},
                                 {"LPX_TABLEAU_CONDENSED": "0"}, {"LPX_TABLEAU_CONDENSED": "0", "LPX_TABLEAU_SMALL": "0"},
                                 {"LPX_TABLEAU_CONDENSED": "0", "LPX_TABLEAU_LOOKAHEAD": "0"}],
                         ids=lambda e: ",".join(f"{k[12:]}={v}" for k, v in e.items()))
def test_product_path_lookahead_and_kernel_variants(env):
    """The default run above overlaps the exchange of pivot t+1 with the update of pivot t (lpx_tableau_commit_begin /
    _update) behind the single-workgroup prologue; here the multi-workgroup select/prepare chain with lookahead, and both
    chains with the plain propose -> all-gather -> commit loop, and the full (every column) layout instead of the condensed
    one.  All must give the oracle's pivot sequence."""
    _run("gpu", 2, 64, 128, 2, 100000, **env)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,m,n,env", [
    (2, 64, 128, {"LPX_TABLEAU_BLOCK": "8"}), (3, 50, 121, {"LPX_TABLEAU_BLOCK": "32"}),
    (2, 64, 128, {"LPX_TABLEAU_BLOCK": "16", "LPX_TABLEAU_CONDENSED": "0"}),
    (3, 50, 121, {"LPX_TABLEAU_BLOCK": "8", "LPX_TABLEAU_CONDENSED": "0"})],
    ids=lambda v: ",".join(f"{k[12:]}={x}" for k, x in v.items()) if isinstance(v, dict) else str(v))
def test_product_path_deferred_updates(nproc, m, n, env):
    """The column-partitioned loop with deferred updates (lpx_tableau_block_*: propose -> all-gather -> commit per pivot
    from the stale slice plus the pending pairs, one pass over each rank's slice per block), forced on at test sizes: slack
    and float32-inverted starting bases, column-slice construction, both layouts -- the oracle's pivot sequences."""
    _run("gpu", nproc, m, n, 2, 100000, LPX_TABLEAU_BLOCK_MIN_MB="0", **env)


@pytest.mark.gpu
def test_product_path_ranks_agree_on_one_block_size():
    """Ragged slices at a size threshold: lpx_tableau_block_size() of the ranks' own slices differs (here forced: rank 0
    alone gets deferred updates).  Every rank must run with the same block -- else they flush and stop at different pivots
    and the collective counts diverge (a hang) -- so the engine all-reduces it (tableau.py: TableauEngine.kb)."""
    _run("gpu", 3, 50, 121, 2, 100000, LPX_TEST_RAGGED_KB="1", LPX_TABLEAU_BLOCK="16")


@pytest.mark.gpu
@pytest.mark.parametrize("mode,env", [
    ("gpu", {"LPX_TABLEAU_BLOCK": "8"}), ("gpu", {"LPX_TABLEAU_BLOCK": "32", "LPX_TABLEAU_PIPELINE_MIN_MB": "0"}),
    ("gpu", {"LPX_TABLEAU_BLOCK": "16", "LPX_TABLEAU_PIPELINE_MIN_MB": "0", "LPX_TB_PIPE_CUS": "0"}),
    ("gpu", {"LPX_TABLEAU_BLOCK": "16", "LPX_TABLEAU_CONDENSED": "0"}), ("gpu-dual", {"LPX_TABLEAU_BLOCK": "8"}),
    ("gpu-dual", {"LPX_TABLEAU_BLOCK": "32", "LPX_TABLEAU_CONDENSED": "0"})],
    ids=lambda v: ",".join(f"{k[4:]}={x}" for k, x in v.items()) if isinstance(v, dict) else str(v))
def test_library_loop_over_rccl_one_rank(mode, env):
    """lpx_tableau_dist_iterate / lpx_tableau_dual_dist_iterate: propose -> ncclAllGather -> commit for every pivot and the
    passes over the slice, enqueued by liblpx itself over a communicator of its own -- world size 1 (RCCL refuses two ranks
    on one GPU), in-place and pipelined chains, both layouts, primal and dual: the oracle's pivot sequences."""
    _run(mode, 1, 64, 128, 2, 100000, LPX_TABLEAU_BLOCK_MIN_MB="0", LPX_TEST_BACKEND="nccl", **env)


@pytest.mark.parametrize("nproc,m,n", [(2, 16, 32), (3, 24, 50)])
def test_dual_protocol_cpu_gloo(nproc, m, n):
    _run("cpu-dual", nproc, m, n, 1, 100000)


@pytest.mark.parametrize("nproc,m,n", [(2, 16, 32), (3, 24, 50)])
def test_dual_protocol_lookahead_cpu_gloo(nproc, m, n):
    """The two-slot ordering of the dual loop with lookahead (decide and gather pivot t+1, then apply pivot t)."""
    _run("cpu-ahead-dual", nproc, m, n, 1, 100000)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,m,n", [(2, 64, 128), (3, 50, 121)])
def test_dual_product_path_two_ranks_one_gpu(nproc, m, n):
    _run("gpu-dual", nproc, m, n, 2, 100000)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"LPX_TABLEAU_LOOKAHEAD": "0"}, {"LPX_TABLEAU_CONDENSED": "0"},
                                 {"LPX_TABLEAU_CONDENSED": "0", "LPX_TABLEAU_LOOKAHEAD": "0"}],
                         ids=lambda e: ",".join(f"{k[12:]}={v}" for k, v in e.items()))
def test_dual_product_path_lookahead_and_layouts(env):
    """The default run above decides the leaving row and the proposal of pivot t+1 before pivot t's update is launched
    (lpx_tableau_dual_commit_begin) and gathers them while it runs; here the plain propose -> all-gather -> commit loop and
    the full (every column) layout, with and without lookahead.  All must give the oracle's pivot sequence."""
    _run("gpu-dual", 2, 64, 128, 2, 100000, **env)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,m,n,env", [
    (2, 64, 128, {"LPX_TABLEAU_BLOCK": "8"}), (3, 50, 121, {"LPX_TABLEAU_BLOCK": "32"}),
    (2, 64, 128, {"LPX_TABLEAU_BLOCK": "16", "LPX_TABLEAU_CONDENSED": "0"}),
    (3, 50, 121, {"LPX_TABLEAU_BLOCK": "8", "LPX_TABLEAU_CONDENSED": "0"})],
    ids=lambda v: ",".join(f"{k[12:]}={x}" for k, x in v.items()) if isinstance(v, dict) else str(v))
def test_dual_product_path_deferred_updates(nproc, m, n, env):
    """The column-partitioned DUAL loop with deferred updates (lpx_tableau_dual_block_*), forced on at test sizes, both
    layouts: the oracle's pivot sequences."""
    _run("gpu-dual", nproc, m, n, 2, 100000, LPX_TABLEAU_BLOCK_MIN_MB="0", **env)
