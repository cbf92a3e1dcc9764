"""Worker of the batch-sharding tests (one process per rank, launched by torch.distributed.run).

mode "cpu": the host protocol alone -- shard(), ResultGather's packing / all-gather / unpacking of ragged shards -- with
            the oracle standing in for the device kernel, gloo on CPU tensors.
mode "gpu": the product path (ShardedBatch around BatchedTableauSolver / BatchedPrimalSolver) with every rank on cuda:0
            and the gloo backend staging the one all-gather through the host (rehearsal on the one-GPU box).
Both compare the gathered batch with a single-rank solve of the whole batch, bit for bit, on every rank.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import linprog_amd  # noqa: E402,F401
from linprog_amd import problems  # noqa: E402
from linprog_amd.batched import ResultGather, shard  # noqa: E402
from oracle import oracle as O  # noqa: E402

KEYS = ("x", "cost", "basis", "status", "iters", "pivots")


def main():
    mode, count, m, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    probs = [problems.gen_primal(m, n, 4000 + s) for s in range(count)]
    c, A, b, basis = [np.stack([p[k] for p in probs]) for k in range(4)]
    lo, hi = shard(count, rank, world)
    cover = [shard(count, g, world) for g in range(world)]
    assert cover[0][0] == 0 and cover[-1][1] == count and all(cover[g][1] == cover[g + 1][0] for g in range(world - 1))
    if mode == "cpu":
        whole = O.primal_solve_batch(c, A, b, basis, 100000, threads=2)
        mine = O.primal_solve_batch(c[lo:hi], A[lo:hi], b[lo:hi], basis[lo:hi], 100000, threads=2)
        rg = ResultGather(dist.group.WORLD, count, m, n, "cpu")
        rg.pack(*(torch.from_numpy(np.ascontiguousarray(mine[k])) for k in KEYS))
        rg.all_gather()
        got = rg.unpack()
        for k in KEYS:
            assert np.array_equal(got[k], whole[k]), f"rank {rank}: {k} differs after the gather"
    else:
        from linprog_amd.batched import BatchedPrimalSolver, BatchedTableauSolver, ShardedBatch

        torch.cuda.set_device(0)
        for cls in (BatchedTableauSolver, BatchedPrimalSolver):
            one = cls(c, A, b, basis)
            one.enqueue(100000)
            whole = one.results()
            sb = ShardedBatch(cls, c, A, b, basis, dist.group.WORLD)
            assert (sb.lo, sb.hi) == (lo, hi)
            got = sb.solve(100000)
            for k in KEYS:
                assert np.array_equal(got[k], whole[k]), f"rank {rank}: {cls.__name__} {k} differs from the single-rank run"
            # every rank passing only its own slice gives the same
            sl = ShardedBatch(cls, c[lo:hi], A[lo:hi], b[lo:hi], basis[lo:hi], dist.group.WORLD, local=True)
            got = sl.solve(100000)
            for k in KEYS:
                assert np.array_equal(got[k], whole[k])
        assert (whole["status"] == 1).all()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
