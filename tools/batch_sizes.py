#!/usr/bin/env python3
"""LPs/s of the batched primal solver over a range of shapes (wave kernel for m <= 64, generic above), each checked bit for
bit against the oracle on a sample.  GPU box: python3 tools/batch_sizes.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from linprog_amd import problems
    from linprog_amd.batched import BatchedPrimalSolver
    from oracle import oracle as O

    shapes = [(16, 32, 4096), (32, 64, 4096), (64, 128, 4096), (64, 256, 2048), (80, 170, 1024), (96, 192, 1024)]
    for m, n, count in shapes:
        probs = [problems.gen_primal(m, n, s) for s in range(count)]
        c, A, b, basis = [np.stack([p[k] for p in probs]) for k in range(4)]
        s = BatchedPrimalSolver(c, A, b, basis)
        basis0 = s.basis.clone()
        s.enqueue(100000)
        torch.cuda.synchronize()
        reps = 3
        t = time.perf_counter()
        for _ in range(reps):
            s.basis.copy_(basis0)
            s.enqueue(100000)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        piv = int(s.pivots.sum().item())
        k = min(count, 64)
        t = time.perf_counter()
        r = O.primal_solve_batch(c[:k], A[:k], b[:k], basis[:k], 100000, threads=16)
        dc = time.perf_counter() - t
        res = s.results()
        same = np.array_equal(res["x"][:k], r["x"]) and np.array_equal(res["basis"][:k], r["basis"])
        print(f"{m}x{n} x{count}: {count / dt:10.0f} LPs/s  {piv / dt / 1e6:7.1f} M pivots/s  {dt * 1e3:8.2f} ms   "
              f"oracle(16 thr) {k / dc:8.0f} LPs/s   bit-identical sample: {same}", flush=True)


if __name__ == "__main__":
    main()
