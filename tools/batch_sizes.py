#!/usr/bin/env python3
"""LPs/s of the batched primal solvers over a range of shapes: the revised-form kernels (wave kernel for m <= 64, generic
above; bit for bit against the oracle on a sample) and the condensed-tableau kernels (one wave up to 64 x 64 slots, 2 or 4
waves per LP up to 128 x 256; bit for bit against the oracle's condensed tableau on a sample).
GPU box: python3 tools/batch_sizes.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def bench_revised(torch, O, BatchedPrimalSolver, m, n, count, c, A, b, basis, reps):
    s = BatchedPrimalSolver(c, A, b, basis)
    basis0 = s.basis.clone()
    s.enqueue(100000)
    torch.cuda.synchronize()
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
    print(f"{m}x{n} x{count} revised: {count / dt:10.0f} LPs/s  {piv / dt / 1e6:7.1f} M pivots/s  {dt * 1e3:8.2f} ms   "
          f"oracle(16 thr) {k / dc:8.0f} LPs/s   bit-identical sample: {same}", flush=True)


def main():
    import torch

    from linprog_amd import problems
    from linprog_amd.batched import BatchedPrimalSolver, BatchedTableauSolver
    from oracle import oracle as O

    shapes = [(16, 32, 4096), (32, 64, 4096), (64, 128, 4096), (65, 130, 4096), (64, 256, 2048), (80, 170, 2048),
              (96, 192, 2048), (128, 256, 1024), (128, 384, 1024), (200, 300, 1024)]
    only_tab = os.environ.get("SIZES_TABLEAU_ONLY") == "1"
    for m, n, count in shapes:
        probs = [problems.gen_primal(m, n, s) for s in range(count)]
        c, A, b, basis = [np.stack([p[k] for p in probs]) for k in range(4)]
        reps = 3
        if not only_tab:
            try:
                bench_revised(torch, O, BatchedPrimalSolver, m, n, count, c, A, b, basis, reps)
            except Exception as e:  # beyond the LDS-resident kernels' shapes
                print(f"{m}x{n} x{count} revised: {str(e)[:110]}", flush=True)
        s = BatchedTableauSolver(c, A, b, basis)
        basis0 = s.basis.clone()
        s.enqueue(100000)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            s.basis.copy_(basis0)
            s.enqueue(100000)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        piv = int(s.pivots.sum().item())
        res = s.results()
        k = min(count, 8)
        same = True
        for i in range(k):
            o = O.OracleTableau(c[i], A[i], b[i], basis[i]).solve(100000)
            same = same and np.array_equal(res["x"][i], o.x) and np.array_equal(res["basis"][i], o.basis)
        print(f"{m}x{n} x{count} tableau: {count / dt:10.0f} LPs/s  {piv / dt / 1e6:7.1f} M pivots/s  {dt * 1e3:8.2f} ms   "
              f"{piv / dt * m * (n - m) / 1e12:6.2f} T elements/s   bit-identical to the condensed oracle (sample of {k}): {same}", flush=True)


if __name__ == "__main__":
    main()
