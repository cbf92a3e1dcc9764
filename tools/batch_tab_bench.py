#!/usr/bin/env python3
"""LPs/s of the two batched kernels (revised form / condensed tableau) on config 3: 4096 x gen_primal(64, 128).
GPU box: python3 tools/batch_tab_bench.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from linprog_amd import problems
    from linprog_amd.batched import BatchedPrimalSolver, BatchedTableauSolver

    for m, n, count in [(64, 128, 4096), (32, 64, 4096), (16, 32, 4096)]:
        probs = [problems.gen_primal(m, n, s) for s in range(count)]
        args = [np.stack([p[k] for p in probs]) for k in range(4)]
        out = {}
        for name, cls in (("revised", BatchedPrimalSolver), ("tableau", BatchedTableauSolver)):
            s = cls(*args)
            basis0 = s.basis.clone()
            s.enqueue(100000)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5):
                s.basis.copy_(basis0)
                s.enqueue(100000)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 5
            out[name] = s.results()
            print(f"{m}x{n} x{count} {name:8s}: {count / dt:10.0f} LPs/s  {int(s.pivots.sum()) / dt / 1e6:7.1f} M pivots/s  {dt * 1e3:7.2f} ms",
                  flush=True)
        same = np.array_equal(out["revised"]["basis"], out["tableau"]["basis"]) and np.array_equal(out["revised"]["iters"], out["tableau"]["iters"])
        print("   same bases and iteration counts:", same, " max |x diff|:", float(np.abs(out["revised"]["x"] - out["tableau"]["x"]).max()))


if __name__ == "__main__":
    main()
