#!/usr/bin/env python3
"""Random shapes through the batched condensed-tableau kernels (one wave / wave grid, standard and inequality form) against
the oracle's tableau loop, bit for bit.  GPU box: python3 tools/fuzz_batched_tab.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from linprog_amd import _lib, problems
    from linprog_amd.batched import BatchedTableauSolver
    from oracle import oracle as O

    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    lib = _lib.load()
    bad = 0
    for it in range(cases):
        while True:
            m, k = int(rng.integers(1, 257)), int(rng.integers(1, 257))
            if lib.lpx_batched_tableau_waves(m, k):
                break
        n = m + k
        maxiters = int(rng.choice([100000, 100000, 5, 1, 0]))
        probs = [problems.gen_primal(m, n, 1000 * it + s) for s in range(3)]
        if rng.random() < 0.3:  # an unbounded LP in the batch
            probs[1][1][:, :k] *= -1
        args = [np.stack([p[q] for p in probs]) for q in range(4)]
        tab = BatchedTableauSolver(*args, log_cap=8192)
        rt = tab.solve(maxiters)
        ineq = BatchedTableauSolver.from_inequalities(args[0][:, :k], args[1][:, :, :k], args[2])
        ri = ineq.solve(maxiters)
        ok = True
        for i, p in enumerate(probs):
            o = O.OracleTableau(*p, log_cap=8192).solve(maxiters)
            ok = ok and tab.pivot_logs()[i].tolist() == o.log.tolist() and rt[i].iters == o.iters
            ok = ok and np.array_equal(rt[i].x, o.x) and rt[i].cost == o.cost and rt[i].basis.tolist() == o.basis.tolist()
            ok = ok and int(tab.raw["status"][i]) == int(o.status)
            ok = ok and np.array_equal(ri[i].x, rt[i].x) and ri[i].iters == rt[i].iters
        print(f"{it:3d}: m={m:3d} k={k:3d} waves={lib.lpx_batched_tableau_waves(m, k)} maxiters={maxiters:6d} "
              f"{'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
    print("failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
