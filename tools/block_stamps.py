#!/usr/bin/env python3
"""Where the cycles of the one-launch small-LP kernel go.  Needs liblpx built with -DLPX_BLOCK_STAMPS (diagnostic):
   make -C linear-programming_amd/csrc clean && make -C linear-programming_amd/csrc CXXFLAGS+=-DLPX_BLOCK_STAMPS -j8
GPU box: python3 tools/block_stamps.py; rebuild the library without the flag afterwards."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    import torch

    from linprog_amd import problems
    from linprog_amd.dual_solvers import DualRevisedSimplexSolver
    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver
    from linprog_amd.special_solvers import BoundedVariablePrimalSimplexSolver, PrimalDualAlgorithm

    for m in (64, 128):
        print(f"--- primal {m} x {2 * m}", flush=True)
        PrimalRevisedSimplexSolver(*problems.gen_primal(m, 2 * m, 0)).solve(100000)
        torch.cuda.synchronize()
        print(f"--- dual {m} x {2 * m}", flush=True)
        DualRevisedSimplexSolver(*problems.gen_dual(m, 2 * m, 0)).solve(100000)
        torch.cuda.synchronize()
    print("--- bounded 64 x 128", flush=True)
    BoundedVariablePrimalSimplexSolver(*problems.gen_box(64, 128, 0)).solve(100000)
    torch.cuda.synchronize()
    print("--- primal-dual 64 x 128", flush=True)
    PrimalDualAlgorithm(*problems.gen_dual(64, 128, 0)[:3]).solve(100000, 100000)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
