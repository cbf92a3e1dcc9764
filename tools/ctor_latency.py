"""Where a lone small LP's wall time goes (config 5): constructor, solve() and their sum, median of N repeats, per
problem of bench.py's config-5 set.  python tools/ctor_latency.py [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from linprog_amd import problems
from linprog_amd.dual_solvers import DualRevisedSimplexSolver
from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver
from linprog_amd.special_solvers import BoundedVariablePrimalSimplexSolver

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cases = []
for m in (16, 64, 128):
    cases.append((f"dual m={m}", DualRevisedSimplexSolver, problems.gen_dual(m, 2 * m, 0)))
    cases.append((f"bounded m={m}", BoundedVariablePrimalSimplexSolver, problems.gen_box(m, 2 * m, 0)))
    cases.append((f"primal m={m}", PrimalRevisedSimplexSolver, problems.gen_primal(m, 2 * m, 0)))
for name, cls, args in cases:
    tc, ts = [], []
    for _ in range(N):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s = cls(*args)
        t1 = time.perf_counter()
        r = s.solve(100000)
        t2 = time.perf_counter()
        tc.append(t1 - t0); ts.append(t2 - t1)
    print(f"{name:16s} arena={s._engine.arena is not None}  constructor {1e3 * np.median(tc):7.3f} ms  solve {1e3 * np.median(ts):7.3f} ms  "
          f"total {1e3 * (np.median(tc) + np.median(ts)):7.3f} ms  iters {r.iters}", flush=True)
