"""Where does a lone small solve spend its time?  Constructor vs solve() of the drop-in classes on config-5 problems,
one-launch kernel (default) vs the multi-kernel chain (LPX_SMALL_SOLVE=0), beside the single-thread oracle's loop."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import linprog_amd  # noqa: E402,F401
from linprog_amd import problems  # noqa: E402
from linprog_amd.dual_solvers import DualRevisedSimplexSolver  # noqa: E402
from linprog_amd.engine import RevisedEngine  # noqa: E402
from linprog_amd.special_solvers import BoundedVariablePrimalSimplexSolver  # noqa: E402
from oracle import oracle as O  # noqa: E402


def timed(make, reps=5):
    tc = ts = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s = make()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r = s.solve(100000)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tc += t1 - t0
        ts += t2 - t1
    return tc / reps * 1e3, ts / reps * 1e3, r


for m in (16, 64, 128):
    box = problems.gen_box(m, 2 * m, 0)
    du = problems.gen_dual(m, 2 * m, 0)
    for name, mk, mo in (("bounded", lambda: BoundedVariablePrimalSimplexSolver(*box), lambda: O.OracleBounded(*box)),
                         ("dual", lambda: DualRevisedSimplexSolver(*du), lambda: O.OracleRevised(*du, kind="dual", log_cap=4))):
        timed(mk, 2)
        RevisedEngine.SMALL_SOLVE = True
        c1, s1, r = timed(mk)
        RevisedEngine.SMALL_SOLVE = False
        c0, s0, _ = timed(mk)
        RevisedEngine.SMALL_SOLVE = True
        o = mo()
        t0 = time.perf_counter()
        ro = o.solve(100000)
        to = (time.perf_counter() - t0) * 1e3
        print(f"{name} m={m}: iters {r.iters} | ctor {c1:.2f} ms | solve one-launch {s1:.3f} ms ({s1 * 1e3 / r.iters:.1f} us/iter) | "
              f"chain {s0:.3f} ms ({s0 * 1e3 / r.iters:.1f} us/iter) | oracle loop {to:.3f} ms ({to * 1e3 / ro.iters:.1f} us/iter)", flush=True)
