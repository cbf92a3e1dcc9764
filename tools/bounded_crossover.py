import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from linprog_amd import problems, _lib
from linprog_amd.engine import RevisedEngine
from linprog_amd.special_solvers import BoundedVariablePrimalSimplexSolver
for m in (80, 96, 112, 128, 144):
    args = problems.gen_box(m, 2 * m, 0)
    for lim in (0, 4096):
        RevisedEngine.SMALL_MAX_ROWS[_lib.KIND_BOUNDED] = lim
        ts = []
        for _ in range(8):
            s = BoundedVariablePrimalSimplexSolver(*args)
            fits = s._engine.fits_small(_lib.KIND_BOUNDED)
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = s.solve(100000); ts.append(time.perf_counter() - t0)
        print(f"bounded m={m} one-launch={fits} solve {1e3*np.median(ts):.3f} ms iters {r.iters} per iter {1e6*np.median(ts)/r.iters:.1f} us", flush=True)
