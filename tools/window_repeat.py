"""Is the short window's fixed cost (time = a + b K, a = 2.2 ms) paid per call or once per LP state?  Enqueues the headline
leg's K steps several times in a row with a synchronisation between the calls and prints each call's time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from linprog_amd import _lib, problems
from linprog_amd.tableau import PrimalTableauSimplexSolver

m, n = 20000, 40000
A, b, c, basis = problems.gen_primal_columns(m, n, 0, 0, n - m)
S = type("B", (PrimalTableauSimplexSolver,), {"_LOG_CAP": 8})
solver = S(c, A, b, basis, group=None, columns=(0, n - m), n_global=n)
eng = solver._engine
lib = _lib.load()
_lib.check(lib.lpx_ctrl_begin_solve(eng.s.ctrl, torch.cuda.current_stream().cuda_stream))
eng.enqueue(160)
torch.cuda.synchronize()
for K in [20, 20, 20, 1, 2, 3, 4, 8, 20, 40, 20]:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.enqueue(32 * K)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"K {K:3d}: {1e3 * (t2 - t0):8.3f} ms  (host enqueue {1e3 * (t1 - t0):6.3f} ms)  per step {1e3 * (t2 - t0) / K:7.4f}", flush=True)
