import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
import linprog_amd
from linprog_amd import problems
from linprog_amd.tableau import DualTableauSimplexSolver
for fused in ("1", "0"):
    os.environ["LPX_TABLEAU_FUSED"] = fused
    for (m, n) in ((1000, 2000), (300, 700), (3000, 6000)):
        c, A, b, basis = problems.gen_dual(m, n, 6)
        s = DualTableauSimplexSolver(c, A, b, basis)
        eng = s._engine
        from linprog_amd import _lib
        _lib.check(eng.lib.lpx_ctrl_begin_solve(eng.s.ctrl, torch.cuda.current_stream().cuda_stream))
        eng.enqueue(64); torch.cuda.synchronize()
        K = 256
        t = time.perf_counter(); eng.enqueue(K); torch.cuda.synchronize(); dt = time.perf_counter() - t
        ctrl = eng.read_ctrl()
        print(f"fused={fused} dual {m}x{n}: {K/dt:9.0f} pivots/s ({dt/K*1e6:6.2f} us/pivot) pivots={ctrl.pivots} status={ctrl.status}", flush=True)
