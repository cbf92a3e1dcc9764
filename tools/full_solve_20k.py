import sys, time
sys.path.insert(0, '/root/repo')
import torch, linprog_amd
from linprog_amd import problems
from linprog_amd.tableau import PrimalTableauSimplexSolver
m, n = 20000, 40000
A, b, c, basis = problems.gen_primal_columns(m, n, 0, 0, n - m)
s = PrimalTableauSimplexSolver(c, A, b, basis, columns=(0, n - m), n_global=n)
s._engine.CHUNK = 1024
t = time.time()
r = s.solve(200000)
torch.cuda.synchronize()
print("iters", r.iters, "optimum", r.optimum, "cost", r.cost, "seconds", time.time() - t)
