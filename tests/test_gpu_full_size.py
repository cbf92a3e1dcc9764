"""GPU tests at BASELINE.json's full sizes, through size-independent properties (task rule 3).

config 4: m=20000, n=40000 tableau (6.4 GB): after K pivots the carried tableau must still satisfy
          T[:, basis] = I exactly, rc[basis] = 0 exactly, x_B >= 0, A x = b, cost = c.x, cost non-increasing,
          and the first pivots must equal the oracle's on the same workload.
config 3: 4096 LPs of 64 x 128: all optimal, A x = b, x >= 0, cost = c.x, reduced costs >= -1e-8 at the
          reported basis (optimality certificate computed independently in float64 on the host), plus a sample
          of LPs bit-identical to the oracle.
"""
import numpy as np
import pytest

from tests._util import gen

pytestmark = pytest.mark.gpu


def test_config4_tableau_invariants():
    import torch

    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    m, n, K = 20000, 40000, 24
    A, b, c, basis = gen.gen_primal_columns(m, n, 0, 0, n)
    cls = type("L", (PrimalTableauSimplexSolver,), {"_LOG_CAP": K + 1})
    s = cls(c, A, b, basis, columns=(0, n), n_global=n)
    costs = []
    for k in (8, 8, 8):
        res = s.solve(k)
        costs.append(float(res.cost))
        assert res.iters == k and res.optimum is False
    assert costs[0] >= costs[1] >= costs[2] and costs[2] < 0
    log = s.pivot_log()
    assert len(log) == K
    T = s._engine.T
    bas = torch.from_numpy(res.basis).cuda()
    # basic columns are exact unit vectors, basic reduced costs exactly 0
    cols = T[:, bas]                                   # (m+1) x m gather
    assert bool((cols[:m].diagonal() == 1.0).all())
    assert int((cols[:m] != 0).sum().item()) == m
    assert bool((cols[m] == 0.0).all())
    xB = T[:m, n].cpu().numpy()
    assert xB.min() >= 0.0
    x = res.x
    assert np.array_equal(x[res.basis], xB) and np.count_nonzero(x) <= m
    # A x = b and cost = c.x against the ORIGINAL float32 inputs, in float64 on the host
    resid = A[:, np.flatnonzero(x)].astype(np.float64) @ x[np.flatnonzero(x)] - b.astype(np.float64)
    assert np.abs(resid).max() < 1e-9
    assert abs(float(np.dot(c.astype(np.float64), x)) - res.cost) <= 1e-9 * max(1.0, abs(res.cost))
    assert len(set(res.basis.tolist())) == m
    # the oracle takes the same first pivots on the same workload (OpenMP update: ~0.1 s per pivot)
    kc = 6
    o = O.OracleTableau(c, A, b, basis, log_cap=kc + 1, threads=16).solve(kc)
    assert o.log.tolist() == log[:kc].tolist()


def test_config3_batch_full_size():
    from linprog_amd.batched import BatchedPrimalSolver
    from oracle import oracle as O

    B = 4096
    probs = [gen.gen_primal(64, 128, s) for s in range(B)]
    c = np.stack([p[0] for p in probs]); A = np.stack([p[1] for p in probs])
    b = np.stack([p[2] for p in probs]); basis = np.stack([p[3] for p in probs])
    solver = BatchedPrimalSolver(c, A, b, basis)
    res = solver.solve(100000)
    raw = solver.raw
    assert (raw["status"] == 1).all() and (raw["iters"] == raw["pivots"] + 1).all()
    A32 = A.astype(np.float32).astype(np.float64); b32 = b.astype(np.float32).astype(np.float64)
    c32 = c.astype(np.float32).astype(np.float64)
    x = raw["x"]
    assert x.min() >= -1e-12
    assert np.abs(np.einsum("bij,bj->bi", A32, x) - b32).max() < 1e-9
    assert np.abs(np.einsum("bj,bj->b", c32, x) - raw["cost"]).max() < 1e-9
    # independent optimality certificate: rc = c - c_B B^-1 A >= -1e-8 at the reported basis
    Bm = np.take_along_axis(A32, raw["basis"][:, None, :], axis=2)
    cB = np.take_along_axis(c32, raw["basis"], axis=1)
    y = np.linalg.solve(np.transpose(Bm, (0, 2, 1)), cB[..., None])[..., 0]
    rc = c32 - np.einsum("bi,bij->bj", y, A32)
    assert rc.min() >= -1e-8 - 1e-9
    for i in range(0, B, 293):  # a spread sample is bit-identical to the oracle
        o = O.OracleRevised(*probs[i], log_cap=4).solve(100000)
        assert np.array_equal(res[i].x, o.x) and res[i].cost == o.cost and res[i].iters == o.iters
