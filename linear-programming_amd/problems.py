"""Seeded synthetic LP generators used by tests, bench.py and the golden maker.

All use ``np.random.default_rng(seed)`` (PCG64) with the draw order listed in the survey; k = n - m.
They return float64 arrays; the solver constructors round to float32 (preprocessing.py:6-15).
"""
import numpy as np


def gen_primal(m, n, seed):
    """Feasible (slack basis, b>0) and bounded (G>0) LP: A=[G I], min c'x with c=[-U(0,1), 0]."""
    rng = np.random.default_rng(seed)
    k = n - m
    G = rng.uniform(0, 1, (m, k))
    A = np.hstack([G, np.eye(m)])
    b = rng.uniform(1, 2, m)
    c = np.concatenate([-rng.uniform(0, 1, k), np.zeros(m)])
    basis = np.arange(k, n)
    return c, A, b, basis


def gen_primal_f32(m, n, seed):
    """Same LP as gen_primal for large sizes: draws in float32 blocks to bound host memory.

    NOT bit-identical to gen_primal (different draw dtype); used only for the 20k x 40k bench workload where
    a float64 (m, k) draw would need 3.2 GB twice.  Values are float32-representable by construction.
    """
    rng = np.random.default_rng(seed)
    k = n - m
    A = np.zeros((m, n), dtype=np.float32)
    step = 2048
    for i0 in range(0, m, step):
        i1 = min(m, i0 + step)
        A[i0:i1, :k] = rng.random((i1 - i0, k), dtype=np.float32)
    A[np.arange(m), k + np.arange(m)] = 1.0
    b = (1.0 + rng.random(m, dtype=np.float32)).astype(np.float32)
    c = np.concatenate([-rng.random(k, dtype=np.float32), np.zeros(m, dtype=np.float32)])
    basis = np.arange(k, n)
    return c, A, b, basis


def gen_dual(m, n, seed):
    """Dual-feasible slack basis with b<0 (rows get flipped by preprocessing); c>=0 so also a PrimalDual input."""
    rng = np.random.default_rng(seed)
    k = n - m
    G = rng.uniform(0, 1, (m, k))
    A = np.hstack([-G, np.eye(m)])
    b = -rng.uniform(1, 2, m)
    c = np.concatenate([rng.uniform(0.1, 1, k), np.zeros(m)])
    basis = np.arange(k, n)
    return c, A, b, basis


def gen_box(m, n, seed):
    """Box-bounded LP for BoundedVariablePrimalSimplexSolver (reference limit: m <= 170)."""
    rng = np.random.default_rng(seed)
    k = n - m
    G = rng.uniform(0, 1, (m, k))
    A = np.hstack([G, np.eye(m)])
    ubx = rng.uniform(0.5, 1.5, k)
    b = G @ (0.5 * ubx) + rng.uniform(0.1, 1, m)
    c = np.concatenate([-rng.uniform(0, 1, k), np.zeros(m)])
    lb = np.zeros(n)
    ub = np.concatenate([ubx, 1e6 * np.ones(m)])
    basis = np.arange(k, n)
    lb_nonbasic = np.arange(k)
    ub_nonbasic = np.array([], dtype=int)
    return c, A, b, lb, ub, basis, lb_nonbasic, ub_nonbasic
