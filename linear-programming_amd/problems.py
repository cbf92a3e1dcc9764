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


BLOCK_COLS = 250  # column-block granularity of gen_primal_columns (divides 40000/8, 2000/8, 128/8 is handled by ragged blocks)


def gen_primal_columns(m, n, seed, j0, j1):
    """Columns [j0, j1) of the large-scale variant of gen_primal, float32, reproducible under ANY column split.

    A = [G I], G ~ U(0,1) drawn in float32 per block of BLOCK_COLS columns from its own PCG64 stream
    (SeedSequence([seed, block])), so a rank that owns a column slice generates exactly its part of the one
    global matrix.  Not bit-identical to gen_primal (different draw order/dtype); same distribution.
    Returns (A_cols[m, j1-j0] float32, b[m] float32, c[n] float32, basis[m]).
    """
    k = n - m
    out = np.zeros((m, j1 - j0), dtype=np.float32)
    blk0, blk1 = j0 // BLOCK_COLS, (min(j1, k) - 1) // BLOCK_COLS if j0 < k else -1
    for blk in range(blk0, blk1 + 1):
        lo, hi = blk * BLOCK_COLS, min((blk + 1) * BLOCK_COLS, k)
        g = np.random.default_rng([seed, blk]).random((m, hi - lo), dtype=np.float32)
        a, bnd = max(lo, j0), min(hi, j1)
        out[:, a - j0:bnd - j0] = g[:, a - lo:bnd - lo]
    for j in range(max(j0, k), j1):  # identity part
        out[j - k, j - j0] = 1.0
    rng = np.random.default_rng([seed, 1 << 30])
    b = (1.0 + rng.random(m, dtype=np.float32)).astype(np.float32)
    c = np.concatenate([-rng.random(k, dtype=np.float32), np.zeros(m, dtype=np.float32)])
    return out, b, c, np.arange(k, n)


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


def gen_primal_degenerate(m, n, seed):
    """Small-integer data with zeros in b: ties in the ratio test (degenerate pivots) and equal reduced costs, the
    cases where "first index" tie-breaking (primal_solvers.py:137, :166) decides the pivot sequence."""
    rng = np.random.default_rng(seed)
    k = n - m
    G = rng.integers(0, 3, (m, k)).astype(float)
    A = np.hstack([G, np.eye(m)])
    b = rng.integers(0, 3, m).astype(float)
    c = np.concatenate([-rng.integers(0, 3, k).astype(float), np.zeros(m)])
    return c, A, b, np.arange(k, n)


def gen_box_dual(m, n, seed):
    """Box-bounded LP with a DUAL feasible, primal infeasible start for the bounded-variable dual simplex (BASELINE
    config 5's wording; the build's opt-in solver -- the reference has none): costs of both signs, every structural
    variable nonbasic at the bound its cost prefers (upper when c_j < 0, else lower), slack basis.  Feasible by
    construction (x = ub / 2 leaves positive slacks); the slacks are free above (+inf)."""
    rng = np.random.default_rng(seed)
    k = n - m
    G = rng.uniform(0, 1, (m, k))
    A = np.hstack([G, np.eye(m)])
    ubx = rng.uniform(0.5, 1.5, k)
    b = G @ (0.5 * ubx) + rng.uniform(0.1, 1, m)
    c = np.concatenate([rng.uniform(-1, 1, k), np.zeros(m)])
    lb = np.zeros(n)
    ub = np.concatenate([ubx, np.inf * np.ones(m)])
    basis = np.arange(k, n)
    at_upper = np.flatnonzero(c[:k].astype(np.float32) < 0)
    at_lower = np.flatnonzero(c[:k].astype(np.float32) >= 0)
    return c, A, b, lb, ub, basis, at_lower, at_upper


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
