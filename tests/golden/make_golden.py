"""Generate golden vectors by importing the reference (agmuth/linear-programming) in the build container.

Run from the repo root, CPU only:

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--large]

Writes tests/golden/*.json.  A fixture holds DATA only: the generator id + seed (or the literal small
textbook inputs that the reference's tests/problems.py holds), and the reference's outputs -- pivot triples
(row, entering column, leaving variable), final basis, x, cost, iters, optimum / exception name.  The
reference itself (source or bytecode) is never copied; nothing under tests/ reads /root/reference at test time.

Pivot triples are captured by wrapping ``_update_basis`` (non-invasive monkey-patch).
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")

# load the build's own seeded generators without importing the (GPU-only) package __init__
_spec = importlib.util.spec_from_file_location("lpx_problems", os.path.join(ROOT, "linear-programming_amd", "problems.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)

from linprog import primal_solvers, dual_solvers, special_solvers, simplex  # noqa: E402
from tests import problems as ref_problems  # noqa: E402  (reference's tests/problems.py: data only)


class Trace:
    """Records (row, entering, leaving var) for every _update_basis call of any solver instance."""

    def __init__(self):
        self.events = []
        self._orig = primal_solvers.PrimalNaiveSimplexSolver._update_basis

    def __enter__(self):
        tr = self

        def wrapped(self_, row, col):
            tr.events.append([int(row), int(col), int(self_.basis[row])])
            return tr._orig(self_, row, col)

        primal_solvers.PrimalNaiveSimplexSolver._update_basis = wrapped
        return self

    def __exit__(self, *a):
        primal_solvers.PrimalNaiveSimplexSolver._update_basis = self._orig


def _res(res, events, exc=None):
    out = {"pivots": events, "exception": exc}
    if res is not None:
        out.update(
            x=[float(v) for v in res.x],
            basis=None if res.basis is None else [int(v) for v in res.basis],
            cost=float(res.cost),
            iters=int(res.iters),
            optimum=None if res.optimum is None else bool(res.optimum),
        )
    return out


def run_base(cls, c, A, b, basis, maxiters):
    with Trace() as tr:
        try:
            s = cls(c, A, b, basis)
        except Exception as e:  # constructor feasibility errors
            return _res(None, [], type(e).__name__)
        try:
            res = s.solve(maxiters)
        except Exception as e:
            return _res(None, tr.events, type(e).__name__)
    return _res(res, tr.events)


def lst(a):
    return None if a is None else np.asarray(a).tolist()


def textbook_cases():
    out = []
    base = {
        "PrimalNaiveSimplexSolver": primal_solvers.PrimalNaiveSimplexSolver,
        "PrimalRevisedSimplexSolver": primal_solvers.PrimalRevisedSimplexSolver,
    }
    for name in [k for k in vars(ref_problems) if k.startswith("primal_standard_form_lpp")]:
        p = getattr(ref_problems, name)
        for sn, cls in base.items():
            out.append(dict(problem=name, solver=sn, c=lst(p.c), A=lst(p.A), b=lst(p.b), basis=lst(p.starting_basis),
                            optimal_bfs=lst(p.optimal_bfs), optimal_basis=lst(p.optimal_basis), maxiters=100,
                            ref=run_base(cls, p.c, p.A, p.b, p.starting_basis, 100)))
    p = ref_problems.primal_solver_blands_sequence_problem1
    for sn, cls in base.items():
        out.append(dict(problem="primal_solver_blands_sequence_problem1", solver=sn, c=lst(p.c), A=lst(p.A), b=lst(p.b),
                        basis=lst(p.basis_seq[0]), basis_seq=lst(p.basis_seq), maxiters=100,
                        ref=run_base(cls, p.c, p.A, p.b, p.basis_seq[0], 100)))
    dual = {
        "DualNaiveSimplexSolver": dual_solvers.DualNaiveSimplexSolver,
        "DualRevisedSimplexSolver": dual_solvers.DualRevisedSimplexSolver,
    }
    for name in [k for k in vars(ref_problems) if k.startswith("dual_standard_form_lpp")]:
        p = getattr(ref_problems, name)
        for sn, cls in dual.items():
            out.append(dict(problem=name, solver=sn, c=lst(p.c), A=lst(p.A), b=lst(p.b), basis=lst(p.starting_basis),
                            optimal_bfs=lst(p.optimal_bfs), optimal_basis=lst(p.optimal_basis), maxiters=100,
                            ref=run_base(cls, p.c, p.A, p.b, p.starting_basis, 100)))
    # primal-dual algorithm (inner restricted-primal pivots are traced too)
    for name in [k for k in vars(ref_problems) if k.startswith("primal_dual_test_problem")]:
        p = getattr(ref_problems, name)
        with Trace() as tr:
            res = special_solvers.PrimalDualAlgorithm(p.c, p.A, p.b).solve()
        out.append(dict(problem=name, solver="PrimalDualAlgorithm", c=lst(p.c), A=lst(p.A), b=lst(p.b),
                        optimal_bfs=lst(p.optimal_bfs), ref=_res(res, tr.events)))
    # general two-phase driver on the primal problems + the infeasible LP of tests/test_general_solvers.py:37-49
    for name in [k for k in vars(ref_problems) if k.startswith("primal_standard_form_lpp")]:
        p = getattr(ref_problems, name)
        with Trace() as tr:
            res = simplex.SimplexSolver(p.c, p.A, p.b).solve()
        out.append(dict(problem=name, solver="SimplexSolver", c=lst(p.c), A=lst(p.A), b=lst(p.b),
                        optimal_bfs=lst(p.optimal_bfs), optimal_basis=lst(p.optimal_basis), ref=_res(res, tr.events)))
    c = np.array([1, 1, 1, 0, 0, 0, 0, 0])
    A = np.array([[1, 0, 0, 3, 2, 1, 0, 0], [0, 1, 0, 5, 1, 1, 1, 0], [0, 0, 1, 2, 5, 1, 0, 1], [0, 0, 0, 1, 1, 1, 0, 0]])
    b = np.array([1, 3, 4, -1])
    with Trace() as tr:
        try:
            simplex.SimplexSolver(c, A, b).solve()
            r = _res(None, tr.events, None)
        except Exception as e:
            r = _res(None, tr.events, type(e).__name__)
    out.append(dict(problem="infeasible_general", solver="SimplexSolver", c=lst(c), A=lst(A), b=lst(b), ref=r))
    # bounded-variable primal simplex, Bazaraa ex. 5.6 (tests/test_simplex_with_bounded_variables.py:9-16)
    c = np.array([-2, -4, -1, 0, 0]); b = np.array([10, 4])
    A = np.array([[2, 1, 1, 1, 0], [1, 1, -1, -0, 1]])
    lb = np.array([0, 0, 1, 0, 0]); ub = np.array([4, 6, 4, np.inf, np.inf])
    out.append(dict(problem="bazaraa_5_6", solver="BoundedVariablePrimalSimplexSolver", c=lst(c), A=lst(A), b=lst(b),
                    lb=lst(lb), ub=[4, 6, 4, "inf", "inf"], basis=[3, 4], lb_nonbasic=[0, 1, 2], ub_nonbasic=[],
                    ref=run_bounded(c, A, b, lb, ub, [3, 4], [0, 1, 2], [], 100)))
    return out


def run_bounded(c, A, b, lb, ub, basis, lbn, ubn, maxiters):
    """Per-iteration trace by repeated solve(maxiters=1) on one object (state persists between calls)."""
    s = special_solvers.BoundedVariablePrimalSimplexSolver(c, A, b, lb, ub, np.array(basis), np.array(lbn), np.array(ubn))
    steps = []  # one event per iteration: [row, entering, leaving var, leaving goes to ub?] or [-1, flipped var, -1, now at ub?]
    res = None
    try:
        for _ in range(maxiters):
            basis0 = s.basis.copy()
            ub0 = set(int(v) for v in s.ub_nonbasic_vars)
            res = s.solve(maxiters=1)
            ub1 = set(int(v) for v in s.ub_nonbasic_vars)
            if res.optimum:
                break
            ch = np.flatnonzero(basis0 != s.basis)
            if ch.size:
                r = int(ch[0])
                steps.append([r, int(s.basis[r]), int(basis0[r]), int(int(basis0[r]) in ub1)])
            else:
                (j,) = ub0 ^ ub1
                steps.append([-1, int(j), -1, int(j in ub1)])
    except Exception as e:
        return dict(steps=steps, exception=type(e).__name__)
    out = _res(res, [])
    out["steps"] = steps
    out["iters_total"] = len(steps) + 1
    out["ub_nonbasic_final"] = sorted(int(v) for v in s.ub_nonbasic_vars)
    out["lb_nonbasic_final"] = sorted(int(v) for v in s.lb_nonbasic_vars)
    out["ub_final"] = [float(v) if np.isfinite(v) else "inf" for v in s.ub]
    return out


def diet_case():
    """examples/diet.py (config 1): literal inputs transcribed from the example's data tables."""
    spec = importlib.util.spec_from_file_location("ref_diet", "/root/reference/examples/diet.py")
    # the example is a script (solves at import); capture its solver inputs and result by running it
    import io
    import contextlib
    mod = importlib.util.module_from_spec(spec)
    buf = io.StringIO()
    with Trace() as tr, contextlib.redirect_stdout(buf):
        spec.loader.exec_module(mod)
    names = {k: v for k, v in vars(mod).items() if isinstance(v, np.ndarray)}
    res = [v for v in vars(mod).values() if type(v).__name__ == "LinProgResult"][0]
    return dict(problem="diet", solver="SimplexSolver", arrays={k: lst(v) for k, v in names.items()},
                ref=_res(res, tr.events), stdout=buf.getvalue())


def random_cases(large):
    out = []
    PR, DR = primal_solvers.PrimalRevisedSimplexSolver, dual_solvers.DualRevisedSimplexSolver
    PN, DN = primal_solvers.PrimalNaiveSimplexSolver, dual_solvers.DualNaiveSimplexSolver

    def add(kind, solver, cls, g, m, n, seed, maxiters, keep_x=True):
        t0 = time.time()
        c, A, b, basis = g(m, n, seed)
        r = run_base(cls, c, A, b, basis, maxiters)
        if not keep_x:
            r.pop("x", None)
        out.append(dict(gen=kind, solver=solver, m=m, n=n, seed=seed, maxiters=maxiters, ref=r))
        print(f"  {solver} {kind} m={m} seed={seed}: {len(r['pivots'])} pivots, {time.time() - t0:.1f}s", flush=True)

    for m in (4, 8, 16, 32):
        for seed in range(4):
            add("gen_primal", "PrimalRevisedSimplexSolver", PR, gen.gen_primal, m, 2 * m, seed, 100000)
    for seed in range(8):
        add("gen_primal", "PrimalRevisedSimplexSolver", PR, gen.gen_primal, 64, 128, seed, 100000)
    add("gen_primal", "PrimalRevisedSimplexSolver", PR, gen.gen_primal, 128, 256, 0, 100000)
    add("gen_primal", "PrimalRevisedSimplexSolver", PR, gen.gen_primal, 48, 160, 3, 100000)   # n != 2m
    add("gen_primal", "PrimalRevisedSimplexSolver", PR, gen.gen_primal, 64, 128, 0, 50)       # maxiters hit: optimum False
    for m in (8, 32):
        for seed in range(2):
            add("gen_primal", "PrimalNaiveSimplexSolver", PN, gen.gen_primal, m, 2 * m, seed, 100000)
    for m in (4, 8, 16, 32, 64):
        for seed in range(3):
            add("gen_dual", "DualRevisedSimplexSolver", DR, gen.gen_dual, m, 2 * m, seed, 100000)
    add("gen_dual", "DualRevisedSimplexSolver", DR, gen.gen_dual, 128, 256, 0, 100000)
    add("gen_dual", "DualRevisedSimplexSolver", DR, gen.gen_dual, 64, 128, 0, 20)             # optimum None
    for seed in range(2):
        add("gen_dual", "DualNaiveSimplexSolver", DN, gen.gen_dual, 16, 32, seed, 100000)
    # constructor errors: primal-infeasible / dual-infeasible starting basis
    add("gen_dual", "PrimalRevisedSimplexSolver", PR, gen.gen_dual, 8, 16, 0, 100)
    add("gen_primal", "DualRevisedSimplexSolver", DR, gen.gen_primal, 8, 16, 0, 100)
    # unbounded: flip the sign of G so that the entering direction has no positive entry
    c, A, b, basis = gen.gen_primal(8, 16, 0)
    A[:, :8] *= -1
    out.append(dict(gen="gen_primal_negG", solver="PrimalRevisedSimplexSolver", m=8, n=16, seed=0, maxiters=100,
                    ref=run_base(PR, c, A, b, basis, 100)))
    # dual unbounded (primal infeasible): nonnegative row with negative right-hand side, c >= 0
    for m_, seed_ in ((8, 0), (16, 1)):
        c, A, b, basis = gen.gen_primal(m_, 2 * m_, seed_)
        c = np.abs(c)
        b[3] = -1.5
        out.append(dict(gen="gen_dual_unbounded", solver="DualRevisedSimplexSolver", m=m_, n=2 * m_, seed=seed_,
                        maxiters=1000, ref=run_base(DR, c, A, b, basis, 1000)))
        with Trace() as tr:
            try:
                res = special_solvers.PrimalDualAlgorithm(c, A, b).solve(1000, 1000)
                r = _res(res, tr.events)
            except Exception as e:
                r = _res(None, tr.events, type(e).__name__)
        out.append(dict(gen="gen_dual_unbounded", solver="PrimalDualAlgorithm", m=m_, n=2 * m_, seed=seed_, ref=r))
    # odd shapes (padding paths of the device layouts)
    for m_, n_, seed_ in ((1, 2, 0), (3, 5, 1), (5, 11, 2), (7, 9, 3), (33, 67, 4)):
        add("gen_primal", "PrimalRevisedSimplexSolver", PR, gen.gen_primal, m_, n_, seed_, 100000)
        add("gen_dual", "DualRevisedSimplexSolver", DR, gen.gen_dual, m_, n_, seed_, 100000)
    # naive solvers at the batched kernels' size (float32 re-inversion per pivot)
    add("gen_primal", "PrimalNaiveSimplexSolver", PN, gen.gen_primal, 64, 128, 0, 100000)
    add("gen_dual", "DualNaiveSimplexSolver", DN, gen.gen_dual, 64, 128, 0, 100000)
    if large:
        add("gen_primal", "PrimalRevisedSimplexSolver", PR, gen.gen_primal, 256, 512, 0, 100000, keep_x=False)
        add("gen_primal", "PrimalRevisedSimplexSolver", PR, gen.gen_primal, 1000, 2000, 0, 400, keep_x=False)
        add("gen_primal", "PrimalRevisedSimplexSolver", PR, gen.gen_primal, 2000, 4000, 0, 60, keep_x=False)
    return out


def gen_columns(m, n, seed):
    """bench.py's generator (float32 column blocks) in the (c, A, b, basis) order of the others."""
    A, b, c, basis = gen.gen_primal_columns(m, n, seed, 0, n)
    return c, A, b, basis


def xlarge_cases():
    """Prefixes on matrices large enough for the code paths that only engage on big inputs (staged pricing at A >= 64 MB,
    the fused update + y pass, non-temporal streams, 4-KiB row alignment, multi-strip column gathers), and the bench's own
    generator at three sizes.  The reference needs about 1 s per pivot at m = 5000 and 8 s at m = 10000 here, so these are prefixes (maxiters hit)."""
    out = []
    PR, DR = primal_solvers.PrimalRevisedSimplexSolver, dual_solvers.DualRevisedSimplexSolver

    def add(kind, solver, cls, g, m, n, seed, maxiters):
        t0 = time.time()
        c, A, b, basis = g(m, n, seed)
        r = run_base(cls, c, A, b, basis, maxiters)
        r.pop("x", None)
        out.append(dict(gen=kind, solver=solver, m=m, n=n, seed=seed, maxiters=maxiters, ref=r))
        print(f"  {solver} {kind} m={m} seed={seed}: {len(r['pivots'])} pivots, {time.time() - t0:.1f}s", flush=True)

    add("gen_primal_columns", "PrimalRevisedSimplexSolver", PR, gen_columns, 64, 128, 0, 100000)
    add("gen_primal_columns", "PrimalRevisedSimplexSolver", PR, gen_columns, 1000, 2000, 0, 400)
    add("gen_primal_columns", "PrimalRevisedSimplexSolver", PR, gen_columns, 5000, 10000, 0, 120)
    add("gen_dual", "DualRevisedSimplexSolver", DR, gen.gen_dual, 5000, 10000, 0, 120)
    add("gen_primal_columns", "PrimalRevisedSimplexSolver", PR, gen_columns, 10000, 20000, 0, 24)  # bench.py's revised10k
    return out


def degenerate_cases():
    """Integer data with ties (degenerate pivots, equal reduced costs): first-index tie-breaking decides the sequence.
    Primal only: on the dual counterpart (integer data, b <= 0) the reference's tolerance-free dual ratio test pivots on
    rounding noise -- its pivot sequences there depend on BLAS summation order and 3 of 9 of its "optimal" answers
    disagree with HiGHS, so they cannot serve as golden vectors."""
    out = []
    PR = primal_solvers.PrimalRevisedSimplexSolver
    for m, n in ((4, 8), (8, 16), (12, 30), (16, 32), (32, 64), (64, 128)):
        for seed in range(3):
            c, A, b, basis = gen.gen_primal_degenerate(m, n, seed)
            r = run_base(PR, c, A, b, basis, 5000)
            out.append(dict(gen="gen_primal_degenerate", solver="PrimalRevisedSimplexSolver", m=m, n=n, seed=seed,
                            maxiters=5000, ref=r))
            print(f"  gen_primal_degenerate m={m} seed={seed}: {len(r['pivots'])} pivots, exc={r['exception']}", flush=True)
    return out


def bounded_cases():
    out = []
    for m in (16, 64, 128, 170):
        for seed in range(2):
            t0 = time.time()
            c, A, b, lb, ub, basis, lbn, ubn = gen.gen_box(m, 2 * m, seed)
            r = run_bounded(c, A, b, lb, ub, basis, lbn, ubn, 100000)
            out.append(dict(gen="gen_box", solver="BoundedVariablePrimalSimplexSolver", m=m, n=2 * m, seed=seed, ref=r))
            print(f"  bounded m={m} seed={seed}: {r.get('iters_total')} iters, {time.time() - t0:.1f}s", flush=True)
    return out


PHASE_ONE_PROBLEMS = [
    # small-integer LPs with redundant (duplicated / summed) equations: an artificial variable stays basic at level zero
    # after phase one and is pivoted out or its row dropped (special_solvers.py:61-89); every pivot is exact
    ([[1, 0, 0, 1, 0, 0], [0, 1, 0, 0, 1, 0], [0, 0, 1, 0, 0, 1], [1, 1, 0, 1, 1, 0]], [2, 3, 1, 5]),
    ([[1, 1, 0, 0], [1, 1, 0, 0], [0, 0, 1, 1]], [2, 2, 3]),
    ([[1, 2, 1, 0, 0], [2, 4, 2, 0, 0], [0, 1, 0, 1, 0], [1, 0, 0, 0, 1]], [4, 8, 3, 2]),
    ([[1, 1, 1, 0], [1, -1, 0, 1], [2, 0, 1, 1]], [4, 0, 4]),
    ([[1, 0, 1, 0, 2], [0, 1, 0, 1, 1], [1, 1, 1, 1, 3], [2, 1, 2, 1, 5]], [3, 2, 5, 8]),
    ([[2, 1, 0, 1], [4, 2, 0, 2], [0, 1, 1, 0]], [6, 12, 2]),
]


def phase_one_cases():
    """PhaseOneSimplexSolver (special_solvers.py:11-91) on LPs with redundant rows: the basis handed to phase two, the
    rows kept (A, b after the drop) and every pivot incl. the drive-out pivots."""
    out = []
    for A, b in PHASE_ONE_PROBLEMS:
        A, b = np.array(A, dtype=float), np.array(b, dtype=float)
        with Trace() as tr:
            p1 = special_solvers.PhaseOneSimplexSolver(np.zeros(A.shape[1]), A.copy(), b.copy())
            try:
                p1.solve(1000)
                exc = None
            except Exception as e:
                exc = type(e).__name__
        out.append(dict(problem="phase_one_redundant", solver="PhaseOneSimplexSolver", A=A.tolist(), b=b.tolist(),
                        ref=dict(pivots=tr.events, exception=exc, basis=[int(v) for v in p1.basis], m=int(p1.m),
                                 A=np.asarray(p1.A, dtype=float).tolist(), b=np.asarray(p1.b, dtype=float).tolist())))
        print(f"  phase one {A.shape}: basis {out[-1]['ref']['basis']}, rows kept {p1.m}, exc={exc}", flush=True)
    return out


class DualStepSpy:
    """What the dual step of PrimalDualAlgorithm.solve really computes with (special_solvers.py:405, :430-433): the dtype
    of every matrix handed to np.linalg.inv (the restricted primal's constructor inverts twice in float32; the dual step's
    FRESH inverse is float64, because np.hstack([A float32, np.eye(m)]) at :411-413 is a float64 array) and the vector
    y @ A of every admissibility test (:405) -- the last one carries every dual step of the run."""

    def __enter__(self):
        self.inv_dtypes, self.adm_sizes, self.yA = [], [], None
        self._inv, self._isclose = np.linalg.inv, np.isclose

        def inv(a):
            self.inv_dtypes.append((sys._getframe(1).f_code.co_name, str(np.asarray(a).dtype)))  # (calling function, dtype)
            return self._inv(a)

        def isclose(a, b, *args, **kw):
            out = self._isclose(a, b, *args, **kw)
            if np.ndim(a) == 1 and np.ndim(b) == 1 and len(a) > 1:  # :405 (the optimality tests compare scalars)
                self.yA = np.array(a, dtype=float)
                self.adm_sizes.append(int(out.sum()))
            return out

        np.linalg.inv, np.isclose = inv, isclose
        return self

    def __exit__(self, *a):
        np.linalg.inv, np.isclose = self._inv, self._isclose


def primal_dual_cases():
    out = []
    for m in (4, 8, 16, 32, 64, 128):  # 64: the size bench.py --workload config5 times; 128: beyond the one-launch kernel
        for seed in range(2):
            t0 = time.time()
            c, A, b, _ = gen.gen_dual(m, 2 * m, seed)
            with Trace() as tr, DualStepSpy() as spy:
                try:
                    res = special_solvers.PrimalDualAlgorithm(c, A, b).solve(100000, 100000)
                    r = _res(res, tr.events)
                except Exception as e:
                    r = _res(None, tr.events, type(e).__name__)
            # per outer iteration: the inner solver's __init__ and _check_basis_feasibility, then solve's own at :433
            r["dual_step_inverse_dtypes"] = sorted(set(d for f, d in spy.inv_dtypes if f == "solve"))
            r["constructor_inverse_dtypes"] = sorted(set(d for f, d in spy.inv_dtypes if f != "solve"))
            r["adm_sizes"] = spy.adm_sizes
            r["yA_last"] = lst(spy.yA)
            out.append(dict(gen="gen_dual", solver="PrimalDualAlgorithm", m=m, n=2 * m, seed=seed, ref=r))
            print(f"  primal-dual m={m} seed={seed}: iters={r.get('iters')} dual-step inverse {r['dual_step_inverse_dtypes']} "
                  f"{time.time() - t0:.1f}s", flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--large", action="store_true", help="also m=256 full solve and m=1000/2000 prefixes (minutes)")
    ap.add_argument("--xlarge", action="store_true", help="also the m=5000 prefixes and the bench generator's goldens (minutes)")
    ap.add_argument("--only", default=None, help="write only this fixture file (e.g. random.json)")
    args = ap.parse_args()
    meta = dict(numpy=np.__version__, python=sys.version.split()[0],
                note="generated by importing the reference in the build container; data only")

    def dump(name, cases):
        if args.only and name != args.only:
            return
        cases = cases()
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(dict(meta=meta, cases=cases), f)
        print("wrote", name, len(cases), "cases", flush=True)

    dump("textbook.json", textbook_cases)
    dump("diet.json", lambda: [diet_case()])
    dump("bounded.json", bounded_cases)
    dump("primal_dual.json", primal_dual_cases)
    dump("phase_one.json", phase_one_cases)
    dump("random.json", lambda: random_cases(False))
    dump("degenerate.json", degenerate_cases)
    if args.large:
        dump("random_large.json", lambda: random_cases(True)[-3:])
    if args.xlarge:
        dump("random_xlarge.json", xlarge_cases)


if __name__ == "__main__":
    main()
