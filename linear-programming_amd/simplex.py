"""General-form front end and two-phase driver (reference: linprog/simplex.py:10-102).

Host orchestration only; both phases run the device primal loop.
"""
import numpy as np

from .preprocessing import ProblemPreprocessingUtils
from .primal_solvers import PrimalRevisedSimplexSolver
from .special_solvers import PhaseOneSimplexSolver


class SimplexSolver(PrimalRevisedSimplexSolver):
    """min c'x  s.t.  Ax = b, Gx <= h, lb <= x <= ub  (simplex.py:11-77).

    The standard form (slack columns, simplex.py:35-62 / preprocessing.py:17-28), the bound rows (:30-69) and the
    phase-one artificial block (special_solvers.py:48-52) are assembled on the device (``standard_form_on_device``);
    ``self.A`` downloads the standard-form matrix only when read."""

    def __init__(self, c, A=None, b=None, G=None, h=None, lb=None, ub=None):
        have_eq = A is not None and b is not None
        have_ineq = G is not None and h is not None
        if not (have_eq or have_ineq):
            raise ValueError("Input polyhedral misspcified.")
        self.num_slack_vars = 0 if G is None else np.asarray(G).shape[0]
        # (when both are given the reference raises TypeError: np.zeros called with two positional sizes, simplex.py:57;
        # the intended stacking [[A 0], [G I]] is what is built here)
        self._inputs = (np.array(c), A if have_eq else None, b if have_eq else None, G if have_ineq else None,
                        h if have_ineq else None)
        self._dev_problem = ProblemPreprocessingUtils.standard_form_on_device(*self._inputs)
        self.c, self.b = self._dev_problem.c32, self._dev_problem.b32
        nstruct = self._dev_problem.n - self.num_slack_vars
        if lb is None:
            lb = np.repeat(0, nstruct)
        self.lb = np.concatenate([np.array(lb), np.repeat(0, self.num_slack_vars)])
        if ub is None:
            ub = np.repeat(np.inf, nstruct)
        self.ub = np.concatenate([np.array(ub), np.repeat(np.inf, self.num_slack_vars)])
        self.num_vars = self._dev_problem.n

    def solve(self, maxiters1: int = 100, maxiters2: int = 100):
        """Phase one for a basis, then the primal loop from it (simplex.py:79-102)."""
        c, A, b, G, h = self._inputs
        nstruct = self.num_vars - self.num_slack_vars
        aux = ProblemPreprocessingUtils.standard_form_on_device(c, A, b, G, h, self.lb[:nstruct], self.ub[:nstruct],
                                                                artificials=True)
        phase_one = PhaseOneSimplexSolver(None, aux, None)
        phase_one.solve(maxiters=maxiters1)
        solver = PrimalRevisedSimplexSolver(None, aux.core, None, phase_one.basis)
        res = solver.solve(maxiters=maxiters2)
        res.x = res.x[: self.num_vars - self.num_slack_vars]  # drop slack variables
        res.basis = None  # not interpretable without them (simplex.py:101)
        return res
