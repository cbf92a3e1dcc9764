"""General-form front end and two-phase driver (reference: linprog/simplex.py:10-102).

Host orchestration only; both phases run the device primal loop.
"""
import numpy as np

from .preprocessing import ProblemPreprocessingUtils
from .primal_solvers import PrimalRevisedSimplexSolver
from .special_solvers import PhaseOneSimplexSolver


class SimplexSolver(PrimalRevisedSimplexSolver):
    """min c'x  s.t.  Ax = b, Gx <= h, lb <= x <= ub  (simplex.py:11-77)."""

    def __init__(self, c, A=None, b=None, G=None, h=None, lb=None, ub=None):
        have_eq = A is not None and b is not None
        have_ineq = G is not None and h is not None
        self.num_slack_vars = 0 if G is None else np.asarray(G).shape[0]
        if have_eq and not have_ineq:
            A_, b_ = np.array(A), np.array(b)
        elif have_ineq and not have_eq:
            _, A_, b_ = ProblemPreprocessingUtils.canonical_form_to_standard_form(c, G, h)
        elif have_eq and have_ineq:
            # the reference raises TypeError here (np.zeros called with two positional sizes, simplex.py:57);
            # the intended stacking is [[A 0], [G I]]
            _, G_, h_ = ProblemPreprocessingUtils.canonical_form_to_standard_form(c, G, h)
            A_ = np.vstack([np.hstack([np.array(A), np.zeros((np.asarray(A).shape[0], self.num_slack_vars))]), G_])
            b_ = np.concatenate([np.array(b), np.array(h_)])
        else:
            raise ValueError("Input polyhedral misspcified.")
        c_ = np.concatenate([np.array(c), np.zeros(self.num_slack_vars)])
        nstruct = A_.shape[1] - self.num_slack_vars
        if lb is None:
            lb = np.repeat(0, nstruct)
        self.lb = np.concatenate([np.array(lb), np.repeat(0, self.num_slack_vars)])
        if ub is None:
            ub = np.repeat(np.inf, nstruct)
        self.ub = np.concatenate([np.array(ub), np.repeat(np.inf, self.num_slack_vars)])
        self.c, self.A, self.b = ProblemPreprocessingUtils.preprocess_problem(c_, A_, b_)
        self.num_vars = self.A.shape[1]

    def solve(self, maxiters1: int = 100, maxiters2: int = 100):
        """Phase one for a basis, then the primal loop from it (simplex.py:79-102)."""
        c1, A1, b1 = ProblemPreprocessingUtils.add_variables_bounds_to_coefficient_matrix(
            self.c, self.A, self.b, self.lb, self.ub)
        phase_one = PhaseOneSimplexSolver(c1, A1, b1)
        phase_one.solve(maxiters=maxiters1)
        solver = PrimalRevisedSimplexSolver(c1, A1, b1, phase_one.basis)
        res = solver.solve(maxiters=maxiters2)
        res.x = res.x[: self.num_vars - self.num_slack_vars]  # drop slack variables
        res.basis = None  # not interpretable without them (simplex.py:101)
        return res
