"""Dual simplex solvers behind the reference's class API (linprog/dual_solvers.py:9-120)."""
import numpy as np

from . import _lib
from .exceptions import BasisIsDualInfeasibleError, DualIsUnboundedError
from .primal_solvers import PrimalNaiveSimplexSolver, PrimalRevisedSimplexSolver


class DualNaiveSimplexSolver(PrimalNaiveSimplexSolver):
    """Dual simplex, first-negative-row leaving rule, textbook dual ratio test (dual_solvers.py:9-108)."""

    _KIND = _lib.KIND_DUAL

    def _check_basis_feasibility(self, inv32, bfs32):
        # dual_solvers.py:14-16 -> primal_solvers.py:50-53 (float32, strict)
        if not np.all(self.c[self.basis] @ inv32 @ self.A <= self.c):
            raise BasisIsDualInfeasibleError

    def _raise_for_status(self, status):
        if status == _lib.DUAL_UNBOUNDED:
            raise DualIsUnboundedError  # dual_solvers.py:41-44

    def solve(self, maxiters: int = 100):
        """Dual simplex loop (dual_solvers.py:82-108).  ``optimum`` is never reset to False (quirk kept:
        a non-converged dual solve returns ``optimum=None``)."""
        self.counter = 0
        ctrl = self._engine.run(self._device_kind(), maxiters)
        self.counter = int(ctrl.counter)
        if ctrl.status not in (_lib.RUNNING, _lib.OPTIMAL):
            self._sync_basis()
            self._raise_for_status(ctrl.status)
        if ctrl.status == _lib.OPTIMAL:
            self.optimum = True
        return self._get_solver_return_object()


class DualRevisedSimplexSolver(PrimalRevisedSimplexSolver, DualNaiveSimplexSolver):
    """Revised dual simplex (dual_solvers.py:111-120): DualNaive's loop with the revised pivot."""

    _KIND = _lib.KIND_DUAL
