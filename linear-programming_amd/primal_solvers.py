"""Primal simplex solvers behind the reference's class API (linprog/primal_solvers.py:10-263).

Same constructor, ``solve(maxiters)``, ``pivot(row, col)``, attributes and error behaviour as the reference;
the pivot loop itself runs on the GPU through liblpx (revised form: explicit B^-1 resident in HBM, rank-1
update instead of the reference's dense ``E @ B^-1``).  Constructor-time host work that the reference does
once with NumPy (float32 cast, ``np.linalg.inv`` of the starting basis, feasibility check) is done the same
way here, so the starting state is bit-identical to the reference's.
"""
import numpy as np

from . import _lib
from .data_classes import LinProgResult
from .engine import RevisedEngine
from .exceptions import BasisIsPrimalInfeasibleError, PrimalIsUnboundedError
from .preprocessing import DeviceProblem, ProblemPreprocessingUtils


def signed_identity_diagonal(B):
    """The diagonal of B if B is a diagonal matrix of +1 / -1 entries (a slack basis, possibly with rows flipped by
    preprocessing.py:6-15), else None."""
    m = B.shape[0]
    d = B[np.arange(m), np.arange(m)]
    if np.count_nonzero(B) == m and np.all(np.abs(d) == 1.0):
        return d.astype(np.float32)
    return None


def _initial_inverse_device(prob, basis):
    """``_initial_inverse`` for a problem whose matrix is on the device: the artificial basis of phase one is the identity
    by construction; any other basis costs an m x m download of its columns for the reference's float32 LAPACK inverse."""
    if prob.basis_is_artificial_identity(basis):
        return np.eye(prob.m, dtype=np.float32), prob.b32.copy()
    return _initial_inverse(prob.columns(basis), np.arange(prob.m), prob.b32)


def _initial_inverse(A32, basis, b32):
    """inv_basis_matrix and bfs of the constructor (primal_solvers.py:37-38), float32.

    A starting basis whose columns are the identity up to sign (the slack basis of every synthetic config; minus signs
    where preprocessing flipped a row) is recognised without LAPACK: ``np.linalg.inv`` of such a matrix is the matrix
    itself, exactly, so the result is the same and a 20k x 20k float32 inversion on the host is avoided.
    """
    m = A32.shape[0]
    B = A32[:, basis]
    d = signed_identity_diagonal(B)
    if d is not None:
        return np.diag(d), d * b32
    inv32 = np.linalg.inv(B)
    return inv32, inv32 @ b32


class PrimalNaiveSimplexSolver:
    """Primal simplex with first-negative-index entering rule (reference: primal_solvers.py:10-209).

    The naive variant re-inverts the basis after every pivot in float32 (primal_solvers.py:92-94); on the
    device that is the reinversion kernel (``lpx_reinvert``), the revised variant uses the rank-1 update.
    """

    _KIND = _lib.KIND_PRIMAL
    _NAIVE = True
    _LOG_CAP = 0

    def __init__(self, c, A, b, basis):
        """``A`` may be a ``preprocessing.DeviceProblem`` (matrix already in HBM, preprocessed): c and b are then taken
        from it and ``self.A`` downloads the matrix only if somebody reads it."""
        self._dev_problem = A if isinstance(A, DeviceProblem) else None
        if self._dev_problem is not None:
            self.c, self.b = A.c32, A.b32
            self.m, self.n = A.shape
        else:
            self.c, self._A, self.b = self._preprocess_problem(c, A, b)
            self.m, self.n = self._A.shape
        self.row_offset = 1
        self.col_offset = 1
        self.basis = np.array(basis).astype(int)
        if self._dev_problem is not None:
            inv32, bfs32 = _initial_inverse_device(A, self.basis)
        else:
            inv32, bfs32 = _initial_inverse(self._A, self.basis, self.b)
        self.counter = None
        self.optimum = None
        self._check_basis_feasibility(inv32, bfs32)
        self._engine = RevisedEngine(self.c, A if self._dev_problem is not None else self._A, self.b, self.basis, inv32,
                                     bfs32, log_cap=self._LOG_CAP, naive=self._NAIVE)

    @property
    def A(self):
        return self._dev_problem.host() if getattr(self, "_dev_problem", None) is not None else self._A

    @A.setter
    def A(self, value):
        self._dev_problem = None
        self._A = value

    # ------------------------------------------------------------------ constructor helpers
    def _preprocess_problem(self, c, A, b):
        return ProblemPreprocessingUtils.preprocess_problem(c, A, b)  # preprocessing.py:6-15

    def _check_basis_feasibility(self, inv32, bfs32):
        if not np.all(bfs32 >= 0.0):  # primal_solvers.py:43-48 (strict, float32, no tolerance)
            raise BasisIsPrimalInfeasibleError

    # ------------------------------------------------------------------ attributes other code reads
    @property
    def inv_basis_matrix(self):
        Binv = self._engine.host_Binv()
        return Binv.astype(np.float32) if self._engine.read_ctrl().f32_phase else Binv

    @property
    def bfs(self):
        x = self._engine.host_xB()
        return x.astype(np.float32) if self._engine.read_ctrl().f32_phase else x

    def _sync_basis(self):
        self.basis[:] = self._engine.host_basis()  # in place: results alias this array (primal_solvers.py:83-86)

    # ------------------------------------------------------------------ result
    def _get_bfs_expanded(self):
        return self._engine.expand()[0]

    def _calc_current_cost(self):
        return self._engine.expand()[1]

    def _get_solver_return_object(self):
        x, cost = self._engine.expand()
        self._sync_basis()
        return LinProgResult(x=x, basis=self.basis, cost=cost, iters=self.counter, optimum=self.optimum)

    def _get_reduced_costs(self):
        return self._engine.reduced_costs()

    # ------------------------------------------------------------------ pivot / loop
    def pivot(self, col_in_basis_to_leave_basis, col_in_A_to_enter_basis):
        """Basis change at a caller-chosen position (primal_solvers.py:169-174 / :255-263)."""
        self._engine.pivot(col_in_basis_to_leave_basis, col_in_A_to_enter_basis)
        self._sync_basis()

    def refine_inverse(self, steps=1):
        """Opt-in extension (no counterpart in the reference, which carries the product-form inverse for the whole solve):
        Newton-Schulz refinement of ``inv_basis_matrix`` and recomputation of ``bfs`` in float64 on the device.  Returns
        ``(max|I - B B^-1| before, after)``.  Meant for very long runs; ``solve()`` never calls it."""
        if self._NAIVE:
            raise ValueError("the naive solvers re-invert every pivot; there is nothing to refine")
        return self._engine.refine(steps=steps, update_xB=True)

    def reinvert(self):
        """Opt-in extension: ``inv_basis_matrix = inv(A[:, basis])`` from scratch in float64 on the device (what the naive
        solvers do per pivot in float32, primal_solvers.py:92-94) and ``bfs`` from it.  ``solve()`` never calls it."""
        if self._NAIVE:
            raise ValueError("the naive solvers re-invert every pivot")
        self._engine.reinvert(update_xB=True)

    def _raise_for_status(self, status):
        if status == _lib.PRIMAL_UNBOUNDED:
            raise PrimalIsUnboundedError  # primal_solvers.py:130-132

    def _device_kind(self):
        if self._NAIVE:
            return {_lib.KIND_PRIMAL: _lib.KIND_PRIMAL_NAIVE, _lib.KIND_DUAL: _lib.KIND_DUAL_NAIVE}[self._KIND]
        return self._KIND

    def solve(self, maxiters: int = 100):
        """Primal simplex loop (primal_solvers.py:176-209); resumes from the current device state."""
        self.counter = 0
        self.optimum = False
        ctrl = self._engine.run(self._device_kind(), maxiters)
        self.counter = int(ctrl.counter)
        if ctrl.status not in (_lib.RUNNING, _lib.OPTIMAL):
            self._sync_basis()
            self._raise_for_status(ctrl.status)
        self.optimum = ctrl.status == _lib.OPTIMAL
        return self._get_solver_return_object()

    def pivot_log(self):
        """(row, entering, leaving variable) triples recorded on the device (needs ``_LOG_CAP`` > 0)."""
        return self._engine.host_log(self._engine.read_ctrl().pivots)


class PrimalRevisedSimplexSolver(PrimalNaiveSimplexSolver):
    """Revised primal simplex: product-form update of B^-1 (reference: primal_solvers.py:212-263)."""

    _NAIVE = False
