"""Special solvers behind the reference's class API (linprog/special_solvers.py:11-480).

* ``BoundedVariablePrimalSimplexSolver`` -- bounded pricing (float32 reduced costs, largest-coefficient rule),
  three-way ratio test, bound flips: all on the device (LPX_KIND_BOUNDED).
* ``PrimalDualAlgorithm`` -- outer loop on the host, every array operation of an outer iteration on the device:
  admissible mask, restricted primal (= full [A | I] with inadmissible columns excluded, solved by the primal
  loop from the artificial basis), restricted dual, theta reduction, dual update.
* ``PhaseOneSimplexSolver`` -- host orchestration around the device primal solver.
"""
import numpy as np

from . import _lib
from .data_classes import LinProgResult
from .device import stream_ptr, to_device, torch_cuda
from .engine import RevisedEngine
from .exceptions import BasisIsDualInfeasibleError, DualIsUnboundedError, PrimalIsInfeasibleError, PrimalIsUnboundedError
from .preprocessing import DeviceProblem, ProblemPreprocessingUtils
from .dual_solvers import DualRevisedSimplexSolver
from .primal_solvers import PrimalRevisedSimplexSolver, _initial_inverse
from .utils import get_bounds_on_bfs


class PhaseOneSimplexSolver(PrimalRevisedSimplexSolver):
    """Phase I of the two-phase method (special_solvers.py:11-91): finds a starting basis, sets ``.basis``.

    ``A`` may be a ``preprocessing.DeviceProblem`` built with ``artificials=True``: the auxiliary problem ``[A | I]`` is
    then in HBM already (one kernel, no host hstack) and phase one never sees the matrix on the host."""

    def __init__(self, c, A, b):
        if isinstance(A, DeviceProblem):
            assert A.artificial_from is not None, "phase one needs the artificial block (standard_form_on_device(artificials=True))"
            self._aux = A
            self._dev_problem = A.core
            self.c, self.b = A.core.c32, A.core.b32
            self.m, self.n = A.core.shape
        else:
            self._aux = None
            self.c, self.A, self.b = self._preprocess_problem(c, A, b)
            self.m, self.n = self.A.shape
        self.row_offset = 1
        self.col_offset = 1
        self.basis = None

    def solve(self, maxiters: int = 100):
        basis = np.arange(self.n, self.n + self.m)
        if self._aux is not None:
            solver = PrimalRevisedSimplexSolver(None, self._aux, None, basis)  # the hot path: device primal loop
        else:
            c = np.hstack([np.zeros(self.n), np.ones(self.m)])
            A = np.hstack([self.A, np.eye(self.m)])
            b = np.array(self.b)
            solver = PrimalRevisedSimplexSolver(c, A, b, basis)  # the hot path: device primal loop
        res = solver.solve(maxiters)

        if res.cost > 0:  # special_solvers.py:55-59
            if res.optimum:
                raise PrimalIsInfeasibleError
            raise ValueError("Phase one did not converge.")

        if res.basis.max() >= self.n:
            # artificial variables still basic at zero level: pivot them out where a positive entry of the
            # tableau row allows it (special_solvers.py:61-80)
            for art in range(self.n, self.n + self.m):
                if art not in res.basis:
                    continue
                row = int(np.argmax(res.basis == art))
                candidates = np.array([j for j in range(self.n) if j not in res.basis])
                tableau_row = solver._engine.tableau_row(row)[candidates]  # B^-1[row, :] @ A on the device (:69-71)
                usable = tableau_row > 0
                if usable.any():
                    solver.pivot(row, candidates[np.argmax(usable)])
                    res = solver._get_solver_return_object()

        if res.basis.max() >= self.n:
            # remaining artificials mark redundant equations (special_solvers.py:82-89, `<=` kept as is)
            keep = res.basis <= self.n
            if self._dev_problem is not None:
                self._dev_problem = self._dev_problem.select_rows(keep)
            else:
                self.A = self.A[keep]
            self.b = self.b[keep]
            self.m = keep.sum()
            res.basis = res.basis[keep]

        self.basis = res.basis


class BoundedVariablePrimalSimplexSolver(PrimalRevisedSimplexSolver):
    """Primal simplex for lb <= x <= ub without bound rows (special_solvers.py:94-343)."""

    _KIND = _lib.KIND_BOUNDED
    _NAIVE = False

    def __init__(self, c, A, b, lb, ub, basis, lb_nonbasic_vars, ub_nonbasic_vars):
        self.c, self.A, self.b = self._preprocess_problem(c, A, b)
        self.lb = np.array(lb, dtype=np.float64)
        self.ub = np.array(ub, dtype=np.float64)
        self.m, self.n = self.A.shape
        self.row_offset = 1
        self.col_offset = 1
        self.basis = np.array(basis).astype(int)
        lbn = np.array(lb_nonbasic_vars).astype(int)
        ubn = np.array(ub_nonbasic_vars).astype(int)
        # unconstrained variables get the largest magnitude any bfs can reach (special_solvers.py:139-142)
        M_bounds = get_bounds_on_bfs(self.A, self.b)
        self.lb[self.lb == -np.inf] = -M_bounds
        self.ub[self.ub == np.inf] = M_bounds
        inv32, _ = _initial_inverse(self.A, self.basis, self.b)
        var_state = np.full(self.n, _lib.VS_EXCLUDED, dtype=np.int32)
        var_state[lbn] = _lib.VS_LOWER
        var_state[ubn] = _lib.VS_UPPER
        var_state[self.basis] = _lib.VS_BASIC
        self._engine = RevisedEngine(self.c, self.A, self.b, self.basis, inv32, np.zeros(self.m), var_state=var_state,
                                     lb=self.lb, ub=self.ub, log_cap=self._LOG_CAP)
        self._engine.bounded_update_bfs()  # special_solvers.py:145
        self.counter = None
        self.optimum = None

    @property
    def lb_nonbasic_vars(self):
        """Nonbasic variables at their lower bound (ascending; the reference keeps insertion order)."""
        return np.flatnonzero(self._engine.host_var_state() == _lib.VS_LOWER)

    @property
    def ub_nonbasic_vars(self):
        return np.flatnonzero(self._engine.host_var_state() == _lib.VS_UPPER)

    @property
    def bfs(self):
        return self._engine.host_xB()  # float64 from the start

    def pivot(self, *args, **kwargs):
        raise NotImplementedError(
            "`pivot` operation/method not decoupled from `solve` method for this class."
        )  # special_solvers.py:216-219

    def _raise_for_status(self, status):
        if status == _lib.PRIMAL_UNBOUNDED:
            raise PrimalIsUnboundedError  # special_solvers.py:256-257 / :302-303
        if status == _lib.NOT_NONBASIC:
            raise ValueError("Column to enter basis is not nonbasic.")  # :331


class BoundedVariableDualSimplexSolver(BoundedVariablePrimalSimplexSolver):
    """OPT-IN EXTENSION, no counterpart in the reference: the DUAL simplex for ``lb <= x <= ub`` without bound rows
    (BASELINE.json config 5: "box-bounded random LP via dual simplex with bounded variables").  The reference has the
    bounded PRIMAL solver (special_solvers.py:94-343) and a dual solver for ``x >= 0`` (dual_solvers.py:47-108) only, so
    parity is unpinned against it; the rules are the reference's dual rules generalised (include/lpx.h,
    LPX_KIND_DUAL_BOUNDED), the arithmetic is checked bit for bit against the oracle's restatement
    (lpo_dual_bounded_solve) and the answers against HiGHS.

    Same constructor as the bounded primal solver.  The start must be DUAL feasible -- reduced cost >= 0 for the
    nonbasic variables at their lower bound, <= 0 for those at their upper bound, else ``BasisIsDualInfeasibleError``
    (dual_solvers.py:14-16 read for two bounds; float32 like the reference's check) -- and may be primal infeasible.
    Infinite bounds stay infinite.  ``solve`` raises ``DualIsUnboundedError`` when the primal is infeasible (:41-44).
    One launch per ``solve()`` call (lpx_revised_solve_small): the state must fit one CU's LDS (m <= 128 at n = 2m)."""

    _KIND = _lib.KIND_DUAL_BOUNDED

    def __init__(self, c, A, b, lb, ub, basis, lb_nonbasic_vars, ub_nonbasic_vars):
        self.c, self.A, self.b = self._preprocess_problem(c, A, b)
        self.lb = np.array(lb, dtype=np.float64)
        self.ub = np.array(ub, dtype=np.float64)
        self.m, self.n = self.A.shape
        self.row_offset = 1
        self.col_offset = 1
        self.basis = np.array(basis).astype(int)
        lbn = np.array(lb_nonbasic_vars).astype(int)
        ubn = np.array(ub_nonbasic_vars).astype(int)
        if np.any(np.isinf(self.lb[lbn])) or np.any(np.isinf(self.ub[ubn])):
            raise ValueError("a nonbasic variable cannot sit at an infinite bound")
        inv32, _ = _initial_inverse(self.A, self.basis, self.b)
        rc32 = self.c - self.c[self.basis] @ inv32 @ self.A  # dual_solvers.py:14-16, float32
        if np.any(rc32[lbn] < 0) or np.any(rc32[ubn] > 0):
            raise BasisIsDualInfeasibleError
        var_state = np.full(self.n, _lib.VS_EXCLUDED, dtype=np.int32)
        var_state[lbn] = _lib.VS_LOWER
        var_state[ubn] = _lib.VS_UPPER
        var_state[self.basis] = _lib.VS_BASIC
        self._engine = RevisedEngine(self.c, self.A, self.b, self.basis, inv32, np.zeros(self.m), var_state=var_state,
                                     lb=self.lb, ub=self.ub, log_cap=self._LOG_CAP)
        if not self._engine.fits_small(self._KIND):
            raise ValueError(f"the bounded-variable dual simplex is a one-launch kernel: the state of a {self.m} x {self.n} LP "
                             "does not fit one CU's LDS (m <= 128 at n = 2m)")
        self._engine.bounded_update_bfs()
        self.counter = None
        self.optimum = None

    def solve(self, maxiters: int = 100):
        """``optimum`` stays ``None`` until optimality is reached, as for the reference's dual solvers (dual_solvers.py:95-102)."""
        self.counter = 0
        ctrl = self._engine.run(self._KIND, maxiters)
        self.counter = int(ctrl.counter)
        if ctrl.status == _lib.DUAL_UNBOUNDED:
            self._sync_basis()
            raise DualIsUnboundedError
        if ctrl.status == _lib.OPTIMAL:
            self.optimum = True
        return self._get_solver_return_object()


class PrimalDualAlgorithm(DualRevisedSimplexSolver):
    """Primal-dual algorithm (special_solvers.py:346-480).

    As in the reference it derives from DualRevisedSimplexSolver (``isinstance`` checks agree) but overrides both
    ``__init__`` and ``solve`` and uses nothing else of it: no engine of the base class is ever built.
    """

    def __init__(self, c, A, b):
        self.c, self.A, self.b = ProblemPreprocessingUtils.preprocess_problem(c, A, b)
        self.m, self.n = self.A.shape
        self.counter = 0
        self.optimum = False
        self.inner_pivots = 0  # basis changes of all restricted-primal solves so far
        self.inner_pivot_logs = []  # (row, entering, leaving) in [A | I] numbering, when _LOG_CAP > 0
        self.admissible_sets = []

    _LOG_CAP = 0

    def solve(self, maxiters1: int = 100, maxiters2: int = 100, warm_start: bool = False):
        """``warm_start=True`` is an opt-in extension: the restricted primal keeps its optimal basis,
        B^-1 and x_B from one outer iteration to the next instead of the reference's cold restart from the artificial
        basis (special_solvers.py:408-425).  Its basic columns have zero reduced cost in the restricted primal, so they
        stay admissible after the dual step and the old basis is a feasible start.  Same optimum, fewer inner pivots,
        different pivot counts than the reference: never the default."""
        torch = torch_cuda()
        lib = _lib.load()
        y0 = np.zeros(self.m)
        expanded = False
        if self.c.min() < 0:  # special_solvers.py:388-400 (big-M row so that a dual feasible start exists)
            expanded = True
            bounding_M = get_bounds_on_bfs(self.A, self.b)
            self.c = np.hstack([self.c, np.zeros(1)])
            self.A = np.vstack([np.hstack([self.A, np.zeros((self.m, 1))]), np.ones((1, self.n + 1))])
            self.b = np.hstack([self.b, self.n * bounding_M * np.ones(1)])
            self.m, self.n = self.A.shape
            y0 = np.hstack([y0, self.c.min() * np.ones(1)])
        m, n = self.m, self.n

        # restricted primal = [A | I] with a per-column admissible mask; costs 0 / 1; artificial starting basis
        c_rp, A_rp, b_rp = ProblemPreprocessingUtils.preprocess_problem(
            np.hstack([np.zeros(n), np.ones(m)]), np.hstack([self.A, np.eye(m)]), self.b)
        art_basis = np.arange(n, n + m)
        eng = RevisedEngine(c_rp, A_rp, b_rp, art_basis, np.eye(m, dtype=np.float32), b_rp, log_cap=self._LOG_CAP)
        y = to_device(y0.astype(np.float64))
        c64 = to_device(np.asarray(self.c, dtype=np.float64))
        v = torch.zeros(n, dtype=torch.float64, device="cuda")
        out = torch.zeros(3, dtype=torch.float64, device="cuda")
        eye = torch.eye(m, dtype=torch.float64, device="cuda")
        b_dev = to_device(b_rp.astype(np.float64))
        art_dev = to_device(art_basis.astype(np.int64))
        st = stream_ptr()
        import ctypes as C

        x_rp = None
        inner_basis = art_basis
        solved = False
        # Small problems: the whole algorithm is ONE launch (lpx_pd_solve_small: same steps in the same order, the inner
        # solves bit-identical to the oracle's primal loop).  The per-outer-iteration logs exist only on the host path.
        one_launch = (not warm_start and not self._LOG_CAP and eng.SMALL_SOLVE and m <= eng.SMALL_MAX_ROWS[_lib.KIND_PRIMAL]
                      and lib.lpx_pd_small_lds_bytes(m, n) <= eng.SMALL_LDS_MAX)
        if one_launch and self.counter < maxiters1:
            eng.touch()
            _lib.check(lib.lpx_ctrl_begin_solve(eng.s.ctrl, st))
            _lib.check(lib.lpx_pd_solve_small(C.byref(eng.s), y.data_ptr(), c64.data_ptr(), n, int(maxiters1 - self.counter),
                                              int(maxiters2), eng.x.data_ptr(), eng.cost.data_ptr(), st))
            ctrl = eng.read_ctrl()
            self.counter += int(ctrl.counter)
            self.inner_pivots += int(ctrl.pivots)
            if ctrl.status == _lib.PRIMAL_UNBOUNDED:
                raise PrimalIsUnboundedError
            if ctrl.status == _lib.DUAL_UNBOUNDED:
                raise DualIsUnboundedError  # :435-436
            self.optimum = ctrl.status == _lib.OPTIMAL
            solved = True
        while not one_launch and self.counter < maxiters1:
            self.counter += 1
            eng.touch()  # this loop writes the engine's tensors itself
            _lib.check(lib.lpx_pd_admissible(C.byref(eng.s), y.data_ptr(), c64.data_ptr(), n, v.data_ptr(), st))  # :405
            if warm_start and self.counter > 1:
                # keep basis / B^-1 / x_B; artificials that left the basis stay available at their bound, and the
                # basic columns are basic whatever isclose() says about them to the last bit
                eng.var_state[n:] = _lib.VS_LOWER
                eng.var_state[eng.basis] = _lib.VS_BASIC
            else:
                # fresh restricted-primal solver state (:408-424): artificial basis, B^-1 = I, bfs = b, float32 phase
                eng.var_state[n:] = _lib.VS_BASIC
                eng.Binv.zero_()
                eng.Binv[:, :m].copy_(eye)
                eng.xB.copy_(b_dev)
                eng.basis.copy_(art_dev)
                _lib.check(lib.lpx_ctrl_init(eng.s.ctrl, 1, st))
            if self._LOG_CAP:
                self.admissible_sets.append(np.flatnonzero(eng.host_var_state()[:n] == _lib.VS_LOWER))
            ctrl = eng.run(_lib.KIND_PRIMAL, maxiters2)  # :425
            self.inner_pivots += int(ctrl.counter) - (1 if ctrl.status == _lib.OPTIMAL else 0)
            if ctrl.status == _lib.PRIMAL_UNBOUNDED:
                raise PrimalIsUnboundedError
            if self._LOG_CAP:
                self.inner_pivot_logs.append(eng.host_log(ctrl.pivots))
            # x, cost of the restricted primal (:72-79) and the dual step (:427-444), gated on the device by that cost:
            # one read of (theta, umax, cost) is the only other host round trip of an outer iteration
            _lib.check(lib.lpx_revised_expand(C.byref(eng.s), eng.x.data_ptr(), eng.cost.data_ptr(), st))
            _lib.check(lib.lpx_pd_dual_step(C.byref(eng.s), y.data_ptr(), c64.data_ptr(), v.data_ptr(), n,
                                            out.data_ptr(), eng.cost.data_ptr(), st))
            theta, umax, cost_rp = out.cpu().numpy()
            solved = True
            if cost_rp > 0.0:
                if umax <= 0:
                    raise DualIsUnboundedError  # :435-436
            else:
                self.optimum = True
                break  # complementary slackness attained

        if solved:
            x_rp = eng.x.cpu().numpy()
            inner_basis = eng.host_basis()
        x = np.zeros(n) if x_rp is None else x_rp[:n].copy()
        basis = inner_basis[inner_basis < n]
        cost = np.dot(self.c, x)  # :464
        if expanded:
            basis = basis[basis != n - 1]
            x = x[:-1]
        self.y = y.cpu().numpy()
        return LinProgResult(x=x, basis=basis, cost=cost, iters=self.counter, optimum=self.optimum)
