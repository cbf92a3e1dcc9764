"""ctypes front end of the CPU restatement (oracle/lpx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package.  The classes mirror the reference's solver classes closely enough that the
parity tests read like the reference's own tests, and additionally return the pivot log.

Constructor-time work that the reference does with NumPy/LAPACK on the host (float32 cast, row flips,
``np.linalg.inv`` of the starting basis, primal_solvers.py:27-41) is done with the same NumPy calls here.
"""
import ctypes as C
import os
import subprocess
from math import factorial

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblpx_oracle.so")

RUNNING, OPTIMAL, PRIMAL_UNBOUNDED, DUAL_UNBOUNDED, NOT_NONBASIC = 0, 1, 2, 3, 4
VS_BASIC, VS_LOWER, VS_UPPER, VS_EXCLUDED = 0, 1, 2, 3

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int64)
_sp = C.POINTER(C.c_int32)


def build(force=False):
    """Compile liblpx_oracle.so with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "lpx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liblpx_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class _Revised(C.Structure):
    _fields_ = [
        ("m", C.c_int64), ("n", C.c_int64),
        ("A", _fp), ("b", _fp), ("c", _fp),
        ("Binv", _dp), ("xB", _dp), ("basis", _ip), ("var_state", _sp),
        ("f32_phase", C.c_int32), ("status", C.c_int32),
        ("counter", C.c_int64), ("pivots", C.c_int64),
        ("log", _ip), ("log_cap", C.c_int64),
        ("y", _dp), ("rc", _dp), ("d", _dp), ("e", _dp), ("rowr", _dp), ("row", _dp),
        ("lb", _dp), ("ub", _dp), ("w", _dp),
    ]


class _Tableau(C.Structure):
    _fields_ = [
        ("m", C.c_int64), ("n", C.c_int64), ("ld", C.c_int64),
        ("T", _dp), ("basis", _ip), ("var_state", _sp),
        ("f32_phase", C.c_int32), ("status", C.c_int32),
        ("counter", C.c_int64), ("pivots", C.c_int64),
        ("log", _ip), ("log_cap", C.c_int64),
        ("d", _dp), ("e", _dp), ("rowr", _dp), ("threads", C.c_int32), ("pricing", C.c_int32),
        ("ratio_rule", C.c_int32), ("pad_", C.c_int32), ("ratio_tol", C.c_double),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.lpo_primal_solve.argtypes = [C.POINTER(_Revised), C.c_int64]
        L.lpo_dual_solve.argtypes = [C.POINTER(_Revised), C.c_int64]
        L.lpo_bounded_solve.argtypes = [C.POINTER(_Revised), C.c_int64]
        L.lpo_revised_pivot.argtypes = [C.POINTER(_Revised), C.c_int64, C.c_int64]
        L.lpo_revised_pivot.restype = None
        L.lpo_bounded_update_bfs.argtypes = [C.POINTER(_Revised)]
        L.lpo_bounded_update_bfs.restype = None
        L.lpo_price.argtypes = [C.POINTER(_Revised)]
        L.lpo_price.restype = None
        L.lpo_bounded_price.argtypes = [C.POINTER(_Revised)]
        L.lpo_bounded_price.restype = None
        L.lpo_direction.argtypes = [C.POINTER(_Revised), C.c_int64]
        L.lpo_direction.restype = None
        L.lpo_dual_row.argtypes = [C.POINTER(_Revised), C.c_int64]
        L.lpo_dual_row.restype = None
        L.lpo_tableau_solve.argtypes = [C.POINTER(_Tableau), C.c_int64]
        L.lpo_tableau_dual_solve.argtypes = [C.POINTER(_Tableau), C.c_int64]
        L.lpo_tableau_solve_condensed.argtypes = [C.POINTER(_Tableau), _ip, C.c_int64]
        L.lpo_tableau_solve_condensed.restype = C.c_int
        L.lpo_tableau_dual_solve.restype = C.c_int
        L.lpo_rank1_update.argtypes = [_dp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _dp, _dp]
        L.lpo_rank1_update.restype = None
        L.lpo_rank1_update_mt.argtypes = L.lpo_rank1_update.argtypes
        L.lpo_rank1_update_mt.restype = None
        L.lpo_eta.argtypes = [_dp, C.c_int64, C.c_int64, _dp]
        L.lpo_eta.restype = None
        L.lpo_first_negative.argtypes = [_dp, _sp, C.c_int64, _dp, _ip]
        L.lpo_first_argmax.argtypes = [_dp, C.c_int64, _dp, _ip]
        L.lpo_ratio_primal.argtypes = [_dp, _dp, C.c_int64, C.c_int, _ip, _dp, _dp]
        L.lpo_ratio_primal.restype = None
        L.lpo_ratio_dual.argtypes = [_dp, _dp, _sp, C.c_int64, C.c_int, _ip, _dp, _dp]
        L.lpo_ratio_dual.restype = None
        L.lpo_ratio_bounded.argtypes = [_dp, _dp, _dp, _dp, C.c_int64, C.c_double, _dp, _ip, _dp, _ip]
        L.lpo_ratio_bounded.restype = None
        L.lpo_pd_admissible.argtypes = [_dp, _fp, _dp, C.c_int64, C.c_int64, C.c_int64, _dp, _sp]
        L.lpo_pd_admissible.restype = None
        L.lpo_pd_theta.argtypes = [_dp, _fp, _dp, _dp, _sp, C.c_int64, C.c_int64, C.c_int64, _dp, _dp, _dp]
        L.lpo_pd_theta.restype = None
        L.lpo_expand_and_cost.argtypes = [_dp, _ip, _fp, C.c_int64, C.c_int64, _dp]
        L.lpo_expand_and_cost.restype = C.c_double
        L.lpo_primal_solve_batch.argtypes = [C.c_int64, C.c_int64, C.c_int64, _fp, _fp, _fp, _fp, _ip, _dp, _dp, _sp, _ip,
                                             _ip, C.c_int64, C.c_int]
        L.lpo_primal_solve_batch.restype = None
        L.lpo_dual_solve_batch.argtypes = L.lpo_primal_solve_batch.argtypes
        L.lpo_dual_solve_batch.restype = None
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def preprocess(c, A, b):
    """preprocessing.py:6-15."""
    c = np.array(c).astype(np.float32)
    A = np.ascontiguousarray(np.array(A).astype(np.float32))
    b = np.array(b).astype(np.float32)
    neg = b < 0
    A[neg] *= -1
    b[neg] *= -1
    return c, A, b


# ------------------------------------------------------------------------------------------------ kernels
def rank1_update(M, r, e, cols=None, threads=1):
    """In-place pivot update of the row-major 2-D float64 array M (leading dimension = M.shape[1])."""
    assert M.dtype == np.float64 and M.flags.c_contiguous
    rows, ld = M.shape
    cols = ld if cols is None else cols
    e = np.ascontiguousarray(e, dtype=np.float64)
    scratch = np.empty(ld)
    f = lib().lpo_rank1_update_mt if threads > 1 else lib().lpo_rank1_update
    f(_p(M, _dp), rows, cols, ld, int(r), _p(e, _dp), _p(scratch, _dp))
    return M


def eta(d, r):
    d = np.ascontiguousarray(d, dtype=np.float64)
    e = np.empty_like(d)
    lib().lpo_eta(_p(d, _dp), d.size, int(r), _p(e, _dp))
    return e


def first_negative(rc, var_state=None):
    rc = np.ascontiguousarray(rc, dtype=np.float64)
    mn, j = C.c_double(), C.c_int64()
    vs = None
    if var_state is not None:
        var_state = np.ascontiguousarray(var_state, dtype=np.int32)
        vs = _p(var_state, _sp)
    opt = lib().lpo_first_negative(_p(rc, _dp), vs, rc.size, C.byref(mn), C.byref(j))
    return bool(opt), mn.value, j.value


def first_argmax(rc):
    rc = np.ascontiguousarray(rc, dtype=np.float64)
    mx, j = C.c_double(), C.c_int64()
    opt = lib().lpo_first_argmax(_p(rc, _dp), rc.size, C.byref(mx), C.byref(j))
    return bool(opt), mx.value, j.value


def ratio_primal(x, d, f32=False):
    x = np.ascontiguousarray(x, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    r, th, dm = C.c_int64(), C.c_double(), C.c_double()
    lib().lpo_ratio_primal(_p(x, _dp), _p(d, _dp), x.size, int(f32), C.byref(r), C.byref(th), C.byref(dm))
    return r.value, th.value, dm.value


def ratio_dual(rc, row, f32=False, var_state=None):
    rc = np.ascontiguousarray(rc, dtype=np.float64)
    row = np.ascontiguousarray(row, dtype=np.float64)
    vs = None
    if var_state is not None:
        var_state = np.ascontiguousarray(var_state, dtype=np.int32)
        vs = _p(var_state, _sp)
    j, th, rm = C.c_int64(), C.c_double(), C.c_double()
    lib().lpo_ratio_dual(_p(rc, _dp), _p(row, _dp), vs, rc.size, int(f32), C.byref(j), C.byref(th), C.byref(rm))
    return j.value, th.value, rm.value


def ratio_bounded(x, d, lbB, ubB, sign):
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, d, lbB, ubB)]
    g1, g2, r1, r2 = C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
    lib().lpo_ratio_bounded(*[_p(a, _dp) for a in arrs], arrs[0].size, float(sign),
                            C.byref(g1), C.byref(r1), C.byref(g2), C.byref(r2))
    return g1.value, r1.value, g2.value, r2.value


def expand_and_cost(xB, basis, c32, n):
    xB = np.ascontiguousarray(xB, dtype=np.float64)
    basis = np.ascontiguousarray(basis, dtype=np.int64)
    c32 = np.ascontiguousarray(c32, dtype=np.float32)
    x = np.empty(n)
    cost = lib().lpo_expand_and_cost(_p(xB, _dp), _p(basis, _ip), _p(c32, _fp), xB.size, n, _p(x, _dp))
    return x, cost


def solve_batch(c, A, b, basis, maxiters=100, threads=1, kind="primal"):
    """A batch of LPs of one shape (batch-major arrays) through lpo_primal_solve / lpo_dual_solve, one LP per OpenMP
    thread.  Inputs go through the reference's preprocessing per LP (float32, rows with b < 0 flipped,
    preprocessing.py:6-15); a starting basis that is not the identity gets its float32 inverse (primal_solvers.py:37).
    Returns dict(x, basis, cost, status, iters, pivots)."""
    A = np.array(A).astype(np.float32)
    b = np.array(b).astype(np.float32)
    c = np.ascontiguousarray(np.asarray(c).astype(np.float32))
    neg = b < 0
    A[neg] *= -1
    b[neg] *= -1
    A = np.ascontiguousarray(A)
    basis = np.ascontiguousarray(np.array(basis).astype(np.int64))
    B, m, n = A.shape
    Bm = np.take_along_axis(A, basis[:, None, :], axis=2)
    inv0 = None
    if not np.array_equal(Bm, np.broadcast_to(np.eye(m, dtype=np.float32), Bm.shape)):
        inv0 = np.ascontiguousarray(np.linalg.inv(Bm).astype(np.float32))
    x, cost = np.empty((B, n)), np.empty(B)
    status = np.empty(B, dtype=np.int32)
    iters, pivots = np.empty(B, dtype=np.int64), np.empty(B, dtype=np.int64)
    f = lib().lpo_primal_solve_batch if kind == "primal" else lib().lpo_dual_solve_batch
    f(B, m, n, _p(A, _fp), _p(b, _fp), _p(c, _fp), None if inv0 is None else _p(inv0, _fp), _p(basis, _ip), _p(x, _dp),
      _p(cost, _dp), _p(status, _sp), _p(iters, _ip), _p(pivots, _ip), int(maxiters), int(threads))
    return dict(x=x, basis=basis, cost=cost, status=status, iters=iters, pivots=pivots)


def primal_solve_batch(c, A, b, basis, maxiters=100, threads=1):
    return solve_batch(c, A, b, basis, maxiters, threads, "primal")


# ------------------------------------------------------------------------------------------------ solvers
class OracleResult(dict):
    __getattr__ = dict.__getitem__


class OracleRevised:
    """Restatement of PrimalRevisedSimplexSolver / DualRevisedSimplexSolver (primal_solvers.py:212-263,
    dual_solvers.py:111-120).  kind = "primal" | "dual"."""

    def __init__(self, c, A, b, basis, kind="primal", log_cap=1 << 16, var_state=None, check=True):
        self.kind = kind
        self.c, self.A, self.b = preprocess(c, A, b)
        self.m, self.n = self.A.shape
        self.basis = np.array(basis).astype(np.int64)
        inv32 = np.linalg.inv(self.A[:, self.basis])  # float32, primal_solvers.py:37
        bfs32 = inv32 @ self.b                        # :38
        if check:
            if kind == "primal" and not np.all(bfs32 >= 0.0):          # :43-48
                raise ValueError("BasisIsPrimalInfeasibleError")
            if kind == "dual" and not np.all(self.c[self.basis] @ inv32 @ self.A <= self.c):  # dual_solvers.py:14-16
                raise ValueError("BasisIsDualInfeasibleError")
        self.Binv = np.ascontiguousarray(inv32.astype(np.float64))
        self.xB = bfs32.astype(np.float64)
        if var_state is None:
            var_state = np.full(self.n, VS_LOWER, dtype=np.int32)
        self.var_state = np.ascontiguousarray(var_state, dtype=np.int32)
        self.var_state[self.basis] = VS_BASIC
        self.log = np.zeros((log_cap, 3), dtype=np.int64)
        m, n = self.m, self.n
        self._scr = {k: np.zeros(sz) for k, sz in
                     dict(y=m, rc=n, d=m, e=m, rowr=max(m, n), row=n, w=m).items()}
        self.lb = None
        self.ub = None
        self.s = _Revised()
        self._bind()
        self.s.f32_phase = 1
        self.s.pivots = 0

    def _bind(self):
        s = self.s
        s.m, s.n = self.m, self.n
        s.A, s.b, s.c = _p(self.A, _fp), _p(self.b, _fp), _p(self.c, _fp)
        s.Binv, s.xB, s.basis = _p(self.Binv, _dp), _p(self.xB, _dp), _p(self.basis, _ip)
        s.var_state = _p(self.var_state, _sp)
        s.log, s.log_cap = _p(self.log, _ip), self.log.shape[0]
        for k, a in self._scr.items():
            setattr(s, k, _p(a, _dp))
        if self.lb is not None:
            s.lb, s.ub = _p(self.lb, _dp), _p(self.ub, _dp)

    @property
    def pivots(self):
        return int(self.s.pivots)

    def pivot_log(self):
        return self.log[: min(self.pivots, self.log.shape[0])].copy()

    def pivot(self, row, col):
        lib().lpo_revised_pivot(C.byref(self.s), int(row), int(col))

    def _result(self, status):
        x, cost = expand_and_cost(self.xB, self.basis, self.c, self.n)
        return OracleResult(x=x, basis=self.basis.copy(), cost=cost, iters=int(self.s.counter),
                            status=status, pivots=self.pivots, log=self.pivot_log())

    def solve(self, maxiters=100):
        f = lib().lpo_primal_solve if self.kind == "primal" else lib().lpo_dual_solve
        status = f(C.byref(self.s), int(maxiters))
        return self._result(status)


class OracleBounded(OracleRevised):
    """Restatement of BoundedVariablePrimalSimplexSolver (special_solvers.py:94-343)."""

    def __init__(self, c, A, b, lb, ub, basis, lb_nonbasic_vars, ub_nonbasic_vars, log_cap=1 << 16):
        c32, A32, b32 = preprocess(c, A, b)
        n = A32.shape[1]
        vs = np.full(n, VS_EXCLUDED, dtype=np.int32)
        vs[np.array(lb_nonbasic_vars).astype(int)] = VS_LOWER
        vs[np.array(ub_nonbasic_vars).astype(int)] = VS_UPPER
        super().__init__(c, A, b, basis, kind="bounded", log_cap=log_cap, var_state=vs, check=False)
        self.lb = np.array(lb, dtype=np.float64)
        self.ub = np.array(ub, dtype=np.float64)
        M = bounds_on_bfs(self.A, self.b)  # special_solvers.py:140-142
        self.lb[self.lb == -np.inf] = -M
        self.ub[self.ub == np.inf] = M
        self._bind()
        lib().lpo_bounded_update_bfs(C.byref(self.s))  # :145

    def _result(self, status):
        x = np.zeros(self.n)  # special_solvers.py:201-206
        x[self.basis] += self.xB
        lo, up = self.var_state == VS_LOWER, self.var_state == VS_UPPER
        x[lo] += self.lb[lo]
        x[up] += self.ub[up]
        cost = float(np.dot(self.c.astype(np.float64), x))  # equals special_solvers.py:149-167 up to rounding
        return OracleResult(x=x, basis=self.basis.copy(), cost=cost, iters=int(self.s.counter), status=status,
                            pivots=self.pivots, log=self.pivot_log(), var_state=self.var_state.copy())

    def solve(self, maxiters=100):
        status = lib().lpo_bounded_solve(C.byref(self.s), int(maxiters))
        return self._result(status)


class OracleDualBounded(OracleBounded):
    """Restatement of the build's opt-in bounded-variable DUAL simplex (lpo_dual_bounded_solve): no counterpart in the
    reference, whose bounded solver is the primal one and whose dual solver knows x >= 0 only -- parity unpinned by
    nature; this is the checker of the GPU kernel's arithmetic, HiGHS the checker of its answers.  Infinite bounds stay
    infinite (a free side can never be violated nor carry a nonbasic variable)."""

    def __init__(self, c, A, b, lb, ub, basis, lb_nonbasic_vars, ub_nonbasic_vars, log_cap=1 << 16):
        c32, A32, b32 = preprocess(c, A, b)
        n = A32.shape[1]
        vs = np.full(n, VS_EXCLUDED, dtype=np.int32)
        vs[np.array(lb_nonbasic_vars).astype(int)] = VS_LOWER
        vs[np.array(ub_nonbasic_vars).astype(int)] = VS_UPPER
        OracleRevised.__init__(self, c, A, b, basis, kind="bounded", log_cap=log_cap, var_state=vs, check=False)
        self.lb = np.array(lb, dtype=np.float64)
        self.ub = np.array(ub, dtype=np.float64)
        self._bind()
        lib().lpo_bounded_update_bfs(C.byref(self.s))

    def solve(self, maxiters=100):
        status = lib().lpo_dual_bounded_solve(C.byref(self.s), int(maxiters))
        return self._result(status)


def bounds_on_bfs(A, b):
    """utils.py:20-40 in the reference's own scalar arithmetic."""
    m = A.shape[0]
    return factorial(m) * np.abs(A).max() ** (m - 1) * np.abs(b).max()


class OraclePrimalDual:
    """Restatement of PrimalDualAlgorithm (special_solvers.py:346-480).

    The restricted primal is the full problem [A | I] with inadmissible columns excluded (Bland order over
    the admissible columns followed by the artificials is the order of the reference's compacted matrix).
    The restricted dual y_rp = c_B B^-1 uses the inner solver's float64 product-form inverse; the reference takes a
    fresh LAPACK inverse at :433 -- of a FLOAT64 matrix (A_rp = np.hstack([A float32, np.eye(m)]) is float64, :411-413;
    recorded per case in tests/golden/primal_dual.json, "dual_step_inverse_dtypes"), so the two agree to rounding:
    the golden's last y @ A is matched to 1e-9, every outer iteration's admissible set and inner pivot exactly.
    """

    def __init__(self, c, A, b):
        self.c, self.A, self.b = preprocess(c, A, b)
        self.m, self.n = self.A.shape
        self.counter = 0
        self.optimum = False

    def solve(self, maxiters1=100, maxiters2=100):
        L = lib()
        c = self.c.astype(np.float64)
        A = self.A.astype(np.float64)
        b = self.b.astype(np.float64)
        m, n = self.m, self.n
        y = np.zeros(m)
        expanded = False
        if self.c.min() < 0:  # :388-400
            expanded = True
            M = bounds_on_bfs(self.A, self.b)
            c = np.hstack([c, np.zeros(1)])
            A = np.vstack([np.hstack([A, np.zeros((m, 1))]), np.ones((1, n + 1))])
            b = np.hstack([b, n * M * np.ones(1)])
            m, n = A.shape
            y = np.hstack([y, c.min() * np.ones(1)])
        A32 = np.ascontiguousarray(A.astype(np.float32))
        Aext = np.ascontiguousarray(np.hstack([A32, np.eye(m, dtype=np.float32)]))
        ldA = n + m
        inner_logs, adm_sets = [], []
        v = np.zeros(n)
        u = np.zeros(n)
        adm = np.zeros(n, dtype=np.int32)
        res = None
        while self.counter < maxiters1:
            self.counter += 1
            L.lpo_pd_admissible(_p(y, _dp), _p(Aext, _fp), _p(c, _dp), m, n, ldA, _p(v, _dp), _p(adm, _sp))
            adm_sets.append(np.flatnonzero(adm).tolist())
            vs = np.concatenate([np.where(adm != 0, VS_LOWER, VS_EXCLUDED), np.full(m, VS_LOWER)]).astype(np.int32)
            c_rp = np.concatenate([np.zeros(n), np.ones(m)])
            inner = OracleRevised(c_rp, Aext, b, np.arange(n, n + m), kind="primal", var_state=vs)
            res = inner.solve(maxiters2)
            inner_logs.append(res.log.tolist())
            if res.cost > 0.0:  # :427
                y_rp = c_rp[inner.basis] @ inner.Binv
                theta, umax = C.c_double(), C.c_double()
                L.lpo_pd_theta(_p(y_rp, _dp), _p(Aext, _fp), _p(c, _dp), _p(v, _dp), _p(adm, _sp),
                               m, n, ldA, _p(u, _dp), C.byref(theta), C.byref(umax))
                # :435: all(y_rp @ A_rp <= 0) over admissible columns and the artificial block
                if (u[adm != 0] <= 0).all() and (y_rp <= 0).all():
                    return OracleResult(status=DUAL_UNBOUNDED, iters=self.counter, adm_sets=adm_sets,
                                        inner_logs=inner_logs)
                y = y + theta.value * y_rp
            else:
                self.optimum = True
                break
        x = res.x[:n].copy()
        basis = inner.basis[inner.basis < n].copy()
        cost = float(np.dot(c, x))
        if expanded:
            basis = basis[basis != n - 1]
            x = x[:-1]
        return OracleResult(x=x, basis=basis, cost=cost, iters=self.counter,
                            status=OPTIMAL if self.optimum else RUNNING, adm_sets=adm_sets,
                            inner_logs=inner_logs, y=y)


class OracleTableau:
    """Tableau-form restatement (lpo_tableau_solve): carries T = B^-1 A, x_B and the reduced-cost row."""

    def __init__(self, c, A, b, basis, log_cap=1 << 16, ld=None, threads=1, check=True, pricing=0, kind="primal",
                 condensed=False, ratio="textbook", ratio_tol=1e-9):
        """kind = "dual": the dual simplex rules on the same tableau (lpo_tableau_dual_solve); the constructor then
        checks dual instead of primal feasibility of the starting basis (dual_solvers.py:14-16).
        condensed (primal only): carry the nonbasic columns only (lpo_tableau_solve_condensed), the layout the GPU
        tableau solver uses by default; ``slot_var[k]`` is the variable in column slot k, column ``nslots`` is x_B."""
        self.kind = kind
        self.condensed = bool(condensed)
        self.ratio, self.ratio_tol = ratio, float(ratio_tol)  # "harris": lpo_ratio_harris (opt-in, primal loops only)
        assert ratio in ("textbook", "harris") and not (ratio == "harris" and kind != "primal")
        assert not (condensed and kind != "primal")
        self.c, A32, self.b = preprocess(c, A, b)
        self.m, self.n = A32.shape
        m, n = self.m, self.n
        self.basis = np.array(basis).astype(np.int64)
        self.ld = (n + 1) if self.condensed else (ld or (n + 1))  # condensed: the full tableau is only a stepping stone
        B = A32[:, self.basis]
        slack_start = np.array_equal(B, np.eye(m, dtype=np.float32))
        if self.condensed and slack_start and not np.any(self.c[self.basis]):
            # B = I with zero basic costs: T = A, rc = c, x_B = b -- built directly in the condensed layout
            nb = np.setdiff1d(np.arange(n), self.basis)
            self.slot_var = nb.astype(np.int64)
            self.nslots = k = nb.size
            self.ld = ld or (k + 1)
            self.T = np.zeros((m + 1, self.ld))
            self.T[:m, :k] = A32[:, nb]
            self.T[:m, k] = self.b
            self.T[m, :k] = self.c[nb]
            self._finish(m, n, log_cap, threads, pricing)
            return
        self.T = np.zeros((m + 1, self.ld))
        if slack_start:
            if check and kind == "dual" and not np.all(self.c[self.basis] @ A32 <= self.c):
                raise ValueError("BasisIsDualInfeasibleError")
            self.T[:m, :n] = A32
            self.T[:m, n] = self.b
            self.T[m, :n] = self.c.astype(np.float64) - (self.c[self.basis].astype(np.float64) @ self.T[:m, :n]
                                                          if np.any(self.c[self.basis]) else 0.0)
            self.T[m, n] = -float(self.c[self.basis].astype(np.float64) @ self.T[:m, n])
        else:
            inv32 = np.linalg.inv(B)
            bfs32 = inv32 @ self.b
            if check and kind == "primal" and not np.all(bfs32 >= 0.0):
                raise ValueError("BasisIsPrimalInfeasibleError")
            if check and kind == "dual" and not np.all(self.c[self.basis] @ inv32 @ A32 <= self.c):
                raise ValueError("BasisIsDualInfeasibleError")
            self.T[:m, :n] = inv32 @ A32
            self.T[:m, n] = bfs32
            rc32 = self.c - self.c[self.basis] @ inv32 @ A32
            self.T[m, :n] = rc32
            self.T[m, n] = -float(self.c[self.basis] @ bfs32)
        self.T[m, self.basis] = 0.0
        del A32
        self.nslots = n
        if self.condensed:
            nb = np.setdiff1d(np.arange(n), self.basis)
            self.slot_var = nb.astype(np.int64)
            self.nslots = k = nb.size
            self.ld = ld or (k + 1)
            Tc = np.zeros((m + 1, self.ld))
            Tc[:, :k] = self.T[:, nb]
            Tc[:, k] = self.T[:, n]
            self.T = Tc
        self._finish(m, n, log_cap, threads, pricing)

    def _finish(self, m, n, log_cap, threads, pricing):
        self.var_state = np.full(n, VS_LOWER, dtype=np.int32)
        self.var_state[self.basis] = VS_BASIC
        self.log = np.zeros((log_cap, 3), dtype=np.int64)
        self._d, self._e, self._rowr = np.zeros(m + 1), np.zeros(m + 1), np.zeros(self.ld)
        s = self.s = _Tableau()
        s.m, s.n, s.ld = m, self.nslots, self.ld
        s.T, s.basis, s.var_state = _p(self.T, _dp), _p(self.basis, _ip), _p(self.var_state, _sp)
        s.f32_phase, s.pivots = 1, 0
        s.log, s.log_cap = _p(self.log, _ip), log_cap
        s.d, s.e, s.rowr, s.threads = _p(self._d, _dp), _p(self._e, _dp), _p(self._rowr, _dp), threads
        s.pricing = int(pricing)
        s.ratio_rule, s.ratio_tol = (1 if self.ratio == "harris" else 0), self.ratio_tol

    def solve(self, maxiters=100):
        if self.condensed:
            status = lib().lpo_tableau_solve_condensed(C.byref(self.s), _p(self.slot_var, _ip), int(maxiters))
        else:
            f = lib().lpo_tableau_solve if self.kind == "primal" else lib().lpo_tableau_dual_solve
            status = f(C.byref(self.s), int(maxiters))
        xB = np.ascontiguousarray(self.T[: self.m, self.nslots])
        x, cost = expand_and_cost(xB, self.basis, self.c, self.n)
        p = int(self.s.pivots)
        return OracleResult(x=x, basis=self.basis.copy(), cost=cost, iters=int(self.s.counter), status=status,
                            pivots=p, log=self.log[: min(p, self.log.shape[0])].copy())
