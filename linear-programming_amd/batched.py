"""Batches of independent small LPs: one workgroup per LP, state in LDS (BASELINE.json config 3).

``BatchedPrimalSolver(c, A, b, basis).solve(maxiters)`` takes batch-major arrays (B, n), (B, m, n), (B, m),
(B, m) and returns one ``LinProgResult`` per LP plus the raw arrays.  Semantics per LP are those of
``PrimalRevisedSimplexSolver`` (primal_solvers.py:176-263).  Batches shard over GPUs by LP index with no
collective during the solve (``shard(rank, world)``).
"""
import numpy as np

from . import _lib
from .data_classes import LinProgResult
from .device import stream_ptr, to_device, torch_cuda
from .exceptions import BasisIsDualInfeasibleError, BasisIsPrimalInfeasibleError


class BatchedPrimalSolver:
    ineq_k = 0  # > 0: built by from_inequalities (A holds only G)
    _DUAL = False

    @classmethod
    def from_inequalities(cls, c, G, h, log_cap=0):
        """``min c.x  s.t.  G x <= h, x >= 0`` with ``h >= 0``, batch-major (B, k), (B, m, k), (B, m): the standard form
        ``[G | I]`` of preprocessing.py:17-69 started from the slack basis, without ever building the identity block on
        the host or in HBM (``lpx_batched_primal_solve_ineq``).  Results carry all k+m variables, slacks last."""
        torch = torch_cuda()
        self = cls.__new__(cls)
        self.lib = _lib.load()
        G32 = np.ascontiguousarray(np.asarray(G).astype(np.float32))
        h32 = np.asarray(h).astype(np.float32)
        if np.any(h32 < 0):
            raise BasisIsPrimalInfeasibleError  # the slack basis is the start: x_B = h
        self.B, self.m, k = G32.shape
        self.n, self.ineq_k = k + self.m, k
        self.Binv0 = None
        self.A, self.b, self.c = to_device(G32), to_device(h32), to_device(np.asarray(c).astype(np.float32))
        self.basis = torch.zeros((self.B, self.m), dtype=torch.int64, device="cuda")
        self._alloc_outputs(torch, log_cap)
        return self

    def _alloc_outputs(self, torch, log_cap):
        self.x = torch.zeros((self.B, self.n), dtype=torch.float64, device="cuda")
        self.cost = torch.zeros(self.B, dtype=torch.float64, device="cuda")
        self.status = torch.zeros(self.B, dtype=torch.int32, device="cuda")
        self.iters = torch.zeros(self.B, dtype=torch.int64, device="cuda")
        self.pivots = torch.zeros(self.B, dtype=torch.int64, device="cuda")
        self.log_cap = int(log_cap)
        self.log = torch.zeros((self.B, max(self.log_cap, 1), 3), dtype=torch.int64, device="cuda")

    def __init__(self, c, A, b, basis, log_cap=0):
        torch = torch_cuda()
        self.lib = _lib.load()
        c32 = np.asarray(c).astype(np.float32)
        A32 = np.array(A).astype(np.float32)
        b32 = np.array(b).astype(np.float32)
        neg = b32 < 0  # preprocessing.py:11-13, per LP
        A32[neg] *= -1
        b32[neg] *= -1
        self.B, self.m, self.n = A32.shape
        basis = np.asarray(basis).astype(np.int64)
        # starting inverse per LP (primal_solvers.py:37): identity when the basis columns are the identity
        Bmat = np.take_along_axis(A32, basis[:, None, :], axis=2)
        eye = np.eye(self.m, dtype=np.float32)
        if np.array_equal(Bmat, np.broadcast_to(eye, Bmat.shape)):
            self.Binv0 = None
            bfs = b32
        else:
            inv = np.linalg.inv(Bmat)
            self.Binv0 = to_device(np.ascontiguousarray(inv.astype(np.float32)))
            bfs = np.einsum("bij,bj->bi", inv, b32)
        if self._DUAL:
            inv = np.broadcast_to(eye, Bmat.shape) if self.Binv0 is None else inv
            cB = np.take_along_axis(c32, basis, axis=1)
            if not np.all(np.einsum("bi,bik,bkj->bj", cB, inv, A32) <= c32):  # dual_solvers.py:14-16, per LP
                raise BasisIsDualInfeasibleError
        elif not np.all(bfs >= 0.0):
            raise BasisIsPrimalInfeasibleError
        self.A, self.b, self.c = to_device(A32), to_device(b32), to_device(c32)
        self.basis = to_device(basis)
        self._alloc_outputs(torch, log_cap)

    def enqueue(self, maxiters):
        """Launch the batch (asynchronous).  NOTE: a solve always starts from the constructor's basis."""
        if self.ineq_k:
            _lib.check(self.lib.lpx_batched_primal_solve_ineq(
                self.B, self.m, self.ineq_k, self.A.data_ptr(), self.b.data_ptr(), self.c.data_ptr(), self.basis.data_ptr(),
                self.x.data_ptr(), self.cost.data_ptr(), self.status.data_ptr(), self.iters.data_ptr(),
                self.pivots.data_ptr(), int(maxiters), self.log.data_ptr() if self.log_cap else None, self.log_cap,
                stream_ptr()))
            return
        f = self.lib.lpx_batched_dual_solve if self._DUAL else self.lib.lpx_batched_primal_solve
        _lib.check(f(
            self.B, self.m, self.n, self.A.data_ptr(), self.b.data_ptr(), self.c.data_ptr(),
            None if self.Binv0 is None else self.Binv0.data_ptr(), self.basis.data_ptr(), self.x.data_ptr(),
            self.cost.data_ptr(), self.status.data_ptr(), self.iters.data_ptr(), self.pivots.data_ptr(),
            int(maxiters), self.log.data_ptr() if self.log_cap else None, self.log_cap, stream_ptr()))

    def _optimum(self, status):
        return bool(status == _lib.OPTIMAL)

    def results(self):
        """Raw batch-major outputs of the last launch on the host (one synchronising copy each)."""
        return dict(x=self.x.cpu().numpy(), cost=self.cost.cpu().numpy(), basis=self.basis.cpu().numpy(),
                    status=self.status.cpu().numpy(), iters=self.iters.cpu().numpy(), pivots=self.pivots.cpu().numpy())

    def solve(self, maxiters=100):
        self.enqueue(maxiters)
        self.raw = self.results()
        x, cost, basis, status, iters = (self.raw[k] for k in ("x", "cost", "basis", "status", "iters"))
        return [LinProgResult(x=x[i], basis=basis[i], cost=np.float64(cost[i]), iters=int(iters[i]),
                              optimum=self._optimum(status[i])) for i in range(self.B)]

    def pivot_logs(self):
        p = self.pivots.cpu().numpy()
        lg = self.log.cpu().numpy()
        return [lg[i, : min(int(p[i]), self.log_cap)] for i in range(self.B)]


class BatchedDualSolver(BatchedPrimalSolver):
    """``DualRevisedSimplexSolver(c, A, b, basis).solve(maxiters)`` per LP (dual_solvers.py:9-120): dual-feasible starting
    bases (checked per LP, :14-16), ``optimum`` None until reached as in the reference; ``raw["status"]`` tells dual
    unbounded LPs (``_lib.DUAL_UNBOUNDED``) apart -- a batch cannot raise per LP.  m <= 64, n <= 256."""

    _DUAL = True

    def _optimum(self, status):
        return True if status == _lib.OPTIMAL else None


class BatchedTableauSolver(BatchedPrimalSolver):
    """The same per-LP semantics with the condensed tableau as carried state (``lpx_batched_tableau_solve``): one rank-1
    update per pivot instead of three chains of m dependent fmas, the tableau in registers.  Bit-identical to the oracle's
    tableau loop; the reference's pivot sequences, x and cost within 1e-9.  m <= 64, n - m <= 64.  From the slack basis with
    zero basic costs (``from_inequalities``, every synthetic config) the kernel reads A itself; from any other primal
    feasible basis the float32 starting tableau is formed on the host as the reference's constructor would."""

    def __init__(self, c, A, b, basis, log_cap=0):
        super().__init__(c, A, b, basis, log_cap=log_cap)  # preprocessing, feasibility check (primal_solvers.py:41-53)
        k = self.n - self.m
        if self.m > 64 or k > 64:
            raise ValueError("BatchedTableauSolver: m <= 64 and n - m <= 64")
        c32 = np.asarray(c).astype(np.float32)
        basis = np.asarray(basis).astype(np.int64)
        self.start = None
        slack = (self.Binv0 is None and np.array_equal(basis, np.broadcast_to(np.arange(k, self.n), (self.B, self.m)))
                 and not np.any(c32[:, k:] != 0))
        if not slack:
            # any other start: the reference's float32 constructor state (:37-38, :64) carried as a condensed tableau,
            # formed per LP on the host with the reference's own NumPy expressions
            A32, b32 = self.A.cpu().numpy(), self.b.cpu().numpy()
            T0 = np.empty((self.B, self.m, k), dtype=np.float32)
            rc0 = np.empty((self.B, k), dtype=np.float32)
            xb0 = np.empty((self.B, self.m), dtype=np.float32)
            var0 = np.empty((self.B, k), dtype=np.int64)
            for i in range(self.B):
                inv32 = np.linalg.inv(A32[i][:, basis[i]])
                nb = np.setdiff1d(np.arange(self.n), basis[i])
                if nb.size != k:
                    raise ValueError("a basis needs m distinct columns")
                var0[i] = nb
                T0[i] = (inv32 @ A32[i])[:, nb]
                rc0[i] = (c32[i] - c32[i][basis[i]] @ inv32 @ A32[i])[nb]
                xb0[i] = inv32 @ b32[i]
            self.start = tuple(to_device(np.ascontiguousarray(v)) for v in (T0, rc0, xb0, var0))
            self.basis0 = self.basis.clone()

    @classmethod
    def from_inequalities(cls, c, G, h, log_cap=0):
        self = super().from_inequalities(c, G, h, log_cap=log_cap)
        if self.m > 64 or self.ineq_k > 64:
            raise ValueError("BatchedTableauSolver: m <= 64 and k <= 64")
        return self

    def enqueue(self, maxiters):
        if getattr(self, "start", None) is not None:
            T0, rc0, xb0, var0 = self.start
            self.basis.copy_(self.basis0)  # input and output of the kernel: every solve starts from the constructor's basis
            _lib.check(self.lib.lpx_batched_tableau_solve_from(
                self.B, self.m, self.n, T0.data_ptr(), rc0.data_ptr(), xb0.data_ptr(), var0.data_ptr(), self.c.data_ptr(),
                self.basis.data_ptr(), self.x.data_ptr(), self.cost.data_ptr(), self.status.data_ptr(), self.iters.data_ptr(),
                self.pivots.data_ptr(), int(maxiters), self.log.data_ptr() if self.log_cap else None, self.log_cap, stream_ptr()))
            return
        f = self.lib.lpx_batched_tableau_solve_ineq if self.ineq_k else self.lib.lpx_batched_tableau_solve
        _lib.check(f(
            self.B, self.m, self.ineq_k if self.ineq_k else self.n, self.A.data_ptr(), self.b.data_ptr(), self.c.data_ptr(), self.basis.data_ptr(),
            self.x.data_ptr(), self.cost.data_ptr(), self.status.data_ptr(), self.iters.data_ptr(), self.pivots.data_ptr(),
            int(maxiters), self.log.data_ptr() if self.log_cap else None, self.log_cap, stream_ptr()))


class BatchedDualTableauSolver(BatchedDualSolver):
    """``DualRevisedSimplexSolver`` per LP with the condensed tableau as carried state (``lpx_batched_tableau_dual_solve``).
    Starting basis: the slack block, +-identity after preprocessing flipped the rows with b < 0, with zero costs; dual
    feasibility is checked per LP as in ``BatchedDualSolver``.  Bit-identical to the oracle's dual tableau loop."""

    def __init__(self, c, A, b, basis, log_cap=0):
        super().__init__(c, A, b, basis, log_cap=log_cap)
        k = self.n - self.m
        A32 = self.A.cpu().numpy()
        blk = A32[:, :, k:]
        diag = blk[:, np.arange(self.m), np.arange(self.m)]
        if (not np.array_equal(np.asarray(basis), np.broadcast_to(np.arange(k, self.n), (self.B, self.m)))
                or np.count_nonzero(blk) != self.B * self.m or not np.all(np.abs(diag) == 1.0)
                or np.any(self.c.cpu().numpy()[:, k:] != 0)):
            raise ValueError("BatchedDualTableauSolver starts from the slack basis (+-identity columns n-m..n-1, zero costs)")
        if self.m > 64 or k > 64:
            raise ValueError("BatchedDualTableauSolver: m <= 64 and n - m <= 64")

    def enqueue(self, maxiters):
        _lib.check(self.lib.lpx_batched_tableau_dual_solve(
            self.B, self.m, self.n, self.A.data_ptr(), self.b.data_ptr(), self.c.data_ptr(), self.basis.data_ptr(),
            self.x.data_ptr(), self.cost.data_ptr(), self.status.data_ptr(), self.iters.data_ptr(), self.pivots.data_ptr(),
            int(maxiters), self.log.data_ptr() if self.log_cap else None, self.log_cap, stream_ptr()))


def solve_many(problems, maxiters=100, form="revised"):
    """Front-end for a LIST of LPs ``(c, A, b, basis)`` of mixed shapes: LPs of equal shape go into one launch, the
    launches of different shapes run on separate streams, results come back in input order.  ``form="tableau"``: groups
    that qualify (slack start, m <= 64, n - m <= 64) go through the faster condensed-tableau kernel, whose x and cost
    agree with the revised form's to 1e-9 (same pivots) rather than bit for bit."""
    torch = torch_cuda()
    groups = {}
    for idx, (c, A, b, basis) in enumerate(problems):
        groups.setdefault(np.asarray(A).shape, []).append(idx)
    solvers = []
    main = torch.cuda.current_stream()
    for shape, idxs in groups.items():
        st = torch.cuda.Stream()
        st.wait_stream(main)
        with torch.cuda.stream(st):
            args = [np.stack([np.asarray(problems[i][k]) for i in idxs]) for k in range(4)]
            s = None
            if form == "tableau":
                try:
                    s = BatchedTableauSolver(*args)
                except ValueError:  # not a slack start or too large for a wave: the revised-form kernels take it
                    s = None
            if s is None:
                s = BatchedPrimalSolver(*args)
            s.enqueue(maxiters)
        solvers.append((idxs, s, st))
    out = [None] * len(problems)
    for idxs, s, st in solvers:
        st.synchronize()
        r = s.results()
        for pos, i in enumerate(idxs):
            out[i] = LinProgResult(x=r["x"][pos], basis=r["basis"][pos], cost=np.float64(r["cost"][pos]),
                                   iters=int(r["iters"][pos]), optimum=bool(r["status"][pos] == _lib.OPTIMAL))
    return out


def shard(count, rank, world):
    """Contiguous LP index range of `rank` (independent units: no data-path collective)."""
    return count * rank // world, count * (rank + 1) // world
