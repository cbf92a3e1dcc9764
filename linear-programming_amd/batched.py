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
            # the reference's own expression per LP (primal_solvers.py:38, a BLAS gemv): an einsum over the batch may
            # round a degenerate x_B (about 0) to the other sign
            bfs = np.stack([inv[i] @ b32[i] for i in range(self.B)])
            self._inv32, self._bfs32 = inv, bfs  # reused by BatchedTableauSolver's starting tableau
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
    tableau loop; from the slack basis with zero basic costs (``from_inequalities``, every synthetic config) the kernel
    reads A itself and takes the reference's pivot sequences, x and cost within 1e-9.  One wave per LP for m <= 64 and
    n - m <= 64; a grid of up to eight waves per LP (``k_batched_tab_grid``) up to m <= 128 with n - m <= 256 or m <= 256
    with n - m <= 128.
    From any other primal feasible basis the float32 starting tableau ``inv32 @ A`` is formed on the host as the
    reference's constructor state and CARRIED for the whole solve, whereas the reference recomputes ``inv64 @ A`` in
    float64 from its second pivot on: x and cost then agree with the reference and with the revised-form kernel to
    float32 accuracy only (about 1e-5 relative), and a reduced cost within ~1e-7 of zero may pick another pivot.  Use
    ``BatchedPrimalSolver`` where 1e-9 is needed from a general start (``solve_many`` does).  Same shape limits for
    both starts."""

    def __init__(self, c, A, b, basis, log_cap=0):
        super().__init__(c, A, b, basis, log_cap=log_cap)  # preprocessing, feasibility check (primal_solvers.py:41-53)
        k = self.n - self.m
        if not self.lib.lpx_batched_tableau_waves(self.m, k):
            raise ValueError("BatchedTableauSolver: m <= 128 with n - m <= 256, or m <= 256 with n - m <= 128")
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
                inv32 = self._inv32[i] if self.Binv0 is not None else np.eye(self.m, dtype=np.float32)
                nb = np.setdiff1d(np.arange(self.n), basis[i])
                if nb.size != k:
                    raise ValueError("a basis needs m distinct columns")
                var0[i] = nb
                T0[i] = (inv32 @ A32[i])[:, nb]
                rc0[i] = (c32[i] - c32[i][basis[i]] @ inv32 @ A32[i])[nb]
                xb0[i] = self._bfs32[i] if self.Binv0 is not None else b32[i]  # the value the feasibility check saw
            self.start = tuple(to_device(np.ascontiguousarray(v)) for v in (T0, rc0, xb0, var0))
            self.basis0 = self.basis.clone()

    @classmethod
    def from_inequalities(cls, c, G, h, log_cap=0):
        self = super().from_inequalities(c, G, h, log_cap=log_cap)
        if not self.lib.lpx_batched_tableau_waves(self.m, self.ineq_k):
            raise ValueError("BatchedTableauSolver: m <= 128 with k <= 256, or m <= 256 with k <= 128")
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
    feasibility is checked per LP as in ``BatchedDualSolver``.  Bit-identical to the oracle's dual tableau loop.  Shapes as
    for ``BatchedTableauSolver``: one wave per LP up to 64 x 64 tableau slots, a grid of up to eight waves beyond."""

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
        if not self.lib.lpx_batched_tableau_waves(self.m, k):
            raise ValueError("BatchedDualTableauSolver: m <= 128 with n - m <= 256, or m <= 256 with n - m <= 128")

    def enqueue(self, maxiters):
        _lib.check(self.lib.lpx_batched_tableau_dual_solve(
            self.B, self.m, self.n, self.A.data_ptr(), self.b.data_ptr(), self.c.data_ptr(), self.basis.data_ptr(),
            self.x.data_ptr(), self.cost.data_ptr(), self.status.data_ptr(), self.iters.data_ptr(), self.pivots.data_ptr(),
            int(maxiters), self.log.data_ptr() if self.log_cap else None, self.log_cap, stream_ptr()))


class BatchedBoundedSolver:
    """``BoundedVariablePrimalSimplexSolver(c, A, b, lb, ub, basis, lb_nonbasic_vars, ub_nonbasic_vars).solve(maxiters)``
    (special_solvers.py:94-343) for a batch of box-bounded LPs of one shape: one workgroup per LP runs the whole bounded
    loop -- float32-rounded reduced costs with the largest-coefficient rule, the three-way ratio test, bound flips, x_B
    recomputed from scratch every iteration -- with B^-1 in LDS (``lpx_revised_solve_small_batch``; m <= 128 at n = 2m).
    Batch-major inputs: c (B, n), A (B, m, n), b (B, m), lb / ub (B, n), basis (B, m), var_state (B, n) with
    ``_lib.VS_LOWER / VS_UPPER`` for the nonbasic variables (anything else: not enterable, as a variable in neither of the
    reference's two lists).  Infinite bounds are replaced per LP as the reference does (:139-142).  Bit-identical per LP to
    the oracle's ``lpo_bounded_solve``; a batch cannot raise per LP: ``raw["status"]`` tells unbounded LPs apart."""

    def __init__(self, c, A, b, lb, ub, basis, var_state, log_cap=0):
        from .utils import get_bounds_on_bfs

        torch = torch_cuda()
        self.torch = torch
        self.lib = _lib.load()
        c32 = np.asarray(c).astype(np.float32)
        A32 = np.array(A).astype(np.float32)
        b32 = np.array(b).astype(np.float32)
        neg = b32 < 0  # preprocessing.py:11-13, per LP
        A32[neg] *= -1
        b32[neg] *= -1
        self.B, self.m, self.n = A32.shape
        B, m, n = self.B, self.m, self.n
        lb = np.array(lb, dtype=np.float64)
        ub = np.array(ub, dtype=np.float64)
        for i in range(B):  # special_solvers.py:139-142
            if np.isinf(lb[i]).any() or np.isinf(ub[i]).any():
                M = get_bounds_on_bfs(A32[i], b32[i])
                lb[i][lb[i] == -np.inf] = -M
                ub[i][ub[i] == np.inf] = M
        basis = np.asarray(basis).astype(np.int64)
        vs = np.array(var_state, dtype=np.int32)
        vs[(vs != _lib.VS_LOWER) & (vs != _lib.VS_UPPER)] = _lib.VS_EXCLUDED
        np.put_along_axis(vs, basis, _lib.VS_BASIC, axis=1)
        Bmat = np.take_along_axis(A32, basis[:, None, :], axis=2)
        if np.array_equal(Bmat, np.broadcast_to(np.eye(m, dtype=np.float32), Bmat.shape)):
            inv = np.broadcast_to(np.eye(m), (B, m, m))
        else:
            inv = np.stack([np.linalg.inv(Bmat[i]) for i in range(B)]).astype(np.float64)  # float32 inverse, :144
        self.ldA, self.ldB = n, m
        self.A, self.b, self.c = to_device(np.ascontiguousarray(A32)), to_device(b32), to_device(c32)
        self.At = self.A.transpose(1, 2).contiguous()  # column j of A contiguous: coalesced row reads in the kernel
        self.lb, self.ub = to_device(lb), to_device(ub)
        self.Binv = to_device(np.ascontiguousarray(inv, dtype=np.float64))
        self.xB = torch.zeros((B, m), dtype=torch.float64, device="cuda")
        self.basis, self.var_state = to_device(basis), to_device(vs)
        self.ctrl = torch.zeros((B, 128), dtype=torch.uint8, device="cuda")
        self.log_cap = int(log_cap)
        self.log = torch.zeros((B, max(self.log_cap, 1), 3), dtype=torch.int64, device="cuda")
        self.x = torch.zeros((B, n), dtype=torch.float64, device="cuda")
        self.cost = torch.zeros(B, dtype=torch.float64, device="cuda")
        # one lpx_revised per LP, pointing into the batch-major tensors, as a device array
        import ctypes as C

        arr = (_lib.LpxRevised * B)()
        for i in range(B):
            s = arr[i]
            s.m, s.n, s.ldA, s.ldAt, s.ldB = m, n, n, m, m
            s.A, s.At = self.A[i].data_ptr(), self.At[i].data_ptr()
            s.b, s.c = self.b[i].data_ptr(), self.c[i].data_ptr()
            s.Binv, s.xB = self.Binv[i].data_ptr(), self.xB[i].data_ptr()
            s.basis, s.var_state = self.basis[i].data_ptr(), self.var_state[i].data_ptr()
            s.lb, s.ub, s.ctrl = self.lb[i].data_ptr(), self.ub[i].data_ptr(), self.ctrl[i].data_ptr()
            s.log = self.log[i].data_ptr() if self.log_cap else None
            s.log_cap = self.log_cap
        raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
        self.states = to_device(raw)
        # x_B of the starting state, B^-1 (b - A_lb lb - A_ub ub) (special_solvers.py:145), is formed by the kernel itself
        flags = np.zeros((B, 128), dtype=np.uint8)
        flags[:, 4] = 1  # lpx_ctrl.f32_phase = 1 (little endian int32 at offset 4); status RUNNING, counters 0
        self.ctrl.copy_(to_device(flags))

    def enqueue(self, maxiters):
        _lib.check(self.lib.lpx_revised_solve_small_batch(self.states.data_ptr(), self.B, self.m, self.n, _lib.KIND_BOUNDED,
                                                          int(maxiters), self.x.data_ptr(), self.cost.data_ptr(), stream_ptr()))

    def results(self):
        ctrl = self.ctrl.cpu().numpy()
        status = ctrl[:, 0:4].copy().view(np.int32)[:, 0]
        iters = ctrl[:, 8:16].copy().view(np.int64)[:, 0]
        pivots = ctrl[:, 16:24].copy().view(np.int64)[:, 0]
        return dict(x=self.x.cpu().numpy(), cost=self.cost.cpu().numpy(), basis=self.basis.cpu().numpy(), status=status,
                    iters=iters, pivots=pivots, var_state=self.var_state.cpu().numpy())

    def solve(self, maxiters=100):
        """One solve() call per LP (a second call continues from the state the first one left, as the reference's)."""
        # a new call restarts the per-call iteration counter and the status (primal_solvers.py:189-190)
        flags = self.ctrl.clone()
        flags[:, 0:4] = 0
        flags[:, 8:16] = 0
        self.ctrl.copy_(flags)
        self.enqueue(maxiters)
        self.raw = r = self.results()
        return [LinProgResult(x=r["x"][i], basis=r["basis"][i], cost=np.float64(r["cost"][i]), iters=int(r["iters"][i]),
                              optimum=bool(r["status"][i] == _lib.OPTIMAL)) for i in range(self.B)]

    def pivot_logs(self):
        p, lg = self.raw["pivots"], self.log.cpu().numpy()
        return [lg[i, : min(int(p[i]), self.log_cap)] for i in range(self.B)]


def solve_many(problems, maxiters=100, form="revised"):
    """Front-end for a LIST of LPs ``(c, A, b, basis)`` of mixed shapes: LPs of equal shape go into one launch, the
    launches of different shapes run on separate streams, results come back in input order.  ``form="tableau"``: groups
    that qualify (SLACK start with zero basic costs, up to 128 x 256 or 256 x 128 tableau slots) go through the faster condensed-tableau kernel,
    whose x and cost agree with the revised form's to 1e-9 (same pivots) rather than bit for bit; every other group --
    any non-slack starting basis included -- takes the revised-form kernels in both modes (bit-identical to the oracle's
    revised loop from the same float32 starting inverse; the tableau kernel would carry a float32 starting tableau for
    the whole solve, see ``BatchedTableauSolver``)."""
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
                    if s.start is not None:  # general start: that kernel carries a float32 tableau (1e-5, not 1e-9)
                        s = None
                except ValueError:  # too large for a wave: the revised-form kernels take it
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


class ResultGather:
    """The one collective of a sharded batch: after the solve every rank contributes its slice of
    the results and receives the whole batch.  A result row is ``n + m + 4`` 8-byte words per LP -- x (n float64), cost,
    basis (m int64), status, iters, pivots -- packed into ONE int64 buffer so that a single ``all_gather_into_tensor``
    moves everything.  Slices are padded to the largest shard (``all_gather_into_tensor`` wants equal pieces).
    Backend "nccl" (= RCCL over xGMI) gathers device buffers in stream order; "gloo" (CPU rehearsal, tests) stages
    through the host."""

    def __init__(self, group, total, m, n, device):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.total, self.m, self.n = int(total), int(m), int(n)
        self.words = self.n + self.m + 4
        self.per = max(shard(self.total, g, self.world)[1] - shard(self.total, g, self.world)[0] for g in range(self.world))
        self.send = torch.zeros((self.per, self.words), dtype=torch.int64, device=device)
        self.recv = torch.zeros((self.world * self.per, self.words), dtype=torch.int64, device=device)
        self.staged = dist.get_backend(group) == "gloo" and self.send.is_cuda

    def pack(self, x, cost, basis, status, iters, pivots):
        """Rows of this rank's slice (device or host tensors of the solver's dtypes) into the send buffer."""
        t, n, m = self.torch, self.n, self.m
        cnt = x.shape[0]
        s = self.send
        s[:cnt, :n] = x.view(t.int64)
        s[:cnt, n] = cost.view(t.int64)
        s[:cnt, n + 1:n + 1 + m] = basis
        s[:cnt, n + 1 + m] = status.to(t.int64)
        s[:cnt, n + 2 + m] = iters
        s[:cnt, n + 3 + m] = pivots

    def all_gather(self):
        if self.staged:
            host = self.torch.zeros(self.recv.shape, dtype=self.torch.int64)
            self.dist.all_gather_into_tensor(host, self.send.cpu(), group=self.group)
            self.recv.copy_(host)
        else:
            self.dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        return self.recv

    def unpack(self):
        """The whole batch in LP order as host arrays (x, cost, basis, status, iters, pivots)."""
        n, m = self.n, self.m
        rows = []
        for g in range(self.world):
            lo, hi = shard(self.total, g, self.world)
            rows.append(self.recv[g * self.per:g * self.per + (hi - lo)])
        r = self.torch.cat(rows).cpu().numpy()
        return dict(x=np.ascontiguousarray(r[:, :n]).view(np.float64), cost=np.ascontiguousarray(r[:, n]).view(np.float64),
                    basis=np.ascontiguousarray(r[:, n + 1:n + 1 + m]), status=r[:, n + 1 + m].astype(np.int32),
                    iters=np.ascontiguousarray(r[:, n + 2 + m]), pivots=np.ascontiguousarray(r[:, n + 3 + m]))


class ShardedBatch:
    """A FIXED batch of independent LPs split over the ranks of a process group (BASELINE config 3: 4096 LPs over 1/2/4/8
    GPUs): rank g builds ``solver_cls`` on the contiguous slice ``shard(B, g, world)`` of the batch-major inputs and solves
    it with no collective on the data path; ``gather()`` is the one all-gather of the results, after which every rank holds
    the whole batch -- bit for bit what one rank solving all B LPs returns, since LPs never interact.  Every rank passes the
    same full arrays (or, with ``local=True``, only its own slice)."""

    def __init__(self, solver_cls, c, A, b, basis, group, local=False, **kw):
        import torch.distributed as dist

        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        c, A, b, basis = (np.asarray(v) for v in (c, A, b, basis))
        if local:
            counts = [None] * self.world
            dist.all_gather_object(counts, int(A.shape[0]), group=group)
            self.total = int(sum(counts))
            lo, hi = shard(self.total, self.rank, self.world)
            if hi - lo != A.shape[0]:
                raise ValueError(f"rank {self.rank} holds {A.shape[0]} LPs, its shard of {self.total} is {hi - lo}")
            self.lo, self.hi = lo, hi
        else:
            self.total = int(A.shape[0])
            self.lo, self.hi = shard(self.total, self.rank, self.world)
            c, A, b, basis = c[self.lo:self.hi], A[self.lo:self.hi], b[self.lo:self.hi], basis[self.lo:self.hi]
        if self.hi == self.lo:
            raise ValueError("more ranks than LPs: this rank's shard is empty")
        self.solver = solver_cls(c, A, b, basis, **kw)
        self.results = ResultGather(group, self.total, self.solver.m, self.solver.n, "cuda")

    def enqueue(self, maxiters):
        self.solver.enqueue(maxiters)

    def gather(self):
        """Stream-ordered after the solve: pack this rank's results, all-gather, leave the whole batch in ``results``."""
        s = self.solver
        self.results.pack(s.x, s.cost, s.basis, s.status, s.iters, s.pivots)
        self.results.all_gather()

    def solve(self, maxiters=100):
        """Solve the shard, gather, and return the whole batch's raw arrays (host) -- on every rank."""
        self.enqueue(maxiters)
        self.gather()
        return self.results.unpack()
