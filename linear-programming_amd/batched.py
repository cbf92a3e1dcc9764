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
from .exceptions import BasisIsPrimalInfeasibleError


class BatchedPrimalSolver:
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
        if not np.all(bfs >= 0.0):
            raise BasisIsPrimalInfeasibleError
        self.A, self.b, self.c = to_device(A32), to_device(b32), to_device(c32)
        self.basis = to_device(basis)
        self.x = torch.zeros((self.B, self.n), dtype=torch.float64, device="cuda")
        self.cost = torch.zeros(self.B, dtype=torch.float64, device="cuda")
        self.status = torch.zeros(self.B, dtype=torch.int32, device="cuda")
        self.iters = torch.zeros(self.B, dtype=torch.int64, device="cuda")
        self.pivots = torch.zeros(self.B, dtype=torch.int64, device="cuda")
        self.log_cap = int(log_cap)
        self.log = torch.zeros((self.B, max(self.log_cap, 1), 3), dtype=torch.int64, device="cuda")

    def enqueue(self, maxiters):
        """Launch the batch (asynchronous).  NOTE: a solve always starts from the constructor's basis."""
        _lib.check(self.lib.lpx_batched_primal_solve(
            self.B, self.m, self.n, self.A.data_ptr(), self.b.data_ptr(), self.c.data_ptr(),
            None if self.Binv0 is None else self.Binv0.data_ptr(), self.basis.data_ptr(), self.x.data_ptr(),
            self.cost.data_ptr(), self.status.data_ptr(), self.iters.data_ptr(), self.pivots.data_ptr(),
            int(maxiters), self.log.data_ptr() if self.log_cap else None, self.log_cap, stream_ptr()))

    def results(self):
        """Raw batch-major outputs of the last launch on the host (one synchronising copy each)."""
        return dict(x=self.x.cpu().numpy(), cost=self.cost.cpu().numpy(), basis=self.basis.cpu().numpy(),
                    status=self.status.cpu().numpy(), iters=self.iters.cpu().numpy(), pivots=self.pivots.cpu().numpy())

    def solve(self, maxiters=100):
        self.enqueue(maxiters)
        self.raw = self.results()
        x, cost, basis, status, iters = (self.raw[k] for k in ("x", "cost", "basis", "status", "iters"))
        return [LinProgResult(x=x[i], basis=basis[i], cost=np.float64(cost[i]), iters=int(iters[i]),
                              optimum=bool(status[i] == _lib.OPTIMAL)) for i in range(self.B)]

    def pivot_logs(self):
        p = self.pivots.cpu().numpy()
        lg = self.log.cpu().numpy()
        return [lg[i, : min(int(p[i]), self.log_cap)] for i in range(self.B)]


def shard(count, rank, world):
    """Contiguous LP index range of `rank` (independent units: no data-path collective)."""
    return count * rank // world, count * (rank + 1) // world
