"""Tableau-form primal simplex on the device, unpartitioned or column-partitioned over ranks.

The tableau T = [B^-1 A | x_B ; rc | -z] stays in HBM and is updated in place by the streaming rank-1 kernel.
This is the build's large-scale form of ``PrimalRevisedSimplexSolver`` (same pivot rules,
primal_solvers.py:135-167; same result record); the reference has no tableau class of its own.

Column partition (BASELINE.json config 4): rank g holds columns [col0, col0 + n_local) of the global matrix
plus a replica of the x_B column.  Each pivot needs ONE collective: an all-gather of
(local min rc, first owned negative column, that column of T) -- ``lpx_tableau_msg_len(m)`` doubles per rank.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from .data_classes import LinProgResult
from .device import roundup, stream_ptr, to_device, torch_cuda
from .exceptions import (BasisIsDualInfeasibleError, BasisIsPrimalInfeasibleError, DualIsUnboundedError,
                         PrimalIsUnboundedError)
from .preprocessing import ProblemPreprocessingUtils
from .primal_solvers import signed_identity_diagonal


def row_alignment(cols):
    """Row stride granularity (doubles) of a streamed fp64 matrix.  K1's workgroups each own 512 consecutive columns of
    one row: with 4-KiB-aligned rows every such strip is one aligned 4 KiB block and the update streams at 6.3 TB/s
    instead of 5.7 (tools/oop_test.hip: 40002 -> 5.73, 40016 -> 6.01, 40448 -> 6.30 TB/s on the 20001 x 40001 tableau).
    Small matrices keep the padding small: 128-byte lines, then pairs."""
    return 512 if cols >= 4096 else (16 if cols >= 128 else 2)


def column_range(n, rank, world):
    """Global columns [j0, j1) owned by `rank` in an N-way column partition."""
    return n * rank // world, n * (rank + 1) // world


class ProposalExchange:
    """The one collective of a column-partitioned pivot: all-gather of the ranks' proposals.

    A proposal is ``msg_len`` float64: [local min rc, global index of the first owned negative column (2^62 if
    none), that column of T (m values), its reduced cost].  Backend "nccl" (= RCCL over xGMI) gathers device
    buffers in stream order; backend "gloo" (CPU rehearsal, tests) stages through the host.
    """

    def __init__(self, group, msg_len, device):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.nranks = dist.get_world_size(group)
        self.msg = int(msg_len)
        # two slots: with lookahead the proposals of pivot t+1 are gathered while pivot t's are still in use
        self.sendbufs = [torch.zeros(self.msg, dtype=torch.float64, device=device) for _ in range(2)]
        self.recvbufs = [torch.zeros(self.msg * self.nranks, dtype=torch.float64, device=device) for _ in range(2)]
        self.sendbuf, self.recvbuf = self.sendbufs[0], self.recvbufs[0]
        self.staged = dist.get_backend(group) == "gloo" and self.sendbuf.is_cuda
        self.side = None
        if self.staged:
            self.pinned = [torch.zeros(self.msg * self.nranks, dtype=torch.float64).pin_memory() for _ in range(2)]

    def all_gather(self, slot=0):
        self.wait(self.all_gather_async(slot))
        return self.recvbufs[slot]

    def all_gather_async(self, slot):
        """Start the all-gather of slot `slot`; returns a handle for wait().  With the nccl backend the collective is
        ordered after the work already enqueued on the current stream and runs beside what is enqueued next; wait() makes
        the current stream (not the host) wait for it.  The staged gloo path (CPU rehearsal with device buffers) keeps the
        same discipline on a side stream: the copy of the gathered proposals into the receive slot is ordered after the
        kernels that wrote the send slot, runs beside the update kernel that the caller launches next, and the current
        stream waits for its event -- so the two-slot ordering is exercised without a multi-GPU node."""
        send, recv = self.sendbufs[slot], self.recvbufs[slot]
        torch = self.torch
        if self.staged:
            if self.side is None:
                self.side = torch.cuda.Stream()
            cur = torch.cuda.current_stream()
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                host = torch.zeros(self.msg * self.nranks, dtype=torch.float64)
                self.dist.all_gather_into_tensor(host, send.cpu(), group=self.group)
                self.pinned[slot].copy_(host)
                recv.copy_(self.pinned[slot], non_blocking=True)
                done = torch.cuda.Event()
                done.record(self.side)
            return done
        return self.dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)

    def wait(self, work):
        if work is None:
            return
        if self.staged:
            self.torch.cuda.current_stream().wait_event(work)
        else:
            work.wait()


class TableauEngine:
    """Device state of one (slice of a) tableau."""

    CHUNK = 64
    FIRST_CHUNK = 16
    #: column-partitioned loops (primal and dual): overlap the exchange of pivot t+1 with the update of pivot t
    LOOKAHEAD = os.environ.get("LPX_TABLEAU_LOOKAHEAD", "1") != "0"
    #: largest tableau (bytes) that gets the second buffer of the single-launch chain
    FUSED_MAX_BYTES = int(os.environ.get("LPX_TABLEAU_FUSED_MAX_MB", "256")) << 20
    #: smallest tableau (bytes) that gets a second buffer for the pipelined deferred-update chain (lpx.h,
    #: lpx_tableau_pipelined: the pass over the tableau and the next block's decisions run side by side); negative = never
    PIPELINE_MIN_BYTES = (None if int(os.environ.get("LPX_TABLEAU_PIPELINE_MIN_MB", "1024")) < 0
                          else int(os.environ.get("LPX_TABLEAU_PIPELINE_MIN_MB", "1024")) << 20)

    def __init__(self, T, m, n_local, basis, var_state, c_global, col0=0, n_global=None, log_cap=0, f32_phase=1,
                 group=None, pricing="first_negative", dual=False, slot_var=None, var_slot=None, ratio="textbook",
                 ratio_tol=1e-9):
        torch = torch_cuda()
        self.dual = bool(dual)
        self.slot_var, self.var_slot = slot_var, var_slot  # condensed form: nonbasic columns only (lpx.h)
        assert (slot_var is None) == (var_slot is None)
        self.torch = torch
        self.lib = _lib.load()
        self.m, self.n = int(m), int(n_local)
        self.col0 = int(col0)
        self.n_global = self.n if n_global is None else int(n_global)
        assert T.dtype == torch.float64 and T.is_cuda and T.is_contiguous() and T.shape[0] == m + 1
        self.T = T
        self.ld = T.shape[1]
        self.basis = basis
        self.var_state = var_state
        self.c_global = c_global
        self.ctrl = torch.zeros(128, dtype=torch.uint8, device="cuda")
        self.log_cap = int(log_cap)
        self.log = torch.zeros((max(self.log_cap, 1), 3), dtype=torch.int64, device="cuda")
        ws = self.lib.lpx_tableau_workspace_bytes(self.m, self.n)
        self.workspace = torch.zeros(ws, dtype=torch.uint8, device="cuda")
        self.x = torch.zeros(self.n_global, dtype=torch.float64, device="cuda")
        self.cost = torch.zeros(1, dtype=torch.float64, device="cuda")
        self.host_ctrl = _lib.LpxCtrl()
        s = self.s = _lib.LpxTableau()
        s.m, s.n, s.ld, s.col0, s.n_global = self.m, self.n, self.ld, self.col0, self.n_global
        s.pricing = {"first_negative": _lib.PRICE_FIRST_NEGATIVE, "dantzig": _lib.PRICE_DANTZIG}[pricing]
        s.ratio_rule = {"textbook": _lib.RATIO_TEXTBOOK, "harris": _lib.RATIO_HARRIS}[ratio]
        s.ratio_tol = float(ratio_tol)
        if s.ratio_rule != _lib.RATIO_TEXTBOOK and (group is not None or dual):
            raise ValueError("ratio='harris' is an option of the unpartitioned primal tableau loop only")
        s.T, s.basis, s.var_state, s.ctrl = T.data_ptr(), basis.data_ptr(), var_state.data_ptr(), self.ctrl.data_ptr()
        s.log = self.log.data_ptr() if self.log_cap > 0 else None
        s.log_cap = self.log_cap
        s.workspace, s.workspace_bytes = self.workspace.data_ptr(), ws
        s.slot_var = None if slot_var is None else slot_var.data_ptr()
        s.var_slot = None if var_slot is None else var_slot.data_ptr()
        # launch-bound sizes (unpartitioned loops, primal and dual): a second buffer lets liblpx run one launch per pivot, the
        # tableau ping-ponging between the two (lpx.h, lpx_tableau.T_alt); beyond FUSED_MAX_BYTES the update is
        # HBM-bound, the three-kernel in-place chain costs nothing and the memory is better left alone
        self.T_alt = None
        if group is None and T.numel() * 8 <= self.FUSED_MAX_BYTES:
            self.T_alt = torch.zeros_like(T)
        elif (not self.dual and self.PIPELINE_MIN_BYTES is not None
              and 8 * (self.m + 1) * (self.n + 1) >= self.PIPELINE_MIN_BYTES
              and (group is None or (slot_var is not None and self._c_loop_wanted(group)))):
            # (a column-partitioned slice is pipelined by the library's own loop only: lpx_tableau_dist_iterate)
            self.T_alt = torch.zeros_like(T)
        s.T_alt = None if self.T_alt is None else self.T_alt.data_ptr()
        _lib.check(self.lib.lpx_ctrl_init(s.ctrl, int(f32_phase), stream_ptr()))
        # column partition
        self.group = group
        self.nranks = 1
        #: pivots per pass over the tableau (0: one update per pivot).  Column-partitioned: ONE value for all ranks -- the
        #: slices are ragged, so lpx_tableau_block_size() of a rank's own slice can differ across ranks at a size
        #: threshold, and ranks that disagree issue different numbers of collectives (a hang): the largest wins
        self.kb = int(self.lib.lpx_tableau_block_size(C.byref(s)))
        if group is not None:
            self.exchange = ProposalExchange(group, self.lib.lpx_tableau_msg_len(self.m), "cuda")
            self.nranks = self.exchange.nranks
            import torch.distributed as dist

            t = torch.tensor([self.kb], dtype=torch.int64, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            self.kb = int(t.item())
            # the library's own loop around RCCL (lpx.h, lpx_tableau_dist_iterate): a communicator of liblpx's own over the
            # same ranks -- rank 0's unique id travels through the process group once
            self.dist_comm = None
            if self._c_loop_wanted(group):
                rank = dist.get_rank(group)
                ident = (C.c_char * 128)()
                if rank == 0:
                    _lib.check(self.lib.lpx_dist_unique_id(ident))
                box = [bytes(ident.raw)]
                dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0), group=group)
                comm = C.c_void_p()
                _lib.check(self.lib.lpx_dist_comm_create(box[0], self.nranks, rank, C.byref(comm)))
                self.dist_comm = comm

    #: column-partitioned loops over RCCL run inside liblpx (one C call per chunk of pivots); "0": the Python loop around
    #: lpx_tableau_block_* and torch.distributed's all-gather (what the gloo rehearsals run)
    C_LOOP = os.environ.get("LPX_DIST_C_LOOP", "1") != "0"

    @classmethod
    def _c_loop_wanted(cls, group):
        import torch.distributed as dist

        return cls.C_LOOP and group is not None and dist.get_backend(group) == "nccl"

    def __del__(self):
        comm = getattr(self, "dist_comm", None)
        if comm is not None:
            try:
                self.lib.lpx_dist_comm_destroy(comm)
            except Exception:
                pass
            self.dist_comm = None

    # ------------------------------------------------------------------ construction from a host problem
    @classmethod
    def from_columns(cls, A_cols32, b32, c32_global, basis, col0, n_global, inv32=None, bfs32=None, ld_align=None,
                     var_ids=None, condensed=False, start32=None, **kw):
        """Build the local slice from columns of the float32 A (slack basis unless inv32): columns [col0, col0+n_local),
        or the variables ``var_ids``.  ``condensed``: carry only the nonbasic ones of them (the basic columns of a
        tableau are unit vectors that no pivot changes).  ``start32`` = (inv32 @ A, c - c_B @ inv32 @ A) for ALL columns,
        formed by the caller on the host with the reference's float32 NumPy expressions (the device's float32 matmul
        rounds differently)."""
        torch = torch_cuda()
        basis_np = np.asarray(basis, dtype=np.int64)
        if var_ids is None:
            var_ids = np.arange(col0, col0 + A_cols32.shape[1], dtype=np.int64)
        var_ids = np.asarray(var_ids, dtype=np.int64)
        if condensed:
            keep = ~np.isin(var_ids, basis_np)
            if not keep.all():
                A_cols32, var_ids = A_cols32[:, keep], var_ids[keep]
            if var_ids.size == 0:
                raise ValueError("this rank would hold no nonbasic column")
        else:
            assert np.array_equal(var_ids, np.arange(col0, col0 + var_ids.size)), "the full form needs a column range"
        ids_dev = to_device(var_ids)
        m, n_local = A_cols32.shape
        if ld_align is None:
            ld_align = row_alignment(n_local + 1)
        ld = roundup(n_local + 1, ld_align)
        T = torch.zeros((m + 1, ld), dtype=torch.float64, device="cuda")
        A_dev = torch.from_numpy(np.ascontiguousarray(A_cols32)).to("cuda")
        c_dev = to_device(np.asarray(c32_global, dtype=np.float32))
        cb = np.asarray(c32_global, dtype=np.float32)[basis]
        if inv32 is None:  # B = I: T = A, x_B = b, rc = c - c_B A
            T[:m, :n_local].copy_(A_dev)
            T[:m, n_local].copy_(to_device(np.asarray(b32, dtype=np.float64)))
            rc = c_dev[ids_dev].to(torch.float64)
            if np.any(cb):
                rc = rc - to_device(cb.astype(np.float64)) @ T[:m, :n_local]
            T[m, :n_local].copy_(rc)
            T[m, n_local] = -float(cb.astype(np.float64) @ np.asarray(b32, dtype=np.float64))
        else:  # the reference's float32 starting state (primal_solvers.py:37-38, :64), carried as a tableau
            if start32 is not None:
                BA32, rc32 = start32
                T[:m, :n_local].copy_(to_device(np.ascontiguousarray(BA32[:, var_ids])))
                T[m, :n_local].copy_(to_device(np.ascontiguousarray(rc32[var_ids])))
            else:
                inv_dev = to_device(np.asarray(inv32, dtype=np.float32))
                T[:m, :n_local].copy_(inv_dev @ A_dev)
                y32 = to_device(cb.astype(np.float32)) @ inv_dev
                T[m, :n_local].copy_(c_dev[ids_dev] - y32 @ A_dev)
            T[:m, n_local].copy_(to_device(np.asarray(bfs32, dtype=np.float64)))
            T[m, n_local] = -float(cb @ np.asarray(bfs32, dtype=np.float32))
        del A_dev
        if condensed:
            var_slot = torch.full((int(n_global),), -1, dtype=torch.int64, device="cuda")
            var_slot[ids_dev] = torch.arange(n_local, dtype=torch.int64, device="cuda")
            vs_dev = to_device(np.full(n_local, _lib.VS_LOWER, dtype=np.int32))
            return cls(T, m, n_local, to_device(basis_np), vs_dev, c_dev, col0=0, n_global=n_global, slot_var=ids_dev,
                       var_slot=var_slot, **kw)
        var_state = np.full(n_local, _lib.VS_LOWER, dtype=np.int32)
        loc = basis_np - col0
        loc = loc[(loc >= 0) & (loc < n_local)]
        var_state[loc] = _lib.VS_BASIC
        vs_dev = to_device(var_state)
        T[m, :n_local][vs_dev == _lib.VS_BASIC] = 0.0  # primal_solvers.py:65
        return cls(T, m, n_local, to_device(basis_np), vs_dev, c_dev, col0=col0, n_global=n_global, **kw)

    # ------------------------------------------------------------------ loop
    def read_ctrl(self):
        _lib.check(self.lib.lpx_read_ctrl(self.s.ctrl, C.byref(self.host_ctrl), stream_ptr()))
        return self.host_ctrl

    def enqueue(self, iters):
        """Enqueue `iters` pivots without any host synchronisation."""
        st = stream_ptr()
        if self.group is None:
            f = self.lib.lpx_tableau_dual_iterate if self.dual else self.lib.lpx_tableau_iterate
            _lib.check(f(C.byref(self.s), int(iters), st))
            return
        ex = self.exchange
        iters = int(iters)
        if self.kb > 0 and iters > 0 and getattr(self, "dist_comm", None) is not None:
            # large slices over RCCL: the whole chunk of pivots, collectives included, in ONE call (lpx_tableau_dist_iterate)
            f = self.lib.lpx_tableau_dual_dist_iterate if self.dual else self.lib.lpx_tableau_dist_iterate
            _lib.check(f(C.byref(self.s), self.dist_comm, self.nranks, iters, self.kb, ex.sendbuf.data_ptr(),
                         ex.recvbuf.data_ptr(), st))
            return
        if self.dual:
            kb = self.kb
            if kb > 0 and iters > 0:  # large slices: deferred updates (lpx.h, lpx_tableau_dual_block_*)
                _lib.check(self.lib.lpx_tableau_dual_block_begin(C.byref(self.s), st))
                for it in range(iters):
                    u = it % kb
                    _lib.check(self.lib.lpx_tableau_dual_block_propose(C.byref(self.s), u, ex.sendbuf.data_ptr(), st))
                    ex.all_gather()
                    _lib.check(self.lib.lpx_tableau_dual_block_commit(C.byref(self.s), u, ex.recvbuf.data_ptr(), self.nranks, st))
                    if u == kb - 1 or it + 1 == iters:
                        _lib.check(self.lib.lpx_tableau_block_flush(C.byref(self.s), kb, st))
                return
            if not self.LOOKAHEAD:
                for it in range(iters):
                    _lib.check(self.lib.lpx_tableau_dual_propose(C.byref(self.s), ex.sendbuf.data_ptr(), st))
                    ex.all_gather()
                    _lib.check(self.lib.lpx_tableau_dual_commit(C.byref(self.s), ex.recvbuf.data_ptr(), self.nranks, st))
                return
            # lookahead, as for the primal loop below: the leaving row and the proposal of pivot t+1 are formed from the
            # old tableau, e and the old pivot row before pivot t's update is launched (lpx_tableau_dual_commit_begin)
            if iters <= 0:
                return
            cur = 0
            _lib.check(self.lib.lpx_tableau_dual_propose(C.byref(self.s), ex.sendbufs[cur].data_ptr(), st))
            ex.all_gather(cur)
            for it in range(iters):
                nxt = cur ^ 1
                last = it + 1 == iters
                _lib.check(self.lib.lpx_tableau_dual_commit_begin(C.byref(self.s), ex.recvbufs[cur].data_ptr(), self.nranks,
                                                                  int(it > 0), None if last else ex.sendbufs[nxt].data_ptr(), st))
                work = None if last else ex.all_gather_async(nxt)
                _lib.check(self.lib.lpx_tableau_commit_update(C.byref(self.s), st))
                ex.wait(work)
                cur = nxt
            return
        kb = self.kb
        if kb > 0 and iters > 0:
            # large slices: deferred updates (lpx.h, lpx_tableau_block_*): propose -> all-gather -> commit per pivot, one pass
            # over the slice per kb pivots; T is current again when the call returns
            _lib.check(self.lib.lpx_tableau_block_begin(C.byref(self.s), st))
            for it in range(iters):
                u = it % kb
                _lib.check(self.lib.lpx_tableau_block_propose(C.byref(self.s), u, ex.sendbuf.data_ptr(), st))
                ex.all_gather()
                _lib.check(self.lib.lpx_tableau_block_commit(C.byref(self.s), u, ex.recvbuf.data_ptr(), self.nranks, st))
                if u == kb - 1 or it + 1 == iters:
                    _lib.check(self.lib.lpx_tableau_block_flush(C.byref(self.s), kb, st))
            return
        if not self.LOOKAHEAD:
            for it in range(iters):
                # the update kernel of the previous pivot leaves the pricing partials behind; only the first pivot of a
                # call rescans the reduced-cost row (the caller may have touched T in between)
                _lib.check(self.lib.lpx_tableau_propose(C.byref(self.s), ex.sendbuf.data_ptr(), int(it == 0), st))
                ex.all_gather()
                _lib.check(self.lib.lpx_tableau_commit(C.byref(self.s), ex.recvbuf.data_ptr(), self.nranks, st))
            return
        # lookahead: the proposal of pivot t+1 is formed before pivot t's update is launched, and its all-gather
        # travels while the update runs (lpx_tableau_commit_begin / _update)
        if iters <= 0:
            return
        cur = 0
        _lib.check(self.lib.lpx_tableau_propose(C.byref(self.s), ex.sendbufs[cur].data_ptr(), 1, st))
        ex.all_gather(cur)
        for it in range(iters):
            nxt = cur ^ 1
            last = it + 1 == iters
            _lib.check(self.lib.lpx_tableau_commit_begin(C.byref(self.s), ex.recvbufs[cur].data_ptr(), self.nranks,
                                                         None if last else ex.sendbufs[nxt].data_ptr(), st))
            work = None if last else ex.all_gather_async(nxt)
            _lib.check(self.lib.lpx_tableau_commit_update(C.byref(self.s), st))
            ex.wait(work)
            cur = nxt

    def run(self, maxiters):
        st = stream_ptr()
        _lib.check(self.lib.lpx_ctrl_begin_solve(self.s.ctrl, st))
        left = int(maxiters)
        ctrl = None
        first = True
        # deferred updates (large tableaus): whole blocks between two looks at the control block, and more of them --
        # every call starts by re-reading the reduced-cost row and x_B from T and ends with a pass over T
        kb = self.kb if self.group is not None else int(self.lib.lpx_tableau_block_size(C.byref(self.s)))
        chunk, first_chunk = (8 * kb, kb) if kb > 0 else (self.CHUNK, self.FIRST_CHUNK)
        if kb > 0 and not self.dual and self.lib.lpx_tableau_pipelined(C.byref(self.s)) and (
                self.group is None or getattr(self, "dist_comm", None) is not None):
            # pipelined chain: the first and the last block of a call overlap with nothing, so calls are long
            chunk, first_chunk = 32 * kb, 2 * kb
        while left > 0:
            # short first chunk (eager launches): small solves end here; later chunks are long enough for
            # liblpx to replay them as hipGraphs
            k = min(left, first_chunk if first else chunk)
            first = False
            self.enqueue(k)
            left -= k
            ctrl = self.read_ctrl()
            if ctrl.status != _lib.RUNNING:
                break
        return ctrl if ctrl is not None else self.read_ctrl()

    def expand(self):
        _lib.check(self.lib.lpx_tableau_expand(C.byref(self.s), self.c_global.data_ptr(), self.x.data_ptr(),
                                               self.cost.data_ptr(), stream_ptr()))
        return self.x.cpu().numpy(), np.float64(self.cost.cpu().numpy()[0])

    def host_basis(self):
        return self.basis.cpu().numpy()

    def host_log(self, count):
        return self.log[: min(int(count), self.log_cap)].cpu().numpy()


class PrimalTableauSimplexSolver:
    """``PrimalRevisedSimplexSolver`` semantics on a carried tableau (constructor + solve() + LinProgResult).

    ``group``: a torch.distributed process group => the tableau is column-partitioned over its ranks
    (every rank passes the full problem or, with ``columns=(j0, j1)``, only its own column slice of A).
    """

    _LOG_CAP = 0

    #: carry only the nonbasic columns (same numbers, half the memory and HBM traffic at n = 2m); False = every column
    CONDENSED = os.environ.get("LPX_TABLEAU_CONDENSED", "1") != "0"

    def __init__(self, c, A, b, basis, group=None, columns=None, n_global=None, pricing="first_negative", condensed=None,
                 ratio="textbook", ratio_tol=1e-9):
        """``pricing="dantzig"`` (most negative reduced cost) is an opt-in extension: the reference's standard-form
        solvers only have the first-negative-index rule, which stays the default.  ``ratio="harris"``: a Harris-style
        two-pass ratio test with tolerance ``ratio_tol`` (lpx.h, lpx_tableau.ratio_rule) instead of the reference's
        textbook test (primal_solvers.py:149-167) -- opt-in as well, parity unpinned against the reference (it has no
        such rule), bit-exact against the oracle's restatement of the same rule."""
        condensed = self.CONDENSED if condensed is None else bool(condensed)
        var_ids = None
        if columns is None:
            self.c, A32, self.b = ProblemPreprocessingUtils.preprocess_problem(c, A, b)
            self.m, self.n = A32.shape
            j0, j1 = 0, self.n
            world = 1
            if group is not None:
                import torch.distributed as dist

                rank, world = dist.get_rank(group), dist.get_world_size(group)
            # fewer nonbasic columns than ranks (n = m: none at all): every rank falls back to the full layout
            condensed = condensed and self.n - len(set(np.asarray(basis, dtype=np.int64).tolist())) >= world
            if group is not None:
                if condensed:  # partition the nonbasic variables, not the column range: a slack basis is a whole range
                    nonbasic = np.setdiff1d(np.arange(self.n), np.asarray(basis, dtype=np.int64))
                    lo, hi = column_range(nonbasic.size, rank, world)
                    var_ids = nonbasic[lo:hi]
                else:
                    j0, j1 = column_range(self.n, rank, world)
            A_cols = A32[:, j0:j1] if var_ids is None else A32[:, var_ids]
        else:  # caller supplies only its slice (already float32, b >= 0): large problems
            j0, j1 = columns
            A_cols = np.asarray(A, dtype=np.float32)
            self.c = np.asarray(c, dtype=np.float32)
            self.b = np.asarray(b, dtype=np.float32)
            if np.any(self.b < 0):
                raise ValueError("column-slice construction needs b >= 0 (row flips need the whole matrix)")
            self.m, self.n = A_cols.shape[0], int(n_global)
            A32 = None
        self.basis = np.array(basis).astype(int)
        inv32 = bfs32 = start32 = None
        k = self.n - self.m
        slack = np.array_equal(self.basis, np.arange(k, self.n)) and (
            self._slack_block_is_identity(A_cols, j0, j1, k) if A32 is None else self._slack_block_is_identity(A32, 0, self.n, k))
        if not slack:
            if A32 is None:
                raise ValueError("a non-slack starting basis needs the whole matrix")
            inv32 = np.linalg.inv(A32[:, self.basis])  # primal_solvers.py:37
            bfs32 = inv32 @ self.b
            start32 = (inv32 @ A32, self.c - self.c[self.basis] @ inv32 @ A32)  # :64, float32 like the reference's
        if not np.all((self.b if bfs32 is None else bfs32) >= 0.0):  # primal_solvers.py:43-48
            raise BasisIsPrimalInfeasibleError
        self.counter = None
        self.optimum = None
        self._engine = TableauEngine.from_columns(A_cols, self.b, self.c, self.basis, j0, self.n, inv32=inv32,
                                                  bfs32=bfs32, log_cap=self._LOG_CAP, group=group, pricing=pricing,
                                                  var_ids=var_ids, condensed=condensed, start32=start32, ratio=ratio,
                                                  ratio_tol=ratio_tol)

    @staticmethod
    def _slack_block_is_identity(A_cols, j0, j1, k):
        lo, hi = max(j0, k), j1
        if lo >= hi:
            return True
        blk = A_cols[:, lo - j0:hi - j0]
        rows = np.arange(lo - k, hi - k)
        return np.count_nonzero(blk) == hi - lo and np.all(blk[rows, np.arange(hi - lo)] == 1.0)

    def solve(self, maxiters: int = 100):
        self.counter = 0
        self.optimum = False
        ctrl = self._engine.run(maxiters)
        self.counter = int(ctrl.counter)
        self.basis[:] = self._engine.host_basis()
        if ctrl.status == _lib.PRIMAL_UNBOUNDED:
            raise PrimalIsUnboundedError
        self.optimum = ctrl.status == _lib.OPTIMAL
        x, cost = self._engine.expand()
        return LinProgResult(x=x, basis=self.basis, cost=cost, iters=self.counter, optimum=self.optimum)

    def pivot_log(self):
        return self._engine.host_log(self._engine.read_ctrl().pivots)


class DualTableauSimplexSolver(PrimalTableauSimplexSolver):
    """``DualRevisedSimplexSolver`` semantics (dual_solvers.py:9-120) on the carried tableau: the constructor checks
    dual feasibility of the starting basis (:14-16), ``solve`` runs the dual simplex (:82-108) and raises
    ``DualIsUnboundedError`` (:41-44).  ``optimum`` stays ``None`` until optimality is reached, as in the reference.
    ``group`` column-partitions the tableau exactly as for the primal solver."""

    def __init__(self, c, A, b, basis, group=None, condensed=None):
        condensed = self.CONDENSED if condensed is None else bool(condensed)
        self.c, A32, self.b = ProblemPreprocessingUtils.preprocess_problem(c, A, b)
        self.m, self.n = A32.shape
        self.basis = np.array(basis).astype(int)
        d = signed_identity_diagonal(A32[:, self.basis])
        if d is not None:  # +-identity basis: its inverse is itself and inv32 @ A is a row scaling, both exact
            inv32, bfs32 = np.diag(d), d * self.b
            BA32, z32 = A32 * d[:, None], (self.c[self.basis] * d) @ A32
        else:
            inv32 = np.linalg.inv(A32[:, self.basis])  # primal_solvers.py:37
            bfs32 = inv32 @ self.b
            BA32, z32 = inv32 @ A32, self.c[self.basis] @ inv32 @ A32
        start32 = (BA32, self.c - z32)
        if not np.all(z32 <= self.c):  # dual_solvers.py:14-16
            raise BasisIsDualInfeasibleError
        j0, j1, var_ids = 0, self.n, None
        rank, world = 0, 1
        if group is not None:
            import torch.distributed as dist

            rank, world = dist.get_rank(group), dist.get_world_size(group)
        condensed = condensed and self.n - len(set(self.basis.tolist())) >= world  # else: full layout on every rank
        if group is not None:
            if condensed:
                nonbasic = np.setdiff1d(np.arange(self.n), self.basis)
                lo, hi = column_range(nonbasic.size, rank, world)
                var_ids = nonbasic[lo:hi]
            else:
                j0, j1 = column_range(self.n, rank, world)
        self.counter = None
        self.optimum = None
        A_cols = A32[:, j0:j1] if var_ids is None else A32[:, var_ids]
        self._engine = TableauEngine.from_columns(A_cols, self.b, self.c, self.basis, j0, self.n, inv32=inv32,
                                                  bfs32=bfs32, log_cap=self._LOG_CAP, group=group, dual=True,
                                                  start32=start32, var_ids=var_ids, condensed=condensed)

    def solve(self, maxiters: int = 100):
        self.counter = 0
        ctrl = self._engine.run(maxiters)
        self.counter = int(ctrl.counter)
        self.basis[:] = self._engine.host_basis()
        if ctrl.status == _lib.DUAL_UNBOUNDED:
            raise DualIsUnboundedError
        if ctrl.status == _lib.OPTIMAL:
            self.optimum = True
        x, cost = self._engine.expand()
        return LinProgResult(x=x, basis=self.basis, cost=cost, iters=self.counter, optimum=self.optimum)
