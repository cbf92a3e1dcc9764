"""Device-resident solver state and the host side of the pivot loop.

``RevisedEngine`` keeps (A, A^T, b, c, B^-1, x_B, basis, var_state, control block, pivot log, scratch) in HBM
for the life of a solver object, enqueues iterations through liblpx and reads back one 128-byte control block
per chunk of iterations -- the only host round trip of the loop (north star: "no per-pivot host round-trip
beyond a scalar status flag").
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from .device import roundup, stream_ptr, to_device, to_device_2d, torch_cuda


class RevisedEngine:
    """State of one LP in revised form (explicit B^-1, primal_solvers.py:13-41) on the GPU."""

    #: iterations enqueued between two reads of the control block
    CHUNK = 64
    FIRST_CHUNK = 16

    def __init__(self, c32, A32, b32, basis, inv0, bfs0, var_state=None, lb=None, ub=None, log_cap=0, naive=False):
        torch = torch_cuda()
        self.lib = _lib.load()
        self.torch = torch
        m, n = A32.shape  # a host array, or a preprocessing.DeviceProblem whose matrix is in HBM already
        self.m, self.n = m, n
        # B^-1 is the matrix K1 streams: 4-KiB-aligned rows once it is large (see tableau.row_alignment)
        pad = lambda cols, unit: roundup(cols, 4096 // unit if cols * unit >= 32768 else (128 // unit if cols * unit >= 1024 else 16 // unit))
        self.ldA, self.ldAt, self.ldB = pad(n, 4), pad(m, 4), pad(m, 8)
        self.log_cap = int(log_cap)
        self.naive = bool(naive)
        self.host_ctrl = _lib.LpxCtrl()
        self._result = None  # host copy of the result region after a small solve (arena states only)
        self.arena = None
        if not hasattr(A32, "A_dev") and self._arena_bytes(lb is not None) <= self.ARENA_MAX:
            self._init_arena(c32, A32, b32, basis, inv0, bfs0, var_state, lb, ub)
            return
        if hasattr(A32, "A_dev"):
            if A32.A_dev.shape[1] == self.ldA:
                self.A = A32.A_dev
            else:  # re-pitch on the device
                self.A = torch.zeros((m, self.ldA), dtype=torch.float32, device="cuda")
                self.A[:, :n].copy_(A32.A_dev[:, :n])
        else:
            self.A = to_device_2d(A32.astype(np.float32, copy=False), self.ldA, torch.float32)
        self.At = torch.zeros((n, self.ldAt), dtype=torch.float32, device="cuda")
        self.At[:, :m].copy_(self.A[:, :n].t())
        self.b = to_device(b32.astype(np.float32, copy=False))
        self.c = to_device(c32.astype(np.float32, copy=False))
        self.Binv = to_device_2d(np.asarray(inv0, dtype=np.float64), self.ldB, torch.float64)
        self.xB = to_device(np.asarray(bfs0, dtype=np.float64))
        self.basis = to_device(np.asarray(basis, dtype=np.int64))
        if var_state is None:
            var_state = np.full(n, _lib.VS_LOWER, dtype=np.int32)
            var_state[np.asarray(basis, dtype=np.int64)] = _lib.VS_BASIC
        self.var_state = to_device(np.asarray(var_state, dtype=np.int32))
        self.lb = None if lb is None else to_device(np.asarray(lb, dtype=np.float64))
        self.ub = None if ub is None else to_device(np.asarray(ub, dtype=np.float64))
        self.ctrl = torch.zeros(128, dtype=torch.uint8, device="cuda")
        self.log = torch.zeros((max(self.log_cap, 1), 3), dtype=torch.int64, device="cuda")
        ws_bytes = self.lib.lpx_revised_workspace_bytes(m, n)
        self.workspace = torch.zeros(ws_bytes, dtype=torch.uint8, device="cuda")
        self.x = torch.zeros(n, dtype=torch.float64, device="cuda")
        self.cost = torch.zeros(1, dtype=torch.float64, device="cuda")
        if self.naive:  # float32 re-inversion scratch: W = [B | I]
            nbytes = self.lib.lpx_naive_workspace_bytes(m)
            self.naive_workspace = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        self._bind_state(ws_bytes)
        _lib.check(self.lib.lpx_ctrl_init(self.s.ctrl, 1, stream_ptr()))

    #: states up to this many bytes live in ONE device allocation filled by ONE upload (a lone small LP is latency-bound:
    #: a dozen allocations, fills and uploads cost ~0.3 ms, several times the solve); larger ones keep a tensor per array
    ARENA_MAX = (4 << 20) if os.environ.get("LPX_ARENA", "1") != "0" else -1  # LPX_ARENA=0: a tensor per array

    def _arena_items(self, bounded):
        """(name, dtype, shape) of every array of the state, in arena order; the RESULT region -- control block, cost, x,
        basis, var_state, x_B -- is contiguous so that one download brings back everything solve() returns."""
        m, n = self.m, self.n
        ws = int(self.lib.lpx_revised_workspace_bytes(m, n))
        items = [("ctrl", np.uint8, (128,)), ("cost", np.float64, (1,)), ("x", np.float64, (n,)), ("basis", np.int64, (m,)),
                 ("var_state", np.int32, (n,)), ("xB", np.float64, (m,)),
                 ("A", np.float32, (m, self.ldA)), ("At", np.float32, (n, self.ldAt)), ("b", np.float32, (m,)),
                 ("c", np.float32, (n,)), ("Binv", np.float64, (m, self.ldB)),
                 ("log", np.int64, (max(self.log_cap, 1), 3)), ("workspace", np.uint8, (ws,))]
        if bounded:
            items += [("lb", np.float64, (n,)), ("ub", np.float64, (n,))]
        if self.naive:
            items.append(("naive_workspace", np.uint8, (int(self.lib.lpx_naive_workspace_bytes(m)),)))
        return items

    def _arena_layout(self, bounded):
        offs, off = {}, 0
        for name, dt, shape in self._arena_items(bounded):
            nb = int(np.prod(shape)) * np.dtype(dt).itemsize
            offs[name] = (off, nb, dt, shape)
            off += roundup(max(nb, 1), 256)
        return offs, off

    def _arena_bytes(self, bounded):
        return self._arena_layout(bounded)[1]

    def _init_arena(self, c32, A32, b32, basis, inv0, bfs0, var_state, lb, ub):
        torch = self.torch
        m, n = self.m, self.n
        offs, total = self._arena_layout(lb is not None)
        buf = np.zeros(total, dtype=np.uint8)

        def host(name):
            o, nb, dt, shape = offs[name]
            return np.ndarray(shape, dtype=dt, buffer=buf, offset=o)

        A32 = A32.astype(np.float32, copy=False)
        host("A")[:, :n] = A32
        host("At")[:, :m] = A32.T
        host("b")[:] = b32
        host("c")[:] = c32
        host("Binv")[:, :m] = inv0
        host("xB")[:] = bfs0
        basis = np.asarray(basis, dtype=np.int64)
        host("basis")[:] = basis
        if var_state is None:
            vs = host("var_state")
            vs[:] = _lib.VS_LOWER
            vs[basis] = _lib.VS_BASIC
        else:
            host("var_state")[:] = var_state
        if lb is not None:
            host("lb")[:] = lb
            host("ub")[:] = ub
        c0 = _lib.LpxCtrl()  # what lpx_ctrl_init(ctrl, 1) leaves
        c0.f32_phase, c0.enter_j, c0.leave_r, c0.leave_var = 1, -1, -1, -1
        host("ctrl")[:] = np.frombuffer(bytes(c0), dtype=np.uint8)
        self.arena = torch.from_numpy(buf).to("cuda")  # one allocation, one upload
        self._arena_offs = offs  # tensor views of the arrays are made on first use (__getattr__): most are never touched
        if lb is None:
            self.lb = self.ub = None
        o_last, nb_last = offs["xB"][0], offs["xB"][1]
        self._result_span = (offs["ctrl"][0], o_last + nb_last)
        self._result_offs = {k: offs[k] for k in ("ctrl", "cost", "x", "basis", "var_state", "xB")}
        base = self.arena.data_ptr()
        self._bind_state(offs["workspace"][1], {k: base + v[0] for k, v in offs.items()})

    _TORCH_DTYPES = {np.uint8: "uint8", np.float32: "float32", np.float64: "float64", np.int64: "int64", np.int32: "int32"}

    def __getattr__(self, name):
        # only reached when normal lookup fails: an array of an arena state that nobody has asked for yet
        offs = self.__dict__.get("_arena_offs")
        if offs is None or name not in offs:
            raise AttributeError(name)
        o, nb, dt, shape = offs[name]
        t = self.arena[o:o + nb]
        t = (t if dt is np.uint8 else t.view(getattr(self.torch, self._TORCH_DTYPES[dt]))).view(shape)
        setattr(self, name, t)
        return t

    def _bind_state(self, ws_bytes, ptrs=None):
        """Fill the C struct; ptrs: device addresses by array name (arena states), else taken from the tensors."""
        m, n = self.m, self.n
        if ptrs is None:
            names = ["A", "At", "b", "c", "Binv", "xB", "basis", "var_state", "ctrl", "log", "workspace", "x", "cost"]
            names += ["lb", "ub"] if self.lb is not None else []
            names += ["naive_workspace"] if self.naive else []
            ptrs = {k: getattr(self, k).data_ptr() for k in names}
        s = self.s = _lib.LpxRevised()
        s.m, s.n, s.ldA, s.ldAt, s.ldB = m, n, self.ldA, self.ldAt, self.ldB
        s.A, s.At, s.b, s.c = ptrs["A"], ptrs["At"], ptrs["b"], ptrs["c"]
        s.Binv, s.xB, s.basis = ptrs["Binv"], ptrs["xB"], ptrs["basis"]
        s.var_state = ptrs["var_state"]
        s.lb = ptrs.get("lb")
        s.ub = ptrs.get("ub")
        s.ctrl = ptrs["ctrl"]
        s.log = ptrs["log"] if self.log_cap > 0 else None
        s.log_cap = self.log_cap
        s.workspace, s.workspace_bytes = ptrs["workspace"], ws_bytes
        if self.naive:
            s.naive_workspace, s.naive_workspace_bytes = ptrs["naive_workspace"], int(self.lib.lpx_naive_workspace_bytes(m))
        self._x_ptr, self._cost_ptr = ptrs["x"], ptrs["cost"]

    # ------------------------------------------------------------------ control block
    def read_ctrl(self):
        if self._result is not None:  # downloaded with the result region of the last small solve, still current
            return self.host_ctrl
        _lib.check(self.lib.lpx_read_ctrl(self.s.ctrl, C.byref(self.host_ctrl), stream_ptr()))
        return self.host_ctrl

    # ------------------------------------------------------------------ loop
    #: opt-in: refine B^-1 (lpx_revised_refine) every this many pivots inside run(); None = never (the reference's loop)
    refine_every = None
    #: opt-in: re-invert from scratch (lpx_revised_reinvert_f64) every this many pivots inside run(); None = never
    reinvert_every = None

    #: a lone small LP is launch-bound on the multi-kernel chain (~40 us per iteration whatever its size): when the state
    #: fits one CU's LDS the whole solve() call is ONE launch (lpx_revised_solve_small); LPX_SMALL_SOLVE=0 disables
    SMALL_SOLVE = os.environ.get("LPX_SMALL_SOLVE", "1") != "0"
    SMALL_LDS_MAX = 158 * 1024

    #: rows up to which the one-launch kernel beats the chain (tools/ctor_latency.py, tools/bounded_crossover.py on MI355X,
    #: per iteration: dual 6.5 / 14 us at m = 64 / 128 against the chain's 28 / 28; bounded, whose x_B is recomputed from
    #: scratch through an n-long chain along a row of A every iteration, 10 / 12 us at m = 64 / 80 with A beside the state in
    #: LDS, but 44 / 55 us at m = 96 / 128 where A no longer fits, against the chain's 37 / 35: it takes the kernel only
    #: while A fits)
    SMALL_MAX_ROWS = {_lib.KIND_PRIMAL: 128, _lib.KIND_DUAL: 128, _lib.KIND_BOUNDED: 128,
                      _lib.KIND_DUAL_BOUNDED: 4096}  # the opt-in bounded dual loop exists as the one-launch kernel only

    def fits_small(self, kind):
        return (self.SMALL_SOLVE and not self.naive and not self.refine_every and not self.reinvert_every
                and kind in self.SMALL_MAX_ROWS and self.m <= self.SMALL_MAX_ROWS[kind]
                and self.lib.lpx_revised_small_lds_bytes(self.m, self.n, kind) <= self.SMALL_LDS_MAX
                and (kind != _lib.KIND_BOUNDED or not self.BOUNDED_NEEDS_A_IN_LDS
                     or bool(self.lib.lpx_revised_small_a_in_lds(self.m, self.n, kind))))

    #: routing rule only (the kernel runs either way; tests switch it off to reach the bounded loops' global-memory path)
    BOUNDED_NEEDS_A_IN_LDS = True

    def run(self, kind, maxiters):
        """One ``solve(maxiters)`` call of the reference: returns the control block after the loop ended."""
        st = stream_ptr()
        last_refine = None
        last_reinvert = None
        self._result = None
        _lib.check(self.lib.lpx_ctrl_begin_solve(self.s.ctrl, st))
        if self.fits_small(kind):
            _lib.check(self.lib.lpx_revised_solve_small(C.byref(self.s), kind, int(maxiters), st))
            if self.arena is None:
                return self.read_ctrl()
            # arena state: x and cost are assembled right behind the solve (whatever its status: they are functions of
            # the state) and ONE download brings back the control block and everything solve() returns
            _lib.check(self.lib.lpx_revised_expand(C.byref(self.s), self._x_ptr, self._cost_ptr, st))
            return self._fetch_result()
        left = int(maxiters)
        ctrl = None
        first = True
        while left > 0:
            # short first chunk (eager launches): small solves end here; later chunks are long enough for
            # liblpx to replay them as hipGraphs
            k = min(left, self.FIRST_CHUNK if first else self.CHUNK)
            first = False
            _lib.check(self.lib.lpx_revised_iterate(C.byref(self.s), kind, k, st))
            left -= k
            ctrl = self.read_ctrl()
            if ctrl.status != _lib.RUNNING:
                break
            if self.refine_every and not self.naive:
                if last_refine is None:
                    last_refine = int(ctrl.pivots) - int(ctrl.counter)
                if int(ctrl.pivots) - last_refine >= int(self.refine_every):
                    self.refine(steps=1, update_xB=kind != _lib.KIND_BOUNDED)
                    last_refine = int(ctrl.pivots)
            if self.reinvert_every and not self.naive:
                if last_reinvert is None:
                    last_reinvert = int(ctrl.pivots) - int(ctrl.counter)
                if int(ctrl.pivots) - last_reinvert >= int(self.reinvert_every):
                    self.reinvert(update_xB=kind != _lib.KIND_BOUNDED)
                    last_reinvert = int(ctrl.pivots)
        if ctrl is None:
            ctrl = self.read_ctrl()
        return ctrl

    def touch(self):
        """The caller wrote to the state's tensors (or ran liblpx on them) behind the engine's back: drop the host copy."""
        self._result = None

    def _fetch_result(self):
        """One D2H copy of the arena's result region -> host views of (ctrl, cost, x, basis, var_state, x_B), kept until
        the next call that changes the device state."""
        o0, o1 = self._result_span
        raw = self.arena[o0:o1].cpu().numpy()
        res = {}
        for name, (o, nb, dt, shape) in self._result_offs.items():
            res[name] = np.ndarray(shape, dtype=dt, buffer=raw, offset=o - o0)
        C.memmove(C.byref(self.host_ctrl), res["ctrl"].ctypes.data, 128)
        self._result = res
        return self.host_ctrl

    def enqueue(self, kind, iters):
        """Enqueue iterations without reading the control block (bench / timing windows)."""
        self._result = None
        _lib.check(self.lib.lpx_revised_iterate(C.byref(self.s), kind, int(iters), stream_ptr()))

    def pivot(self, row, col):
        """Caller-chosen basis change; works whatever state the last solve() ended in (kernels gate on RUNNING)."""
        st = stream_ptr()
        self._result = None
        _lib.check(self.lib.lpx_ctrl_begin_solve(self.s.ctrl, st))
        f = self.lib.lpx_naive_pivot if self.naive else self.lib.lpx_revised_pivot
        _lib.check(f(C.byref(self.s), int(row), int(col), st))

    def tableau_row(self, row):
        """Row `row` of B^-1 A on the host (n float64): one device pass, O(n) download (special_solvers.py:69-71)."""
        out = self.torch.empty(self.n, dtype=self.torch.float64, device="cuda")
        _lib.check(self.lib.lpx_revised_tableau_row(C.byref(self.s), int(row), out.data_ptr(), stream_ptr()))
        return out.cpu().numpy()

    def bounded_update_bfs(self):
        self._result = None
        _lib.check(self.lib.lpx_bounded_update_bfs(C.byref(self.s), stream_ptr()))

    def refine(self, steps=1, update_xB=True):
        """Opt-in: `steps` Newton-Schulz steps on B^-1 in fp64 on the matrix cores (lpx_revised_refine).  Returns
        (max|I - B X| before, after).  Not part of the reference's algorithm; solve() never calls it."""
        torch = self.torch
        self._result = None
        nbytes = self.lib.lpx_refine_workspace_bytes(self.m)
        work = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        resid = torch.zeros(2, dtype=torch.float64, device="cuda")
        _lib.check(self.lib.lpx_revised_refine(C.byref(self.s), int(steps), int(bool(update_xB)), work.data_ptr(), nbytes,
                                               resid.data_ptr(), stream_ptr()))
        r = resid.cpu().numpy()
        return float(r[0]), float(r[1])

    def reinvert(self, update_xB=True):
        """Opt-in: B^-1 = inv(A[:, basis]) from scratch in fp64 (lpx_revised_reinvert_f64: blocked Gauss-Jordan, trailing
        updates on the matrix cores) and, with update_xB, x_B = B^-1 b.  Raises numpy's LinAlgError for a singular basis,
        as np.linalg.inv does (primal_solvers.py:92-94).  Not part of the reference's revised loop."""
        torch = self.torch
        self._result = None
        nbytes = self.lib.lpx_invert_workspace_bytes(self.m)
        work = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        info = torch.zeros(1, dtype=torch.int64, device="cuda")
        _lib.check(self.lib.lpx_revised_reinvert_f64(C.byref(self.s), int(bool(update_xB)), work.data_ptr(), nbytes,
                                                     info.data_ptr(), stream_ptr()))
        if int(info.item()) != 0:
            raise np.linalg.LinAlgError("Singular matrix")

    # ------------------------------------------------------------------ results
    def expand(self):
        """(x, cost) on the host: scatter + dot on the device, one D2H copy (primal_solvers.py:72-79)."""
        if self._result is not None:
            return self._result["x"].copy(), np.float64(self._result["cost"][0])
        _lib.check(self.lib.lpx_revised_expand(C.byref(self.s), self.x.data_ptr(), self.cost.data_ptr(), stream_ptr()))
        x = self.x.cpu().numpy()
        return x, np.float64(self.cost.cpu().numpy()[0])

    def reduced_costs(self, bounded=False):
        rc = self.torch.zeros(self.n, dtype=self.torch.float64, device="cuda")
        # the kernels gate on status == RUNNING; query on a running control block
        _lib.check(self.lib.lpx_revised_reduced_costs(C.byref(self.s), int(bounded), rc.data_ptr(), stream_ptr()))
        return rc.cpu().numpy()

    def host_basis(self):
        if self._result is not None:
            return self._result["basis"].copy()
        return self.basis.cpu().numpy()

    def host_var_state(self):
        if self._result is not None:
            return self._result["var_state"].copy()
        return self.var_state.cpu().numpy()

    def host_Binv(self):
        return self.Binv[:, : self.m].cpu().numpy()

    def host_xB(self):
        if self._result is not None:
            return self._result["xB"].copy()
        return self.xB.cpu().numpy()

    def host_log(self, count):
        count = min(int(count), self.log_cap)
        return self.log[:count].cpu().numpy()
