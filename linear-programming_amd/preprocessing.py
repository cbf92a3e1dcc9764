"""Problem construction (reference: linprog/preprocessing.py:4-69).

``preprocess_problem`` sits on the drop-in boundary: every solver constructor pushes (c, A, b) through it,
which fixes the numeric inputs of the device path (float32 values, rows with b<0 negated).  The other two
helpers keep the reference's host signatures; ``standard_form_on_device`` does the same construction -- slack columns,
bound rows, sign normalisation, phase-one artificial block -- as one kernel straight into HBM (``lpx_build_standard_form``),
which is what the two-phase driver uses: the O(rows x columns) matrix is never assembled on the host.
"""
import numpy as np


class DeviceProblem:
    """A standard-form LP whose matrix lives in HBM only: ``A_dev`` is a (m, ld) float32 device tensor (values already
    float32-rounded and sign-normalised, preprocessing.py:6-15), ``b32`` / ``c32`` are host vectors (O(m + n)).  The solver
    constructors accept one in place of ``A``: nothing is uploaded again, and the host sees matrix entries only where the
    reference's constructor needs them (the m x m basis block for its float32 ``np.linalg.inv``)."""

    def __init__(self, A_dev, m, n, b32, c32, artificial_from=None):
        self.A_dev, self.m, self.n = A_dev, int(m), int(n)
        self.b32 = np.asarray(b32, dtype=np.float32)
        self.c32 = np.asarray(c32, dtype=np.float32)
        self.artificial_from = artificial_from  # columns [artificial_from, artificial_from + m) are exactly +I, or None
        self._host = None

    @property
    def shape(self):
        return (self.m, self.n)

    def columns(self, idx):
        """A[:, idx] on the host (float32): an m x len(idx) download."""
        import torch

        sel = torch.as_tensor(np.asarray(idx, dtype=np.int64), device=self.A_dev.device)
        return self.A_dev[:, : self.n].index_select(1, sel).cpu().numpy()

    def host(self):
        """The whole matrix on the host (cached): for callers that read ``solver.A``."""
        if self._host is None:
            self._host = self.A_dev[:, : self.n].cpu().numpy()
        return self._host

    def basis_is_artificial_identity(self, basis):
        a = self.artificial_from
        return a is not None and np.array_equal(np.asarray(basis), np.arange(a, a + self.m))

    def select_rows(self, keep):
        """Rows ``keep`` (boolean mask) of the problem, still on the device (redundant-row removal of phase one)."""
        import torch

        keep = np.asarray(keep, dtype=bool)
        sel = torch.as_tensor(np.flatnonzero(keep), device=self.A_dev.device)
        return DeviceProblem(self.A_dev.index_select(0, sel).contiguous(), int(keep.sum()), self.n, self.b32[keep], self.c32)

    def without_artificials(self, n_core):
        """The same rows with only the first ``n_core`` columns (drops the phase-one artificial block): a view."""
        return DeviceProblem(self.A_dev, self.m, n_core, self.b32, self.c32[:n_core])


class ProblemPreprocessingUtils:
    @staticmethod
    def preprocess_problem(c, A, b):
        """float32 cast and sign normalisation of rows with b < 0 (preprocessing.py:6-15)."""
        c32 = np.array(c).astype(np.float32)
        A32 = np.array(A).astype(np.float32)
        b32 = np.array(b).astype(np.float32)
        flip = b32 < 0
        A32[flip] *= -1
        b32[flip] *= -1
        return c32, A32, b32

    @staticmethod
    def canonical_form_to_standard_form(c, G, h):
        """min cx s.t. Gx <= h  ->  [G I] with slack columns (preprocessing.py:17-28)."""
        c32 = np.array(c).astype(np.float32)
        G32 = np.array(G).astype(np.float32)
        h32 = np.array(h).astype(np.float32)
        k = h32.shape[0]
        A = np.hstack([G32, np.eye(k)])
        cc = np.concatenate([c32, np.zeros(k)])
        return ProblemPreprocessingUtils.preprocess_problem(cc, A, np.array(h32))

    @staticmethod
    def standard_form_on_device(c, A=None, b=None, G=None, h=None, lb=None, ub=None, artificials=False):
        """``canonical_form_to_standard_form`` (:17-28) + ``add_variables_bounds_to_coefficient_matrix`` (:30-69) +
        ``preprocess_problem`` (:6-15) in one device kernel: min c x s.t. A x = b, G x <= h, lb <= x <= ub becomes
        ``[[A 0 0], [G I 0], [E 0 +-I]] x' = b'`` with b' >= 0, written straight into HBM.  ``artificials``: also the +I
        block of PhaseOneSimplexSolver's auxiliary problem (special_solvers.py:48-52) and its 0 / 1 costs.  Same values as
        the host helpers applied in the reference's order (tests compare them entry for entry).  lb / ub cover the
        structural variables; the slack variables of G are 0 <= s < inf as in simplex.py:66-73."""
        import ctypes as C  # noqa: F401

        from . import _lib
        from .device import roundup, stream_ptr, to_device, torch_cuda

        torch = torch_cuda()
        lib = _lib.load()
        c32 = np.array(c).astype(np.float32)
        nx = c32.shape[0]
        A32 = None if A is None else np.ascontiguousarray(np.array(A).astype(np.float32))
        G32 = None if G is None else np.ascontiguousarray(np.array(G).astype(np.float32))
        b32 = np.zeros(0, np.float32) if b is None else np.array(b).astype(np.float32)
        h32 = np.zeros(0, np.float32) if h is None else np.array(h).astype(np.float32)
        me, mi = (0 if A32 is None else A32.shape[0]), (0 if G32 is None else G32.shape[0])
        lb32 = np.repeat(np.float32(0.0), nx) if lb is None else np.array(lb).astype(np.float32)
        ub32 = np.repeat(np.float32(np.inf), nx) if ub is None else np.array(ub).astype(np.float32)
        has_lb, has_ub = ~np.isclose(lb32, 0.0), ~np.isclose(ub32, np.inf)  # :47-48
        bvar = np.concatenate([np.flatnonzero(has_lb), np.flatnonzero(has_ub)]).astype(np.int64)
        bsign = np.concatenate([-np.ones(int(has_lb.sum())), np.ones(int(has_ub.sum()))]).astype(np.float32)
        bval = np.concatenate([lb32[has_lb], ub32[has_ub]]).astype(np.float32)
        nb = bvar.size
        rows, core = me + mi + nb, nx + mi + nb
        na = rows if artificials else 0
        cols = core + na
        ld = roundup(cols, 4)
        out = torch.zeros((rows, ld), dtype=torch.float32, device="cuda")
        out_b = torch.zeros(rows, dtype=torch.float32, device="cuda")
        dev = lambda a: None if a is None or a.size == 0 else to_device(a)  # noqa: E731
        keep = [dev(A32), dev(b32), dev(G32), dev(h32), dev(bvar), dev(bsign), dev(bval)]
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        _lib.check(lib.lpx_build_standard_form(me, mi, nx, nb, na, *(ptr(t) for t in keep), out.data_ptr(), ld,
                                               out_b.data_ptr(), stream_ptr()))
        bb = out_b.cpu().numpy()  # also orders the kernel before `keep` is released
        cc = np.concatenate([c32, np.zeros(mi + nb, np.float32)])
        prob = DeviceProblem(out, rows, cols, bb, cc, artificial_from=core if artificials else None)
        if artificials:  # phase-one costs (special_solvers.py:48); the problem itself is the same rows without that block
            prob.c32 = np.concatenate([np.zeros(core, np.float32), np.ones(rows, np.float32)])
            prob.core = DeviceProblem(out, rows, core, bb, cc)
        return prob

    @staticmethod
    def add_variables_bounds_to_coefficient_matrix(c, A, b, lb, ub):
        """Turn lb <= x <= ub into extra equality rows with surplus/slack columns (preprocessing.py:30-69)."""
        c32 = np.array(c).astype(np.float32)
        A32 = np.array(A).astype(np.float32)
        b32 = np.array(b).astype(np.float32)
        ncols = A32.shape[1]
        lb32 = np.repeat(0.0, ncols) if lb is None else np.array(lb).astype(np.float32)
        ub32 = np.repeat(np.inf, ncols) if ub is None else np.array(ub).astype(np.float32)

        has_lb = ~np.isclose(lb32, 0.0)
        has_ub = ~np.isclose(ub32, np.inf)
        # one new row and one new column per finite bound: x_i - s = lb_i, then x_i + s = ub_i
        rows = [(i, -1.0) for i in np.flatnonzero(has_lb)] + [(i, 1.0) for i in np.flatnonzero(has_ub)]
        m0, k = A32.shape[0], len(rows)
        out = np.zeros((m0 + k, ncols + k), dtype=np.float64 if k else A32.dtype)
        out[:m0, :ncols] = A32
        for t, (i, sgn) in enumerate(rows):
            out[m0 + t, i] += 1
            out[m0 + t, ncols + t] += sgn
        bb = np.concatenate([b32, lb32[has_lb], ub32[has_ub]])
        cc = np.concatenate([c32, np.zeros(k)])
        return ProblemPreprocessingUtils.preprocess_problem(cc, out, bb)
