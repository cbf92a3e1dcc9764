"""Host-side problem construction (reference: linprog/preprocessing.py:4-69).

``preprocess_problem`` sits on the drop-in boundary: every solver constructor pushes (c, A, b) through it,
which fixes the numeric inputs of the device path (float32 values, rows with b<0 negated).  The other two
helpers are one-off host-side constructions used by the two-phase driver.
"""
import numpy as np


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
