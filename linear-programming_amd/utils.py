"""Host scalar helpers (reference: linprog/utils.py:5-40).

``primal_simplex_div`` / ``dual_simplex_div`` are the ratio-test arithmetic; on the device path they live
inside the ratio kernels (csrc/lpx_core.hip: k_direction / lpx_psd, lpx_dsd in csrc/lpx_device.hpp).  The NumPy forms here exist for API completeness (the
reference exports them) and are array expressions rather than per-element Python calls.
"""
from math import factorial

import numpy as np


def primal_simplex_div(n, d):
    """n/d where d > 0, +inf elsewhere (utils.py:5-10)."""
    n = np.asarray(n)
    d = np.asarray(d)
    out = np.full(np.broadcast(n, d).shape, np.inf, dtype=np.result_type(n, d, np.float32))
    np.divide(n, d, out=out, where=d > 0)
    return out


def dual_simplex_div(n, d):
    """(-n)/d where d < 0, +inf elsewhere (utils.py:12-17)."""
    n = np.asarray(n)
    d = np.asarray(d)
    out = np.full(np.broadcast(n, d).shape, np.inf, dtype=np.result_type(n, d, np.float32))
    np.divide(-n, d, out=out, where=d < 0)
    return out


def get_bounds_on_bfs(A, b):
    """m! * alpha^(m-1) * beta, a bound on |x_i| over all basic feasible solutions (utils.py:20-40).

    Kept in the reference's own arithmetic (Python int times NumPy scalars of A's dtype), including its
    overflow behaviour: ``inf`` for m around 35 with float32 inputs and OverflowError for m >= 171.
    """
    m = A.shape[0]
    alpha = np.abs(A).max()
    beta = np.abs(b).max()
    return factorial(m) * alpha ** (m - 1) * beta
