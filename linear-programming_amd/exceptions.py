"""Error convention of the solver classes (reference: linprog/exceptions.py:1-22).

The device status word maps 1:1 onto these; the Python host raises them synchronously out of
``solve()`` exactly where the reference does.
"""


class BasisIsPrimalInfeasibleError(Exception):
    """Starting basis gives a negative basic variable (primal_solvers.py:43-48)."""


class BasisIsDualInfeasibleError(Exception):
    """Starting basis is not dual feasible (dual_solvers.py:14-16)."""


class PrimalIsUnboundedError(Exception):
    """No positive entry in the entering direction (primal_solvers.py:130-132)."""


class PrimalIsInfeasibleError(Exception):
    """Phase one ends with positive cost (special_solvers.py:55-57)."""


class DualIsUnboundedError(Exception):
    """No negative entry in the leaving row (dual_solvers.py:41-44; special_solvers.py:435-436)."""


class DualIsInfeasibleError(Exception):
    """Declared by the reference, never raised (exceptions.py:21-22)."""


class LpxBackendError(RuntimeError):
    """The HIP library is missing, failed to load, or a HIP call failed.  There is no CPU fallback."""
