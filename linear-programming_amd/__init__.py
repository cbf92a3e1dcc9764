"""linprog_amd -- MI355X-native dense simplex pivot path behind the API of agmuth/linear-programming.

    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver   # was: from linprog.primal_solvers ...

Submodules mirror the reference package (primal_solvers, dual_solvers, special_solvers, simplex, data_classes,
exceptions, preprocessing, utils).  The pivot loop runs on the GPU through liblpx.so (include/lpx.h); there is
no CPU fallback.
"""
from .data_classes import LinProgResult  # noqa: F401
from .exceptions import *  # noqa: F401,F403

__all__ = ["LinProgResult"]
