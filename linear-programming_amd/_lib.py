"""ctypes binding of liblpx.so (include/lpx.h) -- the only route from the Python host to the GPU.

There is no CPU fallback: if the library is missing or no HIP device is visible, every solver constructor
raises LpxBackendError.  ``python __graft_entry__.py`` (or ``make -C linear-programming_amd/csrc``) builds
the library in-tree for gfx950.
"""
import ctypes as C
import os

from .exceptions import LpxBackendError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liblpx.so")

# status codes (lpx.h)
RUNNING, OPTIMAL, PRIMAL_UNBOUNDED, DUAL_UNBOUNDED, NOT_NONBASIC = 0, 1, 2, 3, 4
VS_BASIC, VS_LOWER, VS_UPPER, VS_EXCLUDED = 0, 1, 2, 3
PRICE_FIRST_NEGATIVE, PRICE_DANTZIG = 0, 1
RATIO_TEXTBOOK, RATIO_HARRIS = 0, 1
CHAIN_EAGER, CHAIN_FUSED, CHAIN_SMALL, CHAIN_DEFERRED, CHAIN_PIPELINED = 0, 1, 2, 3, 4
KIND_PRIMAL, KIND_DUAL, KIND_BOUNDED, KIND_PRIMAL_NAIVE, KIND_DUAL_NAIVE, KIND_DUAL_BOUNDED = 0, 1, 2, 3, 4, 5

vp = C.c_void_p
i64 = C.c_int64


class LpxCtrl(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("f32_phase", C.c_int32),
        ("counter", i64), ("pivots", i64), ("enter_j", i64), ("leave_r", i64), ("leave_var", i64),
        ("price_val", C.c_double), ("ratio_val", C.c_double), ("aux_val", C.c_double),
        ("flip", C.c_int32), ("leave_state", C.c_int32),
        ("enter_slot", i64), ("reserved", i64 * 5),
    ]


assert C.sizeof(LpxCtrl) == 128


class LpxRevised(C.Structure):
    _fields_ = [
        ("m", i64), ("n", i64), ("ldA", i64), ("ldAt", i64), ("ldB", i64),
        ("A", vp), ("At", vp), ("b", vp), ("c", vp),
        ("Binv", vp), ("xB", vp), ("basis", vp), ("var_state", vp),
        ("lb", vp), ("ub", vp), ("ctrl", vp),
        ("log", vp), ("log_cap", i64),
        ("workspace", vp), ("workspace_bytes", i64),
        ("naive_workspace", vp), ("naive_workspace_bytes", i64),
    ]


class LpxTableau(C.Structure):
    _fields_ = [
        ("m", i64), ("n", i64), ("ld", i64), ("col0", i64), ("n_global", i64), ("pricing", i64),
        ("T", vp), ("basis", vp), ("var_state", vp), ("ctrl", vp),
        ("log", vp), ("log_cap", i64),
        ("workspace", vp), ("workspace_bytes", i64),
        ("slot_var", vp), ("var_slot", vp), ("T_alt", vp),
        ("ratio_rule", i64), ("ratio_tol", C.c_double),
    ]


_SIGS = {
    "lpx_version": (C.c_int, []),
    "lpx_last_error": (C.c_char_p, []),
    "lpx_selfcheck": (C.c_int, [C.POINTER(i64)]),
    "lpx_ctrl_init": (C.c_int, [vp, C.c_int32, vp]),
    "lpx_ctrl_begin_solve": (C.c_int, [vp, vp]),
    "lpx_read_ctrl": (C.c_int, [vp, C.POINTER(LpxCtrl), vp]),
    "lpx_profile_k1": (C.c_int, [C.c_int]),
    "lpx_profile_k1_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(i64)]),
    "lpx_revised_workspace_bytes": (i64, [i64, i64]),
    "lpx_revised_iterate": (C.c_int, [C.POINTER(LpxRevised), C.c_int, i64, vp]),
    "lpx_revised_small_lds_bytes": (i64, [i64, i64, C.c_int]),
    "lpx_revised_small_a_in_lds": (C.c_int, [i64, i64, C.c_int]),
    "lpx_revised_solve_small": (C.c_int, [C.POINTER(LpxRevised), C.c_int, i64, vp]),
    "lpx_revised_solve_small_batch": (C.c_int, [vp, i64, i64, i64, C.c_int, i64, vp, vp, vp]),
    "lpx_revised_pivot": (C.c_int, [C.POINTER(LpxRevised), i64, i64, vp]),
    "lpx_naive_workspace_bytes": (i64, [i64]),
    "lpx_revised_reinvert": (C.c_int, [C.POINTER(LpxRevised), vp]),
    "lpx_naive_pivot": (C.c_int, [C.POINTER(LpxRevised), i64, i64, vp]),
    "lpx_revised_expand": (C.c_int, [C.POINTER(LpxRevised), vp, vp, vp]),
    "lpx_revised_tableau_row": (C.c_int, [C.POINTER(LpxRevised), i64, vp, vp]),
    "lpx_bounded_update_bfs": (C.c_int, [C.POINTER(LpxRevised), vp]),
    "lpx_revised_reduced_costs": (C.c_int, [C.POINTER(LpxRevised), C.c_int, vp, vp]),
    "lpx_pd_admissible": (C.c_int, [C.POINTER(LpxRevised), vp, vp, i64, vp, vp]),
    "lpx_pd_dual_step": (C.c_int, [C.POINTER(LpxRevised), vp, vp, vp, i64, vp, vp, vp]),
    "lpx_build_standard_form": (C.c_int, [i64, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp]),
    "lpx_pd_small_lds_bytes": (i64, [i64, i64]),
    "lpx_pd_solve_small": (C.c_int, [C.POINTER(LpxRevised), vp, vp, i64, i64, i64, vp, vp, vp]),
    "lpx_tableau_workspace_bytes": (i64, [i64, i64]),
    "lpx_tableau_iterate": (C.c_int, [C.POINTER(LpxTableau), i64, vp]),
    "lpx_tableau_msg_len": (i64, [i64]),
    "lpx_tableau_block_size": (i64, [C.POINTER(LpxTableau)]),
    "lpx_tableau_pipelined": (C.c_int, [C.POINTER(LpxTableau)]),
    "lpx_tableau_chain_kind": (C.c_int, [C.POINTER(LpxTableau)]),
    "lpx_tableau_propose": (C.c_int, [C.POINTER(LpxTableau), vp, C.c_int, vp]),
    "lpx_tableau_commit": (C.c_int, [C.POINTER(LpxTableau), vp, i64, vp]),
    "lpx_tableau_commit_begin": (C.c_int, [C.POINTER(LpxTableau), vp, i64, vp, vp]),
    "lpx_tableau_commit_update": (C.c_int, [C.POINTER(LpxTableau), vp]),
    "lpx_tableau_block_begin": (C.c_int, [C.POINTER(LpxTableau), vp]),
    "lpx_tableau_block_propose": (C.c_int, [C.POINTER(LpxTableau), i64, vp, vp]),
    "lpx_tableau_block_commit": (C.c_int, [C.POINTER(LpxTableau), i64, vp, i64, vp]),
    "lpx_tableau_block_flush": (C.c_int, [C.POINTER(LpxTableau), i64, vp]),
    "lpx_dist_unique_id": (C.c_int, [vp]),
    "lpx_dist_comm_create": (C.c_int, [vp, i64, i64, C.POINTER(vp)]),
    "lpx_dist_comm_destroy": (C.c_int, [vp]),
    "lpx_tableau_dist_iterate": (C.c_int, [C.POINTER(LpxTableau), vp, i64, i64, i64, vp, vp, vp]),
    "lpx_tableau_dual_dist_iterate": (C.c_int, [C.POINTER(LpxTableau), vp, i64, i64, i64, vp, vp, vp]),
    "lpx_tableau_dual_block_begin": (C.c_int, [C.POINTER(LpxTableau), vp]),
    "lpx_tableau_dual_block_propose": (C.c_int, [C.POINTER(LpxTableau), i64, vp, vp]),
    "lpx_tableau_dual_block_commit": (C.c_int, [C.POINTER(LpxTableau), i64, vp, i64, vp]),
    "lpx_refine_workspace_bytes": (i64, [i64]),
    "lpx_revised_refine": (C.c_int, [C.POINTER(LpxRevised), i64, C.c_int, vp, i64, vp, vp]),
    "lpx_invert_workspace_bytes": (i64, [i64]),
    "lpx_dinverse": (C.c_int, [i64, vp, i64, vp, i64, vp, i64, vp, vp]),
    "lpx_revised_reinvert_f64": (C.c_int, [C.POINTER(LpxRevised), C.c_int, vp, i64, vp, vp]),
    "lpx_dgemm": (C.c_int, [C.c_int, i64, i64, i64, C.c_double, vp, i64, vp, i64, C.c_double, vp, i64, C.c_double, vp,
                            i64, vp]),
    "lpx_tableau_dual_iterate": (C.c_int, [C.POINTER(LpxTableau), i64, vp]),
    "lpx_tableau_dual_propose": (C.c_int, [C.POINTER(LpxTableau), vp, vp]),
    "lpx_tableau_dual_commit": (C.c_int, [C.POINTER(LpxTableau), vp, i64, vp]),
    "lpx_tableau_dual_commit_begin": (C.c_int, [C.POINTER(LpxTableau), vp, i64, C.c_int, vp, vp]),
    "lpx_tableau_expand": (C.c_int, [C.POINTER(LpxTableau), vp, vp, vp, vp]),
    "lpx_batched_primal_solve": (C.c_int, [i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp]),
    "lpx_batched_dual_solve": (C.c_int, [i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp]),
    "lpx_batched_primal_solve_ineq": (C.c_int, [i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp]),
    "lpx_batched_tableau_solve": (C.c_int, [i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp]),
    "lpx_batched_tableau_solve_ineq": (C.c_int, [i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp]),
    "lpx_batched_tableau_dual_solve": (C.c_int, [i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp]),
    "lpx_batched_tableau_waves": (C.c_int, [i64, i64]),
    "lpx_batched_tableau_solve_from": (C.c_int, [i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp]),
    "lpx_pivot_update": (C.c_int, [vp, i64, i64, i64, i64, vp, vp, vp]),
    "lpx_pivot_update_static": (C.c_int, [vp, i64, i64, i64, i64, vp, vp]),
    "lpx_eta": (C.c_int, [vp, i64, i64, vp, vp]),
    "lpx_reduce_scratch_bytes": (i64, []),
    "lpx_price_first_negative": (C.c_int, [vp, vp, i64, vp, vp, vp, vp]),
    "lpx_price_argmax": (C.c_int, [vp, i64, vp, vp, vp, vp]),
    "lpx_ratio_primal": (C.c_int, [vp, vp, i64, C.c_int, vp, vp, vp, vp]),
    "lpx_ratio_dual": (C.c_int, [vp, vp, vp, i64, C.c_int, vp, vp, vp, vp]),
    "lpx_ratio_bounded": (C.c_int, [vp, vp, vp, vp, i64, C.c_double, vp, vp, vp, vp]),
    "lpx_dual_row_select": (C.c_int, [vp, i64, vp, vp, vp, vp]),
}

EXPORTS = tuple(_SIGS)

_lib = None


def load():
    """Load liblpx.so once; raise LpxBackendError (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm bundles its own HIP runtime; it must be the one already in the process when liblpx's
        # libamdhip64 dependency is resolved, or the two would not share device state.
        import torch  # noqa: F401

        if not os.path.exists(LIB_PATH):
            raise LpxBackendError(
                f"{LIB_PATH} not found: build it with `make -C linear-programming_amd/csrc` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:
            raise LpxBackendError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _SIGS.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
        if os.environ.get("LPX_SELFCHECK") == "1":
            selfcheck()
    return _lib


def selfcheck():
    """Resolve every kernel of liblpx on the current device (lpx.h, lpx_selfcheck); returns the number resolved, raises
    LpxBackendError naming the first kernel without device code."""
    n = i64(0)
    check(load().lpx_selfcheck(C.byref(n)))
    return int(n.value)


def check(rc):
    if rc != 0:
        msg = load().lpx_last_error()
        raise LpxBackendError(f"liblpx error {rc}: {msg.decode() if msg else '?'}")
