"""CPU suite: the C-ABI library loads and exports every symbol include/lpx.h declares; host logic that needs no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests._util import ROOT


def test_library_exports_every_declared_symbol():
    import linprog_amd  # noqa: F401
    from linprog_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "lpx.h")).read()
    declared = set(re.findall(r"\b(lpx_[a-z0-9_]+)\s*\(", hdr))
    L = _lib.load()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in lpx.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert L.lpx_version() == 1
    assert C.sizeof(_lib.LpxCtrl) == 128
    # size queries need no GPU
    assert L.lpx_revised_workspace_bytes(64, 128) > 0 and L.lpx_tableau_workspace_bytes(1000, 2000) > 0
    assert L.lpx_tableau_msg_len(1000) == 1004


def test_product_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from linprog_amd.exceptions import LpxBackendError
    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver

    with pytest.raises(LpxBackendError):
        PrimalRevisedSimplexSolver(np.array([0.0, 1.0]), np.array([[1.0, 1.0]]), np.array([1.0]), np.array([0]))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "linear-programming_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import oracle|from oracle)", src, re.M), f
                assert not re.search(r"#include.*oracle|liblpx_oracle|dlopen", src), f


def test_preprocessing_matches_reference_fixture():
    """diet example: general form -> standard form -> bound rows; shapes and values pinned by SURVEY Appendix D."""
    import json

    from linprog_amd.preprocessing import ProblemPreprocessingUtils as P

    cs = json.load(open(os.path.join(ROOT, "tests", "golden", "diet.json")))["cases"][0]
    a = {k: np.array(v, dtype=float) for k, v in cs["arrays"].items()}
    c, A, b = P.canonical_form_to_standard_form(a["c"], a["G"], a["h"])
    assert A.shape == (4, 10) and A.dtype == np.float32 and (b >= 0).all()
    lb = np.concatenate([a["lb"], np.zeros(4)])
    ub = np.concatenate([a["ub"], np.full(4, np.inf)])
    c1, A1, b1 = P.add_variables_bounds_to_coefficient_matrix(c, A, b, lb, ub)
    assert A1.shape == (6, 12) and c1.shape == (12,)
    assert A1[4, 4] == 1 and A1[4, 10] == -1 and b1[4] == np.float32(0.5)  # fish >= 0.5
    assert A1[5, 1] == 1 and A1[5, 11] == 1 and b1[5] == 1.0               # milk <= 1


def test_generators_are_partition_invariant():
    from tests._util import gen

    full = gen.gen_primal_columns(40, 100, 7, 0, 100)[0]
    parts = np.hstack([gen.gen_primal_columns(40, 100, 7, j, min(j + 13, 100))[0] for j in range(0, 100, 13)])
    assert np.array_equal(full, parts)
    assert np.array_equal(full[:, 60:], np.eye(40, dtype=np.float32))


def test_iterate_without_a_device_fails_cleanly_and_repeatably():
    """No GPU here: a loop entry point whose arguments pass validation must come back with LPX_E_HIP from the first HIP
    call (never crash, never leave library state behind that makes the next call behave differently)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from linprog_amd import _lib

    L = _lib.load()
    m, n = 8, 16
    ws = L.lpx_tableau_workspace_bytes(m, n)
    buf = (C.c_char * (ws + 64))()
    T = (C.c_double * ((m + 1) * (n + 2)))()
    ints = (C.c_int64 * 64)()
    base = C.addressof(buf) + (-C.addressof(buf)) % 16
    s = _lib.LpxTableau()
    s.m, s.n, s.ld, s.col0, s.n_global, s.pricing = m, n, n + 2, 0, n, 0
    s.T, s.basis, s.var_state, s.ctrl = C.addressof(T), C.addressof(ints), C.addressof(ints), C.addressof(ints)
    s.workspace, s.workspace_bytes = base, ws
    for _ in range(2):
        rc = L.lpx_tableau_iterate(C.byref(s), 64, None)
        assert rc in (-2, -1), rc  # LPX_E_HIP (no device) -- or LPX_E_ARG if alignment of the fake buffers is refused
        assert L.lpx_last_error()
    assert L.lpx_tableau_iterate(C.byref(s), 0, None) == 0  # nothing to do: no HIP call at all


def test_every_launchable_kernel_has_device_code_in_its_translation_unit():
    """tools/check_kernels.py on the built objects: the host stub of every kernel instance a translation unit can launch
    has a kernel descriptor in that unit's gfx950 code object (round 2's abort: a template instance that the host side of
    a file referenced and its device side no longer instantiated)."""
    import glob
    import subprocess
    import sys

    objs = sorted(glob.glob(os.path.join(ROOT, "linear-programming_amd", "csrc", "*.o")))
    if not objs:
        pytest.skip("objects not built in this tree")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kernels.py")] + objs, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 without device code" in r.stdout


def test_selfcheck_without_a_device_is_an_error_code_not_a_crash():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from linprog_amd import _lib

    L = _lib.load()
    n = C.c_int64(-1)
    assert L.lpx_selfcheck(C.byref(n)) in (-2, -1) and n.value == 0 and L.lpx_last_error()


def test_chain_selection_is_host_logic_and_follows_the_documented_thresholds(monkeypatch):
    """lpx_tableau_chain_kind / lpx_tableau_block_size / lpx_tableau_pipelined read sizes, pointers and the environment
    only (no device): which launch chain lpx_tableau_iterate runs for a state, as include/lpx.h documents it."""
    import ctypes as C

    from linprog_amd import _lib

    for name in ("LPX_TABLEAU_FUSED", "LPX_TABLEAU_SMALL_CHAIN", "LPX_TABLEAU_BLOCK", "LPX_TABLEAU_BLOCK_MIN_MB",
                 "LPX_TABLEAU_PIPELINE_MIN_MB"):
        monkeypatch.delenv(name, raising=False)
    L = C.CDLL(_lib.LIB_PATH)
    L.lpx_tableau_chain_kind.argtypes = L.lpx_tableau_block_size.argtypes = L.lpx_tableau_pipelined.argtypes = [C.POINTER(_lib.LpxTableau)]
    L.lpx_tableau_block_size.restype = C.c_int64

    def state(m, carried, condensed=True, alt=True, ratio=_lib.RATIO_TEXTBOOK):
        s = _lib.LpxTableau()
        s.m, s.n, s.ld, s.col0, s.n_global = m, carried, carried + 2, 0, (carried + m if condensed else carried)
        s.T = 0x1000  # never dereferenced by these calls
        s.T_alt = 0x2000 if alt else None
        s.slot_var = s.var_slot = (0x3000 if condensed else None)
        s.ratio_rule = ratio
        return s

    kind = lambda s: L.lpx_tableau_chain_kind(C.byref(s))
    assert kind(state(1000, 1000)) == _lib.CHAIN_SMALL  # BASELINE config 2
    assert kind(state(1023, 1023)) == _lib.CHAIN_SMALL and kind(state(1024, 1000)) == _lib.CHAIN_FUSED
    assert kind(state(64, 1024)) == _lib.CHAIN_FUSED  # one carried column too many for a workgroup
    assert kind(state(1000, 1000, alt=False)) == _lib.CHAIN_EAGER
    assert kind(state(1000, 1000, ratio=_lib.RATIO_HARRIS)) == _lib.CHAIN_EAGER
    assert kind(state(300, 700, condensed=False)) == _lib.CHAIN_SMALL
    big = state(20000, 20000)  # BASELINE config 4, condensed: 3.2 GB
    assert kind(big) == _lib.CHAIN_PIPELINED and L.lpx_tableau_block_size(C.byref(big)) == 32 and L.lpx_tableau_pipelined(C.byref(big)) == 1
    assert kind(state(20000, 20000, alt=False)) == _lib.CHAIN_DEFERRED
    mid = state(10000, 10000)  # 800 MB: deferred in place whether or not there is a second buffer
    assert kind(mid) == _lib.CHAIN_DEFERRED and L.lpx_tableau_pipelined(C.byref(mid)) == 0
    assert L.lpx_tableau_block_size(C.byref(state(3000, 3000))) == 16 and L.lpx_tableau_block_size(C.byref(state(2000, 2000))) == 0
    monkeypatch.setenv("LPX_TABLEAU_SMALL_CHAIN", "0")
    assert kind(state(1000, 1000)) == _lib.CHAIN_FUSED
    monkeypatch.setenv("LPX_TABLEAU_FUSED", "0")
    assert kind(state(1000, 1000)) == _lib.CHAIN_EAGER
    monkeypatch.setenv("LPX_TABLEAU_PIPELINE_MIN_MB", "-1")
    assert kind(big) == _lib.CHAIN_DEFERRED
    assert L.lpx_tableau_chain_kind(None) == -1
