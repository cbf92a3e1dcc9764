"""Shared helpers for the parity tests: golden loading, generators, comparison bars."""
import importlib.util
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")

# seeded generators: loaded by path so CPU-only tests do not need the GPU package to import
_spec = importlib.util.spec_from_file_location("lpx_problems", os.path.join(ROOT, "linear-programming_amd", "problems.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)

def gen_columns(m, n, seed):
    """bench.py's generator (float32 column blocks) in the (c, A, b, basis) order of the others."""
    A, b, c, basis = gen.gen_primal_columns(m, n, seed, 0, n)
    return c, A, b, basis


GENERATORS = {"gen_primal": gen.gen_primal, "gen_dual": gen.gen_dual, "gen_box": gen.gen_box,
              "gen_primal_degenerate": gen.gen_primal_degenerate, "gen_primal_columns": gen_columns}

# north_star: objective and solution within 1e-9 relative in fp64.  Truly relative: |got - want| <= RTOL |want| + ATOL,
# with an absolute floor of 1e-12 only for entries that are (nearly) zero -- most x entries and several costs are < 1,
# so a max(1, |want|) scale would hold them to 1e-9 ABSOLUTE.
RTOL = 1e-9
ATOL = 1e-12


def load(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        return []
    with open(path) as f:
        return json.load(f)["cases"]


def case_id(cs):
    if "problem" in cs:
        return f"{cs['problem']}-{cs['solver']}"
    return f"{cs['gen']}-{cs['solver']}-m{cs['m']}-n{cs['n']}-s{cs['seed']}-k{cs.get('maxiters', 0)}"


def problem_of(cs):
    """(c, A, b, basis) of a base-solver golden case."""
    if "gen" in cs:
        if cs["gen"] == "gen_dual_unbounded":
            c, A, b, basis = gen.gen_primal(cs["m"], cs["n"], cs["seed"])
            b[3] = -1.5
            return np.abs(c), A, b, basis
        if cs["gen"] == "gen_primal_negG":
            c, A, b, basis = gen.gen_primal(cs["m"], cs["n"], cs["seed"])
            A[:, : cs["n"] - cs["m"]] *= -1
            return c, A, b, basis
        return GENERATORS[cs["gen"]](cs["m"], cs["n"], cs["seed"])
    return np.array(cs["c"]), np.array(cs["A"]), np.array(cs["b"]), np.array(cs["basis"])


def ub_of(cs):
    return np.array([np.inf if v == "inf" else v for v in cs["ub"]], dtype=float)


def assert_close(got, want, what="", rtol=RTOL, atol=ATOL):
    """rtol / atol are widened only by the tests of the float32-state (naive) solvers, whose reference values are
    themselves float32 results of a LAPACK inverse (each such test says so)."""
    got = np.asarray(got, dtype=float)
    want = np.asarray(want, dtype=float)
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    if not got.size:
        return
    excess = np.abs(got - want) - (rtol * np.abs(want) + atol)
    k = int(np.argmax(excess))
    assert excess.flat[k] <= 0.0, (f"{what}: entry {k}: got {got.flat[k]!r}, want {want.flat[k]!r} "
                                   f"(|diff| {abs(got.flat[k] - want.flat[k]):.3e} > {rtol} |want| + {atol})")
