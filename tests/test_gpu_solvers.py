"""GPU parity of the solver classes (drop-in boundary) against the reference's golden vectors and the oracle.

Reads like the reference's own tests (tests/test_*_solvers.py there): same constructors, ``solve()``, result
fields.  Bar: identical pivot triples / basis / iters / optimum / exception; x and cost within 1e-9 relative.
"""
import numpy as np
import pytest

from tests._util import assert_close, case_id, gen, load, problem_of, ub_of

pytestmark = pytest.mark.gpu

BASE = [cs for cs in load("random.json") + load("textbook.json") + load("random_large.json") + load("random_xlarge.json") + load("degenerate.json")
        if cs["solver"] in ("PrimalRevisedSimplexSolver", "DualRevisedSimplexSolver", "PrimalNaiveSimplexSolver",
                            "DualNaiveSimplexSolver")]


def logged(cls, cap=8192):
    return type(cls.__name__ + "Logged", (cls,), {"_LOG_CAP": cap})


def solver_class(name):
    from linprog_amd import dual_solvers, primal_solvers, special_solvers

    for mod in (primal_solvers, dual_solvers, special_solvers):
        if hasattr(mod, name):
            return getattr(mod, name)
    raise KeyError(name)


@pytest.mark.parametrize("cs", BASE, ids=case_id)
def test_base_solvers_match_reference(cs):
    from linprog_amd import exceptions

    ref = cs["ref"]
    cls = logged(solver_class(cs["solver"]), cap=max(16, len(ref["pivots"]) + 1))
    c, A, b, basis = problem_of(cs)
    if ref["exception"] in ("BasisIsPrimalInfeasibleError", "BasisIsDualInfeasibleError"):
        with pytest.raises(getattr(exceptions, ref["exception"])):
            cls(c, A, b, basis)
        return
    s = cls(c, A, b, basis)
    if ref["exception"]:
        with pytest.raises(getattr(exceptions, ref["exception"])):
            s.solve(cs.get("maxiters", 100))
        assert s.pivot_log().tolist() == ref["pivots"]
        return
    res = s.solve(cs.get("maxiters", 100))
    assert s.pivot_log().tolist() == ref["pivots"], "pivot sequence differs from the reference"
    assert res.iters == ref["iters"]
    assert res.optimum == ref["optimum"]
    assert res.basis.tolist() == ref["basis"]
    assert res.basis is s.basis  # result aliases solver state (primal_solvers.py:83-86)
    naive = "Naive" in cs["solver"]  # float32 state re-inverted every pivot: values agree to float32 accuracy
    if "x" in ref:
        if naive:
            assert np.allclose(res.x, ref["x"], rtol=2e-4, atol=2e-4)
        else:
            assert_close(res.x, ref["x"], "x")
    if naive:
        assert abs(res.cost - ref["cost"]) <= 2e-4 * max(1.0, abs(ref["cost"]))
    else:
        assert_close(res.cost, ref["cost"], "cost")
    if "optimal_basis" in cs:  # the reference's own assertion (tests/test_primal_base_solvers.py:19-21)
        assert np.linalg.norm(res.x[res.basis] - np.array(cs["optimal_bfs"]), 2) < 1e-2
        assert np.array_equal(res.basis, cs["optimal_basis"])


SMALL = [cs for cs in BASE if "Naive" not in cs["solver"] and cs.get("m", 3) <= 64 and not cs["ref"]["exception"]]


@pytest.mark.parametrize("cs", SMALL[::3], ids=case_id)
def test_multi_kernel_chain_on_small_problems(cs, monkeypatch):
    """Small LPs run as one launch by default (lpx_revised_solve_small); the multi-kernel chain that the large problems
    use must take the same pivots on them (LPX_SMALL_SOLVE=0)."""
    import linprog_amd.engine as engine

    monkeypatch.setattr(engine.RevisedEngine, "SMALL_SOLVE", False)
    ref = cs["ref"]
    s = logged(solver_class(cs["solver"]), cap=max(16, len(ref["pivots"]) + 1))(*problem_of(cs))
    assert not s._engine.fits_small(s._device_kind())
    res = s.solve(cs.get("maxiters", 100))
    assert s.pivot_log().tolist() == ref["pivots"] and res.iters == ref["iters"] and res.basis.tolist() == ref["basis"]
    assert_close(res.cost, ref["cost"], "cost")


@pytest.mark.parametrize("kind,m,n,seed", [("primal", 64, 128, 3), ("primal", 37, 100, 4), ("primal", 128, 256, 5),
                                           ("dual", 64, 128, 6), ("dual", 100, 210, 7), ("bounded", 64, 128, 2),
                                           ("bounded", 128, 256, 3), ("bounded", 33, 67, 4)])
def test_one_launch_small_solve_bit_identical_to_oracle(kind, m, n, seed, monkeypatch):
    """lpx_revised_solve_small: the whole solve() call in one launch, B^-1 in LDS.  Dot products run in the oracle's
    order, so the pivot log, B^-1, x_B, basis and the per-variable states are bit-identical to the oracle's loops, also
    when the solve is cut into re-entrant pieces (state written back and reloaded between calls)."""
    from linprog_amd.dual_solvers import DualRevisedSimplexSolver
    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver
    from linprog_amd.special_solvers import BoundedVariablePrimalSimplexSolver
    from oracle import oracle as O
    import linprog_amd.engine as engine

    monkeypatch.setattr(engine.RevisedEngine, "SMALL_MAX_ROWS", {0: 4096, 1: 4096, 2: 4096})  # every shape that fits LDS
    monkeypatch.setattr(engine.RevisedEngine, "BOUNDED_NEEDS_A_IN_LDS", False)  # 128 x 256: the bounded loops with A in HBM
    if kind == "bounded":
        args = gen.gen_box(m, n, seed)
        s = logged(BoundedVariablePrimalSimplexSolver)(*args)
        o = O.OracleBounded(*args)
    else:
        args = (gen.gen_primal if kind == "primal" else gen.gen_dual)(m, n, seed)
        s = logged(PrimalRevisedSimplexSolver if kind == "primal" else DualRevisedSimplexSolver)(*args)
        o = O.OracleRevised(*args, kind=kind)
    assert s._engine.fits_small(s._device_kind())
    for k in (1, 2, 7, 100000):
        res = s.solve(k)
        want = o.solve(k)
        eng = s._engine
        assert s.pivot_log().tolist() == want.log.tolist()
        assert res.iters == want.iters and res.basis.tolist() == want.basis.tolist()
        assert np.array_equal(eng.host_Binv(), o.Binv) and np.array_equal(eng.host_xB(), o.xB)
        assert np.array_equal(res.x, want.x)
        if kind == "bounded":
            assert np.array_equal(eng.host_var_state(), want.var_state)
    assert res.optimum


@pytest.mark.parametrize("name", ["PrimalRevisedSimplexSolver", "PrimalNaiveSimplexSolver"])
def test_blands_sequence_reentrant(name):
    """tests/test_primal_base_solvers.py:24-34 of the reference."""
    cs = [c for c in load("textbook.json") if c["problem"] == "primal_solver_blands_sequence_problem1"
          and c["solver"] == name][0]
    c, A, b, basis = problem_of(cs)
    s = solver_class(name)(c, A, b, basis)
    path = []
    for want in cs["basis_seq"][1:]:
        res = s.solve(maxiters=1)
        path.append(np.array_equal(want, res.basis) and res.iters == 1 and res.optimum is False)
    res = s.solve(maxiters=1)
    path.append(np.array_equal(cs["basis_seq"][-1], res.basis) and res.optimum is True)
    assert all(path)


def test_attributes_and_external_pivot():
    """What PhaseOneSimplexSolver reads from a solver (special_solvers.py:69-80)."""
    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver
    from oracle import oracle as O

    c, A, b, basis = gen.gen_primal(16, 32, 5)
    s = PrimalRevisedSimplexSolver(c, A, b, basis)
    o = O.OracleRevised(c, A, b, basis)
    assert s.inv_basis_matrix.dtype == np.float32 and s.bfs.dtype == np.float32  # Appendix B, before any pivot
    s.pivot(3, 5)
    o.pivot(3, 5)
    assert s.basis[3] == 5
    assert s.inv_basis_matrix.dtype == np.float64
    assert_close(s.inv_basis_matrix, o.Binv, "B^-1 after pivot()")
    assert_close(s.bfs, o.xB, "bfs after pivot()")
    res = s._get_solver_return_object()
    assert_close(res.x, O.expand_and_cost(o.xB, o.basis, o.c, 32)[0], "x")
    # pivot() after a finished solve() (PhaseOne's drive-out does this, special_solvers.py:77-80)
    s.solve(5)
    o.solve(5)
    s.pivot(2, int(np.setdiff1d(np.arange(32), s.basis)[0]))
    o.pivot(2, int(np.setdiff1d(np.arange(32), o.basis)[0]))
    assert s.basis.tolist() == o.basis.tolist()
    assert_close(s.bfs, o.xB, "bfs after solve()+pivot()")


def test_naive_solver_state_is_float32_and_reinverted():
    """PrimalNaiveSimplexSolver keeps a float32 inverse that equals inv(A[:, basis]) after every pivot."""
    from linprog_amd.primal_solvers import PrimalNaiveSimplexSolver

    c, A, b, basis = gen.gen_primal(24, 48, 3)
    s = PrimalNaiveSimplexSolver(c, A, b, basis)
    for _ in range(6):
        s.solve(maxiters=2)
        Binv = s.inv_basis_matrix
        assert Binv.dtype == np.float32 and s.bfs.dtype == np.float32
        B = s.A[:, s.basis].astype(np.float64)
        assert np.abs(Binv.astype(np.float64) @ B - np.eye(24)).max() < 1e-4
        assert np.allclose(s.bfs, np.linalg.solve(B, s.b.astype(np.float64)), rtol=1e-4, atol=1e-4)
    s.pivot(0, int(np.setdiff1d(np.arange(48), s.basis)[0]))
    B = s.A[:, s.basis].astype(np.float64)
    assert np.abs(s.inv_basis_matrix.astype(np.float64) @ B - np.eye(24)).max() < 1e-3


@pytest.mark.parametrize("cs", [c for c in load("textbook.json") if c["solver"] == "BoundedVariablePrimalSimplexSolver"]
                         + load("bounded.json"), ids=case_id)
def test_bounded_matches_reference(cs):
    from linprog_amd.special_solvers import BoundedVariablePrimalSimplexSolver

    ref = cs["ref"]
    if "gen" in cs:
        c, A, b, lb, ub, basis, lbn, ubn = gen.gen_box(cs["m"], cs["n"], cs["seed"])
    else:
        c, A, b, lb, ub = np.array(cs["c"]), np.array(cs["A"]), np.array(cs["b"]), np.array(cs["lb"]), ub_of(cs)
        basis, lbn, ubn = cs["basis"], cs["lb_nonbasic"], cs["ub_nonbasic"]
    s = logged(BoundedVariablePrimalSimplexSolver)(c, A, b, lb, ub, basis, lbn, ubn)
    res = s.solve(100000)
    assert res.optimum is True
    assert res.iters == ref["iters_total"]
    assert s.pivot_log().tolist() == [e[:3] for e in ref["steps"]]
    assert res.basis.tolist() == ref["basis"]
    assert s.ub_nonbasic_vars.tolist() == ref["ub_nonbasic_final"]
    assert s.lb_nonbasic_vars.tolist() == ref["lb_nonbasic_final"]
    assert_close(res.x, ref["x"], "x")
    assert_close(res.cost, ref["cost"], "cost")
    with pytest.raises(NotImplementedError):
        s.pivot(0, 0)


@pytest.mark.parametrize("cs", [c for c in load("textbook.json") if c["solver"] == "PrimalDualAlgorithm"]
                         + load("primal_dual.json"), ids=case_id)
def test_primal_dual_matches_reference(cs):
    from linprog_amd.special_solvers import PrimalDualAlgorithm

    ref = cs["ref"]
    if "gen" in cs:
        c, A, b, _ = gen.gen_dual(cs["m"], cs["n"], cs["seed"])
    else:
        c, A, b = np.array(cs["c"]), np.array(cs["A"]), np.array(cs["b"])
    s = logged(PrimalDualAlgorithm, cap=4096)(c, A, b)
    res = s.solve(100000, 100000)
    assert res.iters == ref["iters"] and res.optimum == ref["optimum"]
    n = s.n
    flat = []  # map [A | I] numbering to the reference's compacted restricted-primal numbering
    for adm, lg in zip(s.admissible_sets, s.inner_pivot_logs):
        pos = {int(col): k for k, col in enumerate(adm)}
        k = len(adm)
        for row, ent, lv in lg.tolist():
            flat.append([row, pos[ent] if ent < n else k + ent - n, pos[lv] if lv < n else k + lv - n])
    assert flat == ref["pivots"]
    assert res.basis.tolist() == ref["basis"]
    assert_close(res.x, ref["x"], "x")
    assert_close(res.cost, ref["cost"], "cost")
    if "yA_last" in ref:  # the dual vector itself (special_solvers.py:430-444; the reference's fresh inverse is float64)
        assert [len(a) for a in s.admissible_sets] == ref["adm_sizes"]
        assert_close(s.y[: s.m] @ np.asarray(s.A, dtype=np.float64), ref["yA_last"], "y @ A")
    if "optimal_bfs" in cs:  # tests/test_primal_dual_algorithm.py:11-16: exact
        assert np.array_equal(res.x, np.array(cs["optimal_bfs"], dtype=float))


@pytest.mark.parametrize("cs", [c for c in load("textbook.json") if c["solver"] == "SimplexSolver"] + load("diet.json"),
                         ids=case_id)
def test_two_phase_driver(cs):
    from linprog_amd import exceptions
    from linprog_amd.simplex import SimplexSolver

    ref = cs["ref"]
    if cs["problem"] == "diet":
        a = {k: np.array(v, dtype=float) for k, v in cs["arrays"].items()}
        solver = SimplexSolver(c=a["c"], G=a["G"], h=a["h"], lb=a["lb"], ub=a["ub"])
    else:
        solver = SimplexSolver(np.array(cs["c"]), np.array(cs["A"]), np.array(cs["b"]))
    if ref["exception"]:
        with pytest.raises(getattr(exceptions, ref["exception"])):
            solver.solve()
        return
    res = solver.solve()
    assert res.iters == ref["iters"] and res.optimum == ref["optimum"] and res.basis is None
    assert_close(res.x, ref["x"], "x")
    assert_close(res.cost, ref["cost"], "cost")


def test_tableau_row_entry_point_and_artificial_drive_out():
    """lpx_revised_tableau_row = (inv_basis_matrix @ A)[row, :] (special_solvers.py:69-71) without downloading B^-1, and
    PhaseOneSimplexSolver's drive-out built on it: on LPs with a duplicated equation an artificial variable stays basic
    at zero level after phase one; it is pivoted out where the tableau row has a positive entry, else its (redundant) row
    is dropped -- either way the basis handed to phase two is a feasible basis of the rows that are kept."""
    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver
    from linprog_amd.special_solvers import PhaseOneSimplexSolver

    c, A, b, basis = gen.gen_primal(40, 90, 21)
    s = PrimalRevisedSimplexSolver(c, A, b, basis)
    s.solve(25)
    T = s.inv_basis_matrix @ s.A.astype(np.float64)
    for row in (0, 17, 39):
        got = s._engine.tableau_row(row)
        assert np.allclose(got, T[row], rtol=1e-12, atol=1e-12)
    # redundant equations with small integer data (every pivot exact): bases, kept rows and right-hand sides are the
    # reference's PhaseOneSimplexSolver on the same inputs (tests/golden/phase_one.json, make_golden.py::phase_one_cases)
    cases = load("phase_one.json")
    assert len(cases) >= 6
    for cs in cases:
        A, b, ref = np.array(cs["A"], dtype=float), np.array(cs["b"], dtype=float), cs["ref"]
        p1 = PhaseOneSimplexSolver(np.zeros(A.shape[1]), A, b)
        p1.solve(1000)
        assert p1.basis.tolist() == ref["basis"] and int(p1.m) == ref["m"]
        assert np.array_equal(np.asarray(p1.A, dtype=float), np.array(ref["A"])) and np.array_equal(np.asarray(p1.b, dtype=float), np.array(ref["b"]))


def test_standard_form_on_device_equals_the_host_helpers():
    """lpx_build_standard_form (slack columns, bound rows, sign normalisation, phase-one artificial block in one kernel)
    against the host restatements of preprocessing.py:6-69 applied in the reference's order, entry for entry."""
    import json
    import os

    from linprog_amd.preprocessing import ProblemPreprocessingUtils as P
    from tests._util import ROOT

    rng = np.random.default_rng(3)
    cases = []
    cs = json.load(open(os.path.join(ROOT, "tests", "golden", "diet.json")))["cases"][0]
    a = {k: np.array(v, dtype=float) for k, v in cs["arrays"].items()}
    cases.append((a["c"], None, None, a["G"], a["h"], a["lb"], a["ub"]))
    for me, mi, nx in ((3, 0, 7), (0, 5, 4), (4, 6, 9)):
        c = rng.standard_normal(nx)
        A = rng.standard_normal((me, nx)) if me else None
        b = rng.standard_normal(me) if me else None  # negative entries: rows get flipped
        G = rng.standard_normal((mi, nx)) if mi else None
        h = rng.standard_normal(mi) if mi else None
        lb = np.where(rng.random(nx) < 0.4, rng.standard_normal(nx), 0.0)
        ub = np.where(rng.random(nx) < 0.4, 3.0 + rng.random(nx), np.inf)
        cases.append((c, A, b, G, h, lb, ub))
    for c, A, b, G, h, lb, ub in cases:
        mi = 0 if G is None else G.shape[0]
        # host: the reference's sequence (simplex.py:35-77, then :83-85)
        if G is not None and A is None:
            c0, A0, b0 = P.canonical_form_to_standard_form(c, G, h)
        elif G is None:
            c0, A0, b0 = P.preprocess_problem(c, A, b)
        else:
            _, G_, h_ = P.canonical_form_to_standard_form(c, G, h)
            c0, A0, b0 = P.preprocess_problem(np.concatenate([c, np.zeros(mi)]),
                                              np.vstack([np.hstack([A, np.zeros((A.shape[0], mi))]), G_]), np.concatenate([b, h_]))
        lbx, ubx = np.concatenate([lb, np.zeros(mi)]), np.concatenate([ub, np.full(mi, np.inf)])
        c1, A1, b1 = P.add_variables_bounds_to_coefficient_matrix(c0, A0, b0, lbx, ubx)
        dev0 = P.standard_form_on_device(c, A, b, G, h)
        assert np.array_equal(dev0.host(), A0) and np.array_equal(dev0.b32, b0) and np.array_equal(dev0.c32, c0)
        dev1 = P.standard_form_on_device(c, A, b, G, h, lb, ub, artificials=True)
        m1, n1 = A1.shape
        assert dev1.core.shape == (m1, n1) and dev1.shape == (m1, n1 + m1) and dev1.artificial_from == n1
        assert np.array_equal(dev1.core.host(), A1) and np.array_equal(dev1.b32, b1) and np.array_equal(dev1.core.c32, c1)
        full = dev1.host()
        assert np.array_equal(full[:, n1:], np.eye(m1, dtype=np.float32))
        assert np.array_equal(dev1.c32, np.concatenate([np.zeros(n1), np.ones(m1)]).astype(np.float32))
        assert np.array_equal(dev1.columns([0, n1 - 1, n1]), full[:, [0, n1 - 1, n1]])


@pytest.mark.parametrize("cs", [c for c in load("textbook.json") if c["solver"] == "PrimalDualAlgorithm"]
                         + load("primal_dual.json") + [c for c in load("random.json") if c["solver"] == "PrimalDualAlgorithm"
                                                       and c["ref"]["exception"] != "IndexError"],  # the reference's own crash
                         ids=case_id)
def test_primal_dual_one_launch_matches_reference(cs):
    """PrimalDualAlgorithm without pivot logging runs as ONE launch (lpx_pd_solve_small): outer iteration count, x, basis,
    cost and the raised exception equal the reference's goldens, and the total of inner pivots equals the golden trace's."""
    from linprog_amd import exceptions
    from linprog_amd.special_solvers import PrimalDualAlgorithm

    ref = cs["ref"]
    if "gen" in cs:
        c, A, b, _ = problem_of(cs)
    else:
        c, A, b = np.array(cs["c"]), np.array(cs["A"]), np.array(cs["b"])
    s = PrimalDualAlgorithm(c, A, b)
    assert s._LOG_CAP == 0
    if ref["exception"]:
        with pytest.raises(getattr(exceptions, ref["exception"])):
            s.solve(100000, 100000) if "gen" in cs else s.solve()
        return
    res = s.solve(100000, 100000) if "gen" in cs else s.solve()
    assert res.iters == ref["iters"] and res.optimum == ref["optimum"]
    assert s.inner_pivots == len(ref["pivots"])
    assert res.basis.tolist() == ref["basis"]
    assert_close(res.x, ref["x"], "x")
    assert_close(res.cost, ref["cost"], "cost")
    if "yA_last" in ref:  # the one-launch kernel's dual vector against the reference's last admissibility test
        assert_close(s.y[: s.m] @ np.asarray(s.A, dtype=np.float64), ref["yA_last"], "y @ A")
    if "optimal_bfs" in cs:  # tests/test_primal_dual_algorithm.py:11-16 asserts exact equality
        assert np.array_equal(res.x, np.array(cs["optimal_bfs"], dtype=float))


def test_primal_dual_is_a_dual_revised_solver_like_the_reference():
    """special_solvers.py:346: class PrimalDualAlgorithm(DualRevisedSimplexSolver)."""
    from linprog_amd.dual_solvers import DualRevisedSimplexSolver
    from linprog_amd.special_solvers import PrimalDualAlgorithm

    assert issubclass(PrimalDualAlgorithm, DualRevisedSimplexSolver)


def test_full_size_config2_properties():
    """BASELINE config 2 (m=1000, n=2000): size-independent properties of a pivot window, plus the oracle's
    pivot sequence on the same window."""
    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver
    from oracle import oracle as O

    c, A, b, basis = gen.gen_primal(1000, 2000, 0)
    K = 150
    s = logged(PrimalRevisedSimplexSolver, cap=K)(c, A, b, basis)
    res = s.solve(K)
    o = O.OracleRevised(c, A, b, basis, log_cap=K).solve(K)
    assert s.pivot_log().tolist() == o.log.tolist()
    assert res.iters == K and res.optimum is False
    A32 = s.A.astype(np.float64)
    # B^-1 B = I, A x = b, x >= 0, cost = c.x, basis entries distinct
    Binv = s.inv_basis_matrix
    assert np.abs(Binv @ A32[:, res.basis] - np.eye(1000)).max() < 1e-9
    assert np.abs(A32 @ res.x - s.b).max() < 1e-9
    assert res.x.min() >= -1e-12 and len(set(res.basis.tolist())) == 1000
    assert_close(res.cost, o.cost, "cost")
    assert_close(res.x, o.x, "x")


def test_primal_dual_on_infeasible_primal_does_not_crash():
    """The reference cannot raise DualIsUnboundedError while an artificial is basic (its y_rp component is 1 > 0,
    special_solvers.py:435) and then dies with IndexError at :449-452 (golden: gen_dual_unbounded).  The device path
    follows the same arithmetic (theta = inf) and reports a non-optimal result instead of crashing."""
    from linprog_amd.special_solvers import PrimalDualAlgorithm

    cs = [c for c in load("random.json") if c["gen"] == "gen_dual_unbounded" and c["solver"] == "PrimalDualAlgorithm"][0]
    assert cs["ref"]["exception"] == "IndexError"
    c, A, b, _ = problem_of(cs)
    res = PrimalDualAlgorithm(c, A, b).solve(12, 100)
    assert res.optimum is False and res.iters == 12


@pytest.mark.parametrize("m,n,seed", [(64, 128, 3), (300, 650, 4), (1000, 2000, 5)])
def test_refine_inverse_restores_a_drifted_inverse(m, n, seed):
    """Opt-in Newton-Schulz refinement (lpx_revised_refine): after some pivots B^-1 is perturbed by 1e-7 relative noise;
    one step brings max|I - B B^-1| down quadratically, a second to fp64 rounding; x_B is recomputed; the solve then
    finishes on the oracle's optimum (cost within 1e-9)."""
    import torch
    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver
    from oracle import oracle as O

    c, A, b, basis = gen.gen_primal(m, n, seed)
    s = PrimalRevisedSimplexSolver(c, A, b, basis)
    s.solve(min(200, 3 * m))
    eng = s._engine
    g = torch.Generator(device="cuda").manual_seed(seed)
    noise = 1.0 + 1e-7 * torch.randn(eng.Binv.shape, generator=g, device="cuda", dtype=torch.float64)
    eng.Binv.mul_(noise)
    before, mid = s.refine_inverse(steps=1)
    assert 1e-9 < before < 1e-3 and mid < 50 * before * before + 1e-12
    again, after = s.refine_inverse(steps=1)
    assert abs(again - mid) <= 1e-14 + 1e-6 * mid and after < 1e-11
    # the refined state is the exact inverse of the current basis matrix
    A32 = np.asarray(A, dtype=np.float32).astype(np.float64)
    Bm = A32[:, eng.host_basis()]
    assert np.max(np.abs(Bm @ eng.host_Binv() - np.eye(m))) < 1e-10
    xB = np.linalg.solve(Bm, np.asarray(b, dtype=np.float32).astype(np.float64))
    assert np.max(np.abs(eng.host_xB() - xB)) <= 1e-9 * max(1.0, np.max(np.abs(xB)))
    if m <= 300:
        res = s.solve(100000)
        want = O.OracleRevised(c, A, b, basis, log_cap=4).solve(100000)
        assert res.optimum and want.status == O.OPTIMAL
        assert abs(res.cost - want.cost) <= 1e-9 * max(1.0, abs(want.cost))
    else:  # a full Bland solve at this size is minutes of oracle time: pivot on and check the invariants instead
        res = s.solve(300)
        Bm = A32[:, eng.host_basis()]
        assert np.max(np.abs(Bm @ eng.host_Binv() - np.eye(m))) < 1e-9
        assert np.max(np.abs(A32 @ res.x - np.asarray(b, dtype=np.float32))) < 1e-8


@pytest.mark.parametrize("m,n,seed", [(8, 16, 0), (16, 32, 1), (32, 64, 2), (64, 128, 3)])
def test_primal_dual_warm_start_same_optimum_fewer_pivots(m, n, seed):
    """Opt-in warm start of the restricted primal (no counterpart in the reference): same optimal cost as the default
    cold-restart mode and as HiGHS, x feasible, and no more inner pivots than the cold mode."""
    from scipy.optimize import linprog
    from linprog_amd.special_solvers import PrimalDualAlgorithm

    c, A, b, _ = gen.gen_dual(m, n, seed)
    cold = PrimalDualAlgorithm(c, A, b)
    rc = cold.solve(100000, 100000)
    warm = PrimalDualAlgorithm(c, A, b)
    rw = warm.solve(100000, 100000, warm_start=True)
    assert rc.optimum and rw.optimum
    assert abs(rw.cost - rc.cost) <= 1e-9 * max(1.0, abs(rc.cost))
    c32, A32, b32 = (np.asarray(v, dtype=np.float32).astype(float) for v in (c, A, b))
    flip = b32 < 0
    A32[flip] *= -1
    b32[flip] *= -1
    assert np.max(np.abs(A32 @ rw.x - b32)) < 1e-8 and rw.x.min() > -1e-10
    ref = linprog(c32, A_eq=A32, b_eq=b32, bounds=(0, None), method="highs")
    assert abs(rw.cost - ref.fun) <= 1e-8 * max(1.0, abs(ref.fun))
    assert 0 < warm.inner_pivots <= cold.inner_pivots
    print(f"inner pivots cold {cold.inner_pivots} warm {warm.inner_pivots}, outer {rc.iters} / {rw.iters}")


@pytest.mark.parametrize("m,n,seed", [(5, 12, 1), (64, 128, 3), (300, 650, 4), (1000, 2000, 5)])
def test_reinvert_rebuilds_the_inverse_from_the_basis(m, n, seed):
    """Opt-in reinversion (lpx_revised_reinvert_f64): after some pivots B^-1 is overwritten with garbage;
    reinvert() rebuilds it from A[:, basis] alone (np.linalg.inv's job at primal_solvers.py:92-94), x_B follows, and the
    solve continues to the oracle's optimum."""
    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver
    from oracle import oracle as O

    c, A, b, basis = gen.gen_primal(m, n, seed)
    s = PrimalRevisedSimplexSolver(c, A, b, basis)
    s.solve(min(200, 3 * m))
    eng = s._engine
    eng.Binv.fill_(float("nan"))
    eng.xB.fill_(-1.0)
    s.reinvert()
    A32 = np.asarray(A, dtype=np.float32).astype(np.float64)
    Bm = A32[:, eng.host_basis()]
    assert np.max(np.abs(Bm @ eng.host_Binv() - np.eye(m))) < 1e-10
    assert np.max(np.abs(eng.host_Binv() - np.linalg.inv(Bm))) <= 1e-9 * np.max(np.abs(np.linalg.inv(Bm)))
    xB = np.linalg.solve(Bm, np.asarray(b, dtype=np.float32).astype(np.float64))
    assert np.max(np.abs(eng.host_xB() - xB)) <= 1e-9 * max(1.0, np.max(np.abs(xB)))
    if m <= 300:
        res = s.solve(100000)
        want = O.OracleRevised(c, A, b, basis, log_cap=4).solve(100000)
        assert res.optimum and want.status == O.OPTIMAL
        assert abs(res.cost - want.cost) <= 1e-9 * max(1.0, abs(want.cost))
    else:
        res = s.solve(300)
        Bm = A32[:, eng.host_basis()]
        assert np.max(np.abs(Bm @ eng.host_Binv() - np.eye(m))) < 1e-9
        assert np.max(np.abs(A32 @ res.x - np.asarray(b, dtype=np.float32))) < 1e-8


def test_reinvert_of_a_singular_basis_raises_like_numpy():
    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver

    c, A, b, basis = gen.gen_primal(20, 40, 2)
    s = PrimalRevisedSimplexSolver(c, A, b, basis)
    s._engine.basis[1] = s._engine.basis[0]  # the same column twice
    with pytest.raises(np.linalg.LinAlgError):
        s.reinvert()


def test_reinvert_every_keeps_a_long_run_exact():
    """Opt-in periodic reinversion inside solve() (engine.reinvert_every), same long Bland run as below."""
    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver

    cs = [c for c in load("random_large.json") if c["m"] == 256][0]
    c, A, b, basis = problem_of(cs)
    s = PrimalRevisedSimplexSolver(c, A, b, basis)
    s._engine.reinvert_every = 500
    res = s.solve(100000)
    assert res.optimum
    assert abs(res.cost - cs["ref"]["cost"]) <= 1e-9 * max(1.0, abs(cs["ref"]["cost"]))
    assert sorted(res.basis.tolist()) == sorted(cs["ref"]["basis"])
    eng = s._engine
    Bm = np.asarray(A, dtype=np.float32).astype(np.float64)[:, eng.host_basis()]
    assert np.max(np.abs(Bm @ eng.host_Binv() - np.eye(256))) < 1e-10


@pytest.mark.parametrize("m,n,seed", [(40, 600, 0), (64, 2400, 1), (120, 5000, 2)])
def test_staged_pricing_takes_the_reference_decisions(m, n, seed, monkeypatch):
    """Staged pricing of the primal revised loop (large A: first eighth of the columns, the rest only when that prefix
    cannot settle primal_solvers.py:135-142).  Forced on for small problems here: the full solve -- early iterations
    resolve in the prefix, the last ones need the whole row -- must give the oracle's pivot sequence, and the same
    B^-1 / x / cost bit for bit as the unstaged chain."""
    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver
    from oracle import oracle as O

    c, A, b, basis = gen.gen_primal(m, n, seed)
    want = O.OracleRevised(c, A, b, basis, log_cap=20000).solve(100000)
    assert want.status == O.OPTIMAL
    cls = logged(PrimalRevisedSimplexSolver, cap=20000)
    monkeypatch.setenv("LPX_STAGED_MIN_MB", "0")
    s1 = cls(c, A, b, basis)
    r1 = s1.solve(100000)
    assert s1.pivot_log().tolist() == want.log.tolist()
    assert r1.optimum and r1.iters == want.iters
    assert_close(r1.x, want.x, "x")
    monkeypatch.setenv("LPX_STAGED_PRICING", "0")
    s0 = cls(c, A, b, basis)
    r0 = s0.solve(100000)
    assert s0.pivot_log().tolist() == s1.pivot_log().tolist()
    assert np.array_equal(r0.x, r1.x) and r0.cost == r1.cost
    assert np.array_equal(s0.inv_basis_matrix, s1.inv_basis_matrix)


def test_refine_every_keeps_a_long_run_exact():
    """Opt-in periodic refinement inside solve() (engine.refine_every): the full Bland solve of gen_primal(256, 512, 0)
    (3470 pivots in the reference) still ends on the reference's optimum, and the final inverse is exact to 1e-12."""
    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver

    cs = [c for c in load("random_large.json") if c["m"] == 256][0]
    c, A, b, basis = problem_of(cs)
    s = PrimalRevisedSimplexSolver(c, A, b, basis)
    s._engine.refine_every = 400
    res = s.solve(100000)
    assert res.optimum
    assert abs(res.cost - cs["ref"]["cost"]) <= 1e-9 * max(1.0, abs(cs["ref"]["cost"]))
    assert sorted(res.basis.tolist()) == sorted(cs["ref"]["basis"])
    eng = s._engine
    Bm = np.asarray(A, dtype=np.float32).astype(np.float64)[:, eng.host_basis()]
    assert np.max(np.abs(Bm @ eng.host_Binv() - np.eye(256))) < 1e-11


@pytest.mark.parametrize("m,seed", [(8, 0), (16, 1), (16, 5), (64, 2), (100, 6), (128, 3)])
def test_bounded_variable_dual_simplex_opt_in(m, seed):
    """BoundedVariableDualSimplexSolver (BASELINE config 5's wording; the reference has the bounded PRIMAL solver and a dual
    solver for x >= 0 only, so this opt-in class is PARITY UNPINNED against it): one launch per solve() call.  Bit for
    bit against the oracle's restatement of the same rules (lpo_dual_bounded_solve: pivot log, B^-1, x_B, per-variable
    states, also when the solve is cut into re-entrant pieces), bounds and A x = b hold, and the optimum is HiGHS's."""
    from scipy.optimize import linprog

    from linprog_amd.special_solvers import BoundedVariableDualSimplexSolver
    from oracle import oracle as O

    args = gen.gen_box_dual(m, 2 * m, seed)
    c, A, b, lb, ub = args[:5]
    s = logged(BoundedVariableDualSimplexSolver)(*args)
    o = O.OracleDualBounded(*args)
    for k in (1, 3, 100000):
        res, want = s.solve(k), o.solve(k)
        eng = s._engine
        assert s.pivot_log().tolist() == want.log.tolist() and res.iters == want.iters
        assert res.basis.tolist() == want.basis.tolist()
        assert np.array_equal(eng.host_Binv(), o.Binv) and np.array_equal(eng.host_xB(), o.xB)
        assert np.array_equal(eng.host_var_state(), want.var_state) and np.array_equal(res.x, want.x)
    assert res.optimum is True
    A32, b32, c32 = (np.asarray(v, dtype=np.float32).astype(float) for v in (A, b, c))
    assert np.abs(A32 @ res.x - b32).max() < 1e-9 and (res.x - lb).min() >= -1e-8 and (ub - res.x).min() >= -1e-8
    hs = linprog(c32, A_eq=A32, b_eq=b32, bounds=list(zip(lb, [None if np.isinf(u) else u for u in ub])), method="highs")
    assert hs.status == 0 and abs(res.cost - hs.fun) <= 1e-9 * max(1.0, abs(hs.fun))
    # the bounded PRIMAL solver (the reference's own) reaches the same optimum from its own feasible start
    from linprog_amd.special_solvers import BoundedVariablePrimalSimplexSolver

    ubf = np.where(np.isinf(ub), 1e6, ub)
    k = A.shape[1] - A.shape[0]
    p = BoundedVariablePrimalSimplexSolver(c, A, b, lb, ubf, args[5], np.arange(k), np.array([], dtype=int)).solve(100000)
    assert p.optimum and abs(p.cost - res.cost) <= 1e-7 * max(1.0, abs(res.cost))


def test_bounded_variable_dual_simplex_errors():
    """Dual infeasible start -> BasisIsDualInfeasibleError (dual_solvers.py:14-16 read for two bounds); an infeasible primal
    -> DualIsUnboundedError out of solve() (:41-44); a state beyond one CU's LDS is refused at construction."""
    from linprog_amd import exceptions
    from linprog_amd.special_solvers import BoundedVariableDualSimplexSolver

    c, A, b, lb, ub, basis, lo, up = gen.gen_box_dual(16, 32, 1)
    with pytest.raises(exceptions.BasisIsDualInfeasibleError):
        BoundedVariableDualSimplexSolver(c, A, b, lb, ub, basis, np.arange(16), np.array([], dtype=int))
    b2 = b.copy()
    b2[3] = -5.0  # row 3: G x + s = -5 with G, x, s >= 0 has no solution (preprocessing flips the row: -G x - s = 5)
    s = BoundedVariableDualSimplexSolver(c, A, b2, lb, ub, basis, lo, up)
    with pytest.raises(exceptions.DualIsUnboundedError):
        s.solve(1000)
    args = gen.gen_box_dual(170, 340, 0)
    with pytest.raises(ValueError):
        BoundedVariableDualSimplexSolver(*args)


@pytest.mark.parametrize("kind,m,n,seed", [("primal", 24, 60, 1), ("dual", 64, 128, 2), ("bounded", 40, 90, 3), ("primal", 150, 320, 4)])
def test_arena_state_equals_a_tensor_per_array(kind, m, n, seed, monkeypatch):
    """RevisedEngine keeps a small state in ONE device allocation (one upload; x, cost, basis, control block back in one
    download after a one-launch solve).  Same results, bit for bit, as the state with a tensor per array -- piecewise
    solves, a caller's pivot() in between (the host copy of the results must not outlive it), the attributes other code
    reads."""
    import linprog_amd.engine as engine
    from linprog_amd.dual_solvers import DualRevisedSimplexSolver
    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver
    from linprog_amd.special_solvers import BoundedVariablePrimalSimplexSolver

    cls = {"primal": PrimalRevisedSimplexSolver, "dual": DualRevisedSimplexSolver, "bounded": BoundedVariablePrimalSimplexSolver}[kind]
    args = {"primal": gen.gen_primal, "dual": gen.gen_dual, "bounded": gen.gen_box}[kind](m, n, seed)
    a = logged(cls)(*args)
    monkeypatch.setattr(engine.RevisedEngine, "ARENA_MAX", -1)
    t = logged(cls)(*args)
    assert a._engine.arena is not None and t._engine.arena is None
    for k in (1, 3, 100000):
        ra, rt = a.solve(k), t.solve(k)
        assert ra.iters == rt.iters and ra.optimum == rt.optimum
        assert np.array_equal(ra.x, rt.x) and ra.cost == rt.cost and np.array_equal(ra.basis, rt.basis)
        assert a.pivot_log().tolist() == t.pivot_log().tolist()
        assert np.array_equal(a.bfs, t.bfs) and np.array_equal(a.inv_basis_matrix, t.inv_basis_matrix)
        if kind == "bounded":
            assert np.array_equal(a.lb_nonbasic_vars, t.lb_nonbasic_vars) and np.array_equal(a.ub_nonbasic_vars, t.ub_nonbasic_vars)
        if kind == "primal" and k == 1:
            # a basis change behind solve()'s back: entering = the next pivot the loop itself would take (always legal)
            probe = logged(cls)(*args)
            probe.solve(2)
            row, col, _ = probe.pivot_log().tolist()[1]
            a.pivot(row, col)
            t.pivot(row, col)
            assert np.array_equal(a.basis, t.basis) and a.basis[row] == col
            assert np.array_equal(a.bfs, t.bfs)
            xa, xt = a._get_bfs_expanded(), t._get_bfs_expanded()
            assert np.array_equal(xa, xt) and xa[col] == a.bfs[row]
    assert ra.optimum


def test_bounded_solver_takes_the_one_launch_kernel_only_while_A_fits_in_lds():
    """Routing rule of RevisedEngine.fits_small for the bounded loops (lpx_revised_small_a_in_lds): 80 x 160 keeps A beside
    the state in LDS and runs as one launch, 96 x 192 does not and takes the multi-kernel chain; both take the oracle's
    pivots (the kernel itself is bit-identical either way: test_one_launch_small_solve_bit_identical_to_oracle)."""
    from linprog_amd import _lib
    from linprog_amd.special_solvers import BoundedVariablePrimalSimplexSolver
    from oracle import oracle as O

    lib = _lib.load()
    assert lib.lpx_revised_small_a_in_lds(80, 160, _lib.KIND_BOUNDED) == 1
    assert lib.lpx_revised_small_a_in_lds(96, 192, _lib.KIND_BOUNDED) == 0
    assert lib.lpx_revised_small_a_in_lds(64, 128, _lib.KIND_PRIMAL) == 1 and lib.lpx_revised_small_a_in_lds(128, 256, _lib.KIND_PRIMAL) == 0
    assert lib.lpx_revised_small_a_in_lds(0, 5, _lib.KIND_PRIMAL) == 0
    for m, one_launch in ((80, True), (96, False)):
        args = gen.gen_box(m, 2 * m, 7)
        s = logged(BoundedVariablePrimalSimplexSolver)(*args)
        assert s._engine.fits_small(_lib.KIND_BOUNDED) is one_launch
        res, want = s.solve(100000), O.OracleBounded(*args).solve(100000)
        assert res.optimum and res.iters == want.iters
        assert s.pivot_log().tolist() == want.log.tolist()
        assert_close(res.x, want.x, "x")
        assert_close(res.cost, want.cost, "cost")
