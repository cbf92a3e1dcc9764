"""GPU parity of the tableau-form solver and the batched small-LP kernel."""
import numpy as np
import pytest

from tests._util import assert_close, case_id, gen, load, problem_of

pytestmark = pytest.mark.gpu

PRIMAL = [cs for cs in load("random.json") + load("textbook.json") + load("random_large.json") + load("random_xlarge.json") + load("degenerate.json")
          if cs["solver"] == "PrimalRevisedSimplexSolver" and not cs["ref"]["exception"]]


def logged(cls, cap):
    return type(cls.__name__ + "Logged", (cls,), {"_LOG_CAP": cap})


def pick_chain(chain, monkeypatch):
    """small: one launch per 8 pivots (the default where a workgroup holds a row and a column per thread); one-launch: one
    launch per pivot on two alternating buffers; three-launch: the in-place three-kernel chain of the large configurations"""
    if chain != "small":
        monkeypatch.setenv("LPX_TABLEAU_SMALL_CHAIN", "0")
    if chain == "three-launch":
        monkeypatch.setenv("LPX_TABLEAU_FUSED", "0")


@pytest.mark.parametrize("chain", ["small", "one-launch", "three-launch"])
@pytest.mark.parametrize("condensed", [True, False], ids=["condensed", "full"])
@pytest.mark.parametrize("cs", PRIMAL, ids=case_id)
def test_tableau_matches_reference(cs, condensed, chain, monkeypatch):
    """Both layouts of the carried tableau -- nonbasic columns only (the default) and every column -- and the three launch
    chains of tableaus this size against the reference's golden pivot sequences and results."""
    from linprog_amd.tableau import PrimalTableauSimplexSolver

    pick_chain(chain, monkeypatch)
    ref = cs["ref"]
    c, A, b, basis = problem_of(cs)
    s = logged(PrimalTableauSimplexSolver, max(16, len(ref["pivots"]) + 1))(c, A, b, basis, condensed=condensed)
    res = s.solve(cs.get("maxiters", 100))
    assert s.pivot_log().tolist() == ref["pivots"]
    assert res.iters == ref["iters"] and res.optimum == ref["optimum"] and res.basis.tolist() == ref["basis"]
    if "x" in ref:
        assert_close(res.x, ref["x"], "x")
    assert_close(res.cost, ref["cost"], "cost")


@pytest.mark.parametrize("chain", ["small", "one-launch", "three-launch"])
def test_tableau_bit_exact_vs_oracle(chain, monkeypatch):
    """Elementwise path (no reductions over floating-point sums): the carried tableau must be bit-identical."""
    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    pick_chain(chain, monkeypatch)
    c, A, b, basis = gen.gen_primal(96, 200, 11)
    s = PrimalTableauSimplexSolver(c, A, b, basis, condensed=False)
    sc = PrimalTableauSimplexSolver(c, A, b, basis, condensed=True)
    assert (s._engine.T_alt is not None) and (sc._engine.T_alt is not None)
    o = O.OracleTableau(c, A, b, basis)
    oc = O.OracleTableau(c, A, b, basis, condensed=True)
    for k in (1, 7, 40):  # 1 and 7: the call ends on odd parity, the state must come back to T
        s.solve(k)
        sc.solve(k)
        o.solve(k)
        oc.solve(k)
        assert np.array_equal(sc._engine.T.cpu().numpy()[:, :105], oc.T[:, :105])
        assert np.array_equal(sc._engine.slot_var.cpu().numpy(), oc.slot_var)
        T = s._engine.T.cpu().numpy()
        assert np.array_equal(T[:, :201], o.T[:, :201])
        # condensed form: slot k carries variable slot_var[k]; the slots and the basis partition the variables
        eng = sc._engine
        Tc, slot_var, var_slot = eng.T.cpu().numpy(), eng.slot_var.cpu().numpy(), eng.var_slot.cpu().numpy()
        assert eng.n == 200 - 96 and sorted(slot_var.tolist() + sc.basis.tolist()) == list(range(200))
        assert np.array_equal(var_slot[slot_var], np.arange(eng.n)) and np.all(var_slot[sc.basis] == -1)
        assert np.array_equal(Tc[:, :eng.n], o.T[:, slot_var])
        assert np.array_equal(Tc[:, eng.n], o.T[:, 200])


@pytest.mark.parametrize("rows", ["0", "1", "4"])
@pytest.mark.parametrize("m,n,condensed", [(300, 700, True), (300, 700, False), (530, 1061, True), (64, 1200, True),
                                           (700, 760, True)])
def test_one_launch_chain_bit_exact_multi_workgroup(m, n, condensed, rows, monkeypatch):
    """The single-launch chain with several lookahead workgroups (m + 1 > 256), several column strips and every update
    tile height, in calls of odd and even length (ping-pong parity, hipGraph replays from 32 iterations on): pivot log,
    tableau, basis and column maps bit-identical to the oracle's loop on the same layout."""
    import ctypes as C

    from linprog_amd import _lib
    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    monkeypatch.setenv("LPX_FUSED_ROWS", rows)
    monkeypatch.setenv("LPX_TABLEAU_SMALL_CHAIN", "0")
    c, A, b, basis = gen.gen_primal(m, n, 5)
    s = logged(PrimalTableauSimplexSolver, 512)(c, A, b, basis, condensed=condensed)
    assert s._engine.T_alt is not None and int(_lib.load().lpx_tableau_chain_kind(C.byref(s._engine.s))) == _lib.CHAIN_FUSED
    o = O.OracleTableau(c, A, b, basis, condensed=condensed, log_cap=512)
    done = 0
    for k in (1, 2, 5, 64, 33):
        res = s.solve(k)
        r = o.solve(k)
        done += k
        assert s.pivot_log().tolist() == o.log[:done].tolist()
        w = o.nslots + 1
        T = s._engine.T.cpu().numpy()
        assert np.array_equal(T[:, :w], o.T[:, :w])
        assert res.basis.tolist() == r.basis.tolist() and res.iters == r.iters and np.array_equal(res.x, r.x)
        assert_close(res.cost, r.cost, "cost")  # dot(c, x): the device sums it as a tree, the oracle in index order
        if condensed:
            assert np.array_equal(s._engine.slot_var.cpu().numpy(), o.slot_var)


def test_one_launch_chain_runs_to_the_reference_optimum_and_stays_there(monkeypatch):
    """Full solve through the single-launch chain (graph replays, termination inside a replayed chunk), then a second
    solve() on the optimal state: iters == 1, nothing moves (tests/test_primal_base_solvers.py:24-34's re-entrancy)."""
    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    monkeypatch.setenv("LPX_TABLEAU_SMALL_CHAIN", "0")
    c, A, b, basis = gen.gen_primal(128, 256, 0)
    s = logged(PrimalTableauSimplexSolver, 1024)(c, A, b, basis)
    res = s.solve(100000)
    want = O.OracleTableau(c, A, b, basis, condensed=True, log_cap=1024).solve(100000)
    assert res.optimum and res.iters == want.iters and s.pivot_log().tolist() == want.log.tolist()
    assert np.array_equal(res.x, want.x)
    assert_close(res.cost, want.cost, "cost")
    T0 = s._engine.T.clone()
    again = s.solve(100)
    assert again.optimum and again.iters == 1 and bool((s._engine.T == T0).all())


def test_tableau_unbounded_and_infeasible_start():
    from linprog_amd import exceptions
    from linprog_amd.tableau import PrimalTableauSimplexSolver

    c, A, b, basis = gen.gen_primal(8, 16, 0)
    A[:, :8] *= -1
    with pytest.raises(exceptions.PrimalIsUnboundedError):
        PrimalTableauSimplexSolver(c, A, b, basis).solve()
    c, A, b, basis = gen.gen_dual(8, 16, 0)
    with pytest.raises(exceptions.BasisIsPrimalInfeasibleError):
        PrimalTableauSimplexSolver(c, A, b, basis)


def test_batched_bit_identical_to_oracle_and_reference_sequences():
    """Config 3 shape (m=64, n=128): every LP's pivot log equals the oracle's, x/cost bit-identical; the LPs
    that have a golden also equal the reference's sequence."""
    from linprog_amd.batched import BatchedPrimalSolver
    from oracle import oracle as O

    seeds = list(range(24))
    probs = [gen.gen_primal(64, 128, s) for s in seeds]
    c = np.stack([p[0] for p in probs]); A = np.stack([p[1] for p in probs])
    b = np.stack([p[2] for p in probs]); basis = np.stack([p[3] for p in probs])
    solver = BatchedPrimalSolver(c, A, b, basis, log_cap=1024)
    res = solver.solve(100000)
    logs = solver.pivot_logs()
    golden = {cs["seed"]: cs["ref"] for cs in load("random.json")
              if cs.get("gen") == "gen_primal" and cs["m"] == 64 and cs["n"] == 128 and cs["maxiters"] == 100000
              and cs["solver"] == "PrimalRevisedSimplexSolver"}
    for i, s in enumerate(seeds):
        o = O.OracleRevised(*probs[i], log_cap=1024).solve(100000)
        assert logs[i].tolist() == o.log.tolist()
        assert res[i].optimum and res[i].iters == o.iters
        assert np.array_equal(res[i].x, o.x) and res[i].cost == o.cost
        assert res[i].basis.tolist() == o.basis.tolist()
        if s in golden:
            assert logs[i].tolist() == golden[s]["pivots"]
            assert_close(res[i].x, golden[s]["x"], "x")
            assert_close(res[i].cost, golden[s]["cost"], "cost")


@pytest.mark.parametrize("m,n", [(4, 8), (16, 32), (32, 64), (48, 160)])
def test_batched_other_shapes_and_limits(m, n):
    from linprog_amd.batched import BatchedPrimalSolver
    from oracle import oracle as O

    probs = [gen.gen_primal(m, n, s) for s in range(6)]
    solver = BatchedPrimalSolver(*[np.stack([p[k] for p in probs]) for k in range(4)], log_cap=512)
    res = solver.solve(30)  # some hit the iteration limit: optimum False, iters == 30
    for i, p in enumerate(probs):
        o = O.OracleRevised(*p, log_cap=512).solve(30)
        assert solver.pivot_logs()[i].tolist() == o.log.tolist()
        assert res[i].iters == o.iters and res[i].optimum == (o.status == O.OPTIMAL)
        assert np.array_equal(res[i].x, o.x)


@pytest.mark.parametrize("m,n", [(5, 12), (8, 16), (13, 40), (24, 64), (37, 100), (40, 128), (56, 200), (64, 256), (61, 129)])
def test_batched_both_kernels_bit_identical(m, n, monkeypatch):
    """The one-wave-per-LP kernel (whole and partial blocks of 8 rows; 1, 2 and 4 columns per lane) and the
    shape-generic kernel return the same bits as each other and as the oracle, on full solves."""
    from linprog_amd.batched import BatchedPrimalSolver
    from oracle import oracle as O

    probs = [gen.gen_primal(m, n, 100 + s) for s in range(5)]
    out = {}
    for kernel in ("wave", "generic"):
        monkeypatch.setenv("LPX_BATCH_KERNEL", kernel)
        solver = BatchedPrimalSolver(*[np.stack([p[k] for p in probs]) for k in range(4)], log_cap=2048)
        res = solver.solve(100000)
        out[kernel] = (res, solver.pivot_logs())
    for i, p in enumerate(probs):
        o = O.OracleRevised(*p, log_cap=2048).solve(100000)
        for kernel in ("wave", "generic"):
            res, logs = out[kernel]
            assert logs[i].tolist() == o.log.tolist(), kernel
            assert res[i].iters == o.iters and res[i].optimum == (o.status == O.OPTIMAL), kernel
            assert np.array_equal(res[i].x, o.x) and res[i].cost == o.cost, kernel
            assert res[i].basis.tolist() == o.basis.tolist(), kernel


def test_batched_generic_kernel_above_one_wave(monkeypatch):
    """m > 64 only fits the generic kernel; pinning the wave kernel there is an error, as is a bad pin."""
    from linprog_amd.batched import BatchedPrimalSolver
    from linprog_amd.exceptions import LpxBackendError
    from oracle import oracle as O

    probs = [gen.gen_primal(80, 170, s) for s in range(3)]
    args = [np.stack([p[k] for p in probs]) for k in range(4)]
    res = BatchedPrimalSolver(*args).solve(100000)
    for i, p in enumerate(probs):
        o = O.OracleRevised(*p).solve(100000)
        assert res[i].iters == o.iters and np.array_equal(res[i].x, o.x) and res[i].cost == o.cost
    for bad in ("wave", "nonsense"):
        monkeypatch.setenv("LPX_BATCH_KERNEL", bad)
        with pytest.raises(LpxBackendError):
            BatchedPrimalSolver(*args).solve(10)


def test_batched_rejects_oversize():
    from linprog_amd import _lib
    from linprog_amd.batched import BatchedPrimalSolver
    from linprog_amd.exceptions import LpxBackendError

    probs = [gen.gen_primal(130, 260, 0)]
    solver = BatchedPrimalSolver(*[np.stack([p[k] for p in probs]) for k in range(4)])
    with pytest.raises(LpxBackendError):
        solver.solve(10)


@pytest.mark.parametrize("m,n,seed", [(16, 32, 0), (64, 128, 1), (200, 520, 2), (600, 1300, 3)])
def test_tableau_dantzig_opt_in_matches_oracle(m, n, seed):
    """Opt-in most-negative pricing: no counterpart in the reference (parity unpinned); checked bit-exactly against
    the oracle's restatement of the same rule, and it must reach the same optimal cost as the default rule."""
    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    c, A, b, basis = gen.gen_primal(m, n, seed)
    s = logged(PrimalTableauSimplexSolver, 20000)(c, A, b, basis, pricing="dantzig")
    res = s.solve(20000)
    o = O.OracleTableau(c, A, b, basis, log_cap=20000, pricing=1).solve(20000)
    assert res.optimum and o.status == O.OPTIMAL
    assert s.pivot_log().tolist() == o.log.tolist()
    assert np.array_equal(res.x, o.x)  # elementwise path: bit-identical; the cost is a reduction (tree order)
    assert abs(res.cost - o.cost) <= 1e-13 * max(1.0, abs(o.cost))
    if m <= 64:
        ref = O.OracleTableau(c, A, b, basis).solve(100000)
        assert abs(ref.cost - res.cost) <= 1e-9 * max(1.0, abs(ref.cost))
        assert res.iters <= ref.iters


DUAL_CASES = [c for c in load("random.json") + load("textbook.json") + load("random_xlarge.json") if c["solver"].startswith("Dual")]


@pytest.mark.parametrize("chain", ["one-launch", "four-launch", "deferred"])
@pytest.mark.parametrize("condensed", [True, False], ids=["condensed", "full"])
@pytest.mark.parametrize("cs", DUAL_CASES, ids=case_id)
def test_dual_tableau_matches_reference(cs, condensed, chain, monkeypatch):
    """DualTableauSimplexSolver takes DualNaive/DualRevisedSimplexSolver's pivots (goldens), raises where they raise,
    and is bit-identical to the oracle's dual tableau loop -- through both launch chains: one launch per pivot on two
    alternating buffers (k_td_fused, default at these sizes) and the in-place row -> ratio -> prepare -> update chain."""
    from linprog_amd import exceptions
    from linprog_amd.tableau import DualTableauSimplexSolver
    from oracle import oracle as O

    if chain == "four-launch":
        monkeypatch.setenv("LPX_TABLEAU_FUSED", "0")
    if chain == "deferred":  # blocks of 8 pivots decided from the stale tableau, one pass applies them (k_tbd_row / _column)
        monkeypatch.setenv("LPX_TABLEAU_BLOCK", "8")
        monkeypatch.setenv("LPX_TABLEAU_BLOCK_MIN_MB", "0")
    ref = cs["ref"]
    c, A, b, basis = problem_of(cs)
    if ref["exception"] == "BasisIsDualInfeasibleError":
        with pytest.raises(exceptions.BasisIsDualInfeasibleError):
            DualTableauSimplexSolver(c, A, b, basis, condensed=condensed)
        return
    cap = max(16, len(ref["pivots"]) + 1)
    Solver = type("T", (DualTableauSimplexSolver,), {"_LOG_CAP": cap})
    solver = Solver(c, A, b, basis, condensed=condensed)
    maxiters = cs.get("maxiters", 100)
    o = O.OracleTableau(c, A, b, basis, kind="dual", log_cap=cap)
    ro = o.solve(maxiters)
    if ref["exception"]:
        with pytest.raises(getattr(exceptions, ref["exception"])):
            solver.solve(maxiters)
        assert solver.pivot_log().tolist() == ref["pivots"]
        return
    res = solver.solve(maxiters)
    assert solver.pivot_log().tolist() == ref["pivots"] == ro.log.tolist()
    assert res.iters == ref["iters"] and res.optimum == ref["optimum"]
    assert res.basis.tolist() == ref["basis"]
    if "Naive" in cs["solver"]:  # the naive reference re-inverts in float32 every pivot: values are float32-accurate
        assert np.allclose(res.x, ref["x"], rtol=1e-5, atol=1e-5) and abs(res.cost - ref["cost"]) < 1e-5
    else:
        if "x" in ref:  # the large prefixes keep the cost only
            assert_close(res.x, ref["x"], "x")
        assert_close(res.cost, ref["cost"], "cost")
    if not condensed or np.array_equal(np.sort(basis), np.arange(len(c) - len(b), len(c))):
        # same layout as the oracle's (or an exact identity start, where the layouts cannot differ): bit-identical x
        assert np.array_equal(res.x, ro.x)  # the cost is a tree dot on the device, sequential in the oracle
        assert abs(res.cost - ro.cost) <= 1e-13 * max(1.0, abs(ro.cost))


@pytest.mark.parametrize("rows", ["0", "1", "4", "deferred8", "deferred16", "deferred32"])
@pytest.mark.parametrize("m,n,condensed", [(300, 700, True), (300, 700, False), (530, 1061, True), (64, 1200, True),
                                           (700, 760, True)])
def test_dual_one_launch_chain_bit_exact_multi_workgroup(m, n, condensed, rows, monkeypatch):
    """The dual single-launch chain with several lookahead workgroups, several column strips and every update tile
    height, in calls of odd and even length (ping-pong parity, hipGraph replays): pivot log, tableau, basis and column
    maps bit-identical to the oracle's dual loop on the same layout; then on to the optimum, and a further solve() there
    moves nothing."""
    from linprog_amd.tableau import DualTableauSimplexSolver
    from oracle import oracle as O

    if rows.startswith("deferred"):  # the deferred-update chain of the dual rules, calls ending anywhere in a block
        monkeypatch.setenv("LPX_TABLEAU_BLOCK", rows[8:])
        monkeypatch.setenv("LPX_TABLEAU_BLOCK_MIN_MB", "0")
    else:
        monkeypatch.setenv("LPX_FUSED_ROWS", rows)
    c, A, b, basis = gen.gen_dual(m, n, 5)
    s = logged(DualTableauSimplexSolver, 4096)(c, A, b, basis, condensed=condensed)
    assert s._engine.T_alt is not None
    o = O.OracleTableau(c, A, b, basis, kind="dual", log_cap=4096)  # every column (the oracle's dual layout)
    done = 0
    for k in (1, 2, 5, 64, 33):
        res = s.solve(k)
        r = o.solve(k)
        done += k
        assert s.pivot_log().tolist() == o.log[:done].tolist()
        eng = s._engine
        T = eng.T.cpu().numpy()
        if condensed:  # slot q carries variable slot_var[q]; the slots and the basis partition the variables
            slot_var, var_slot = eng.slot_var.cpu().numpy(), eng.var_slot.cpu().numpy()
            assert sorted(slot_var.tolist() + res.basis.tolist()) == list(range(n))
            assert np.array_equal(var_slot[slot_var], np.arange(eng.n)) and np.all(var_slot[res.basis] == -1)
            assert np.array_equal(T[:, :eng.n], o.T[:, slot_var]) and np.array_equal(T[:, eng.n], o.T[:, n])
        else:
            assert np.array_equal(T[:, :n + 1], o.T[:, :n + 1])
        assert res.basis.tolist() == r.basis.tolist() and res.iters == r.iters and np.array_equal(res.x, r.x)
        assert_close(res.cost, r.cost, "cost")  # dot(c, x): the device sums it as a tree, the oracle in index order
    res = s.solve(100000)
    r = o.solve(100000)
    assert res.optimum is True and r.status == O.OPTIMAL and res.iters == r.iters
    gl = s.pivot_log().tolist()
    assert len(gl) == done + res.iters - 1 and gl == o.log[:len(gl)].tolist() and np.array_equal(res.x, r.x)
    T0 = s._engine.T.clone()
    again = s.solve(100)
    assert again.optimum is True and again.iters == 1 and bool((s._engine.T == T0).all())


@pytest.mark.parametrize("m,n,seed", [(300, 700, 5), (1000, 2000, 6)])
def test_dual_tableau_larger_vs_oracle(m, n, seed):
    """Multi-strip sizes (several workgroups per reduction, hipGraph replay): sequence and values against the oracle."""
    from linprog_amd.tableau import DualTableauSimplexSolver
    from oracle import oracle as O

    c, A, b, basis = gen.gen_dual(m, n, seed)
    Solver = type("T", (DualTableauSimplexSolver,), {"_LOG_CAP": 4096})
    solver = Solver(c, A, b, basis)  # condensed layout (default)
    res = solver.solve(600)
    ro = O.OracleTableau(c, A, b, basis, kind="dual", log_cap=4096).solve(600)
    assert solver.pivot_log().tolist() == ro.log.tolist() and len(ro.log) > 50
    assert res.iters == ro.iters
    assert_close(res.x, ro.x, "x")
    assert_close(res.cost, ro.cost, "cost")


@pytest.mark.parametrize("m,k", [(8, 8), (24, 40), (64, 64), (37, 100), (80, 120)])
def test_batched_inequality_form_equals_materialised_problem(m, k):
    """lpx_batched_primal_solve_ineq (G x <= h: slack identity generated in the kernel) returns the same bits as the
    standard-form entry point on [G | I] and as the oracle; both kernels (m = 80 takes the generic one)."""
    from linprog_amd.batched import BatchedPrimalSolver
    from oracle import oracle as O

    probs = [gen.gen_primal(m, k + m, 40 + s) for s in range(4)]
    c = np.stack([p[0] for p in probs]); A = np.stack([p[1] for p in probs]); b = np.stack([p[2] for p in probs])
    basis = np.stack([p[3] for p in probs])
    full = BatchedPrimalSolver(c, A, b, basis, log_cap=2048)
    rf = full.solve(100000)
    ineq = BatchedPrimalSolver.from_inequalities(c[:, :k], A[:, :, :k], b, log_cap=2048)
    ri = ineq.solve(100000)
    for i, p in enumerate(probs):
        o = O.OracleRevised(*p, log_cap=2048).solve(100000)
        assert ineq.pivot_logs()[i].tolist() == full.pivot_logs()[i].tolist() == o.log.tolist()
        assert np.array_equal(ri[i].x, rf[i].x) and np.array_equal(ri[i].x, o.x)
        assert ri[i].cost == rf[i].cost == o.cost and ri[i].iters == o.iters
        assert ri[i].basis.tolist() == o.basis.tolist()


def test_solve_many_mixed_shapes():
    from linprog_amd.batched import solve_many
    from oracle import oracle as O

    shapes = [(8, 16), (64, 128), (8, 16), (24, 50), (64, 128), (80, 170)]
    probs = [gen.gen_primal(m, n, 70 + i) for i, (m, n) in enumerate(shapes)]
    res = solve_many(probs, 100000)
    for p, r in zip(probs, res):
        o = O.OracleRevised(*p, log_cap=4).solve(100000)
        assert r.optimum and r.iters == o.iters and np.array_equal(r.x, o.x) and r.cost == o.cost
    fast = solve_many(probs, 100000, form="tableau")  # (80, 170) does not fit a wave: two waves per LP take it
    for r, f in zip(res, fast):
        assert f.optimum and f.iters == r.iters and f.basis.tolist() == r.basis.tolist()
        assert_close(f.x, r.x, "x")


def test_solve_many_routes_non_slack_starting_bases_to_the_revised_form():
    """A group that starts from a general basis (float32 inverse, primal_solvers.py:37-38) must not be routed to the
    tableau kernel by form="tableau": that kernel CARRIES the float32 starting tableau inv32 @ A for the whole solve, while
    the reference (and the revised-form kernel) recompute inv64 @ A in float64 from the second pivot on.  solve_many
    therefore gives such groups the revised-form kernel in both modes: bit-identical results, which are in turn
    bit-identical to the oracle's batch loop from the same float32 starting inverse.  Against the single-LP oracle class
    (x_B = inv32 @ b through BLAS, another summation order in float32) they agree to float32 accuracy -- the accuracy the
    reference's own result has from such a start -- with identical pivot counts and final bases."""
    from linprog_amd.batched import BatchedTableauSolver, solve_many
    from oracle import oracle as O

    probs = []
    for i in range(6):
        c, A, b, basis = gen.gen_primal(24, 48, 900 + i)
        mid = O.OracleRevised(c, A, b, basis, log_cap=16).solve(7).basis.copy()  # a feasible non-slack basis
        probs.append((c, A, b, mid))
    want = [O.OracleRevised(*p, log_cap=4).solve(100000) for p in probs]
    rev = solve_many(probs, 100000, form="revised")
    tab = solve_many(probs, 100000, form="tableau")
    c, A, b, basis = [np.stack([np.asarray(p[k]) for p in probs]) for k in range(4)]
    ob = O.primal_solve_batch(c, A, b, basis, 100000, threads=2)  # forms the float32 starting inverse per LP itself
    for i, (r, t, o) in enumerate(zip(rev, tab, want)):
        assert np.array_equal(r.x, t.x) and r.cost == t.cost and r.iters == t.iters  # same kernel in both modes
        assert np.array_equal(r.x, ob["x"][i]) and r.cost == ob["cost"][i] and r.iters == ob["iters"][i]
        assert r.optimum and r.iters == o.iters and r.basis.tolist() == o.basis.tolist()
        assert np.allclose(r.x, o.x, rtol=1e-5, atol=1e-5) and abs(r.cost - o.cost) <= 1e-5 * max(1, abs(o.cost))
    # the tableau kernel itself from those starts: float32 accuracy as documented in BatchedTableauSolver
    s = BatchedTableauSolver(c, A, b, basis)
    assert s.start is not None
    for r, o in zip(s.solve(100000), want):
        assert r.optimum and np.allclose(r.x, o.x, rtol=1e-5, atol=1e-5) and abs(r.cost - o.cost) <= 1e-5 * max(1, abs(o.cost))


@pytest.mark.parametrize("m,n", [(8, 16), (13, 40), (32, 64), (37, 100), (64, 128), (64, 256), (60, 150)])
def test_batched_dual_bit_identical_to_oracle(m, n):
    """BatchedDualSolver (one wave per LP, dual mode of the wave kernel) against the oracle's dual loop: pivot logs, x,
    cost, basis, iters and status bit for bit; whole and partial row blocks, 1 / 2 / 4 columns per lane."""
    from linprog_amd.batched import BatchedDualSolver
    from oracle import oracle as O

    probs = [gen.gen_dual(m, n, 200 + s) for s in range(5)]
    args = [np.stack([p[k] for p in probs]) for k in range(4)]
    solver = BatchedDualSolver(*args, log_cap=2048)
    res = solver.solve(100000)
    for i, p in enumerate(probs):
        o = O.OracleRevised(*p, kind="dual", log_cap=2048).solve(100000)
        assert solver.pivot_logs()[i].tolist() == o.log.tolist()
        assert res[i].iters == o.iters and res[i].optimum is (True if o.status == O.OPTIMAL else None)
        assert np.array_equal(res[i].x, o.x) and res[i].cost == o.cost
        assert res[i].basis.tolist() == o.basis.tolist()


def test_batched_dual_reference_sequences_limits_and_errors():
    """Goldens of DualRevisedSimplexSolver (m <= 64): the batched dual kernel takes the reference's pivots; a maxiters
    hit leaves optimum None; dual unbounded LPs are flagged in status; a dual infeasible start raises."""
    from linprog_amd import _lib, exceptions
    from linprog_amd.batched import BatchedDualSolver

    cases = [c for c in load("random.json") if c["solver"] == "DualRevisedSimplexSolver" and c.get("gen") == "gen_dual"
             and c["m"] <= 64 and not c["ref"]["exception"]]
    assert len(cases) >= 10
    for cs in cases:
        c, A, b, basis = problem_of(cs)
        s = BatchedDualSolver(c[None], A[None], b[None], basis[None], log_cap=1024)
        r = s.solve(cs["maxiters"])[0]
        ref = cs["ref"]
        assert s.pivot_logs()[0].tolist() == ref["pivots"]
        assert r.iters == ref["iters"] and r.optimum == ref["optimum"] and r.basis.tolist() == ref["basis"]
        assert_close(r.x, ref["x"], "x")
        assert_close(r.cost, ref["cost"], "cost")
    ub = [c for c in load("random.json") if c.get("gen") == "gen_dual_unbounded" and c["solver"] == "DualRevisedSimplexSolver"][0]
    c, A, b, basis = problem_of(ub)
    s = BatchedDualSolver(c[None], A[None], b[None], basis[None], log_cap=64)
    s.solve(1000)
    assert s.raw["status"][0] == _lib.DUAL_UNBOUNDED and s.pivot_logs()[0].tolist() == ub["ref"]["pivots"]
    c, A, b, basis = gen.gen_primal(8, 16, 0)  # not dual feasible at the slack basis
    with pytest.raises(exceptions.BasisIsDualInfeasibleError):
        BatchedDualSolver(c[None], A[None], b[None], basis[None])


def test_tableau_without_nonbasic_columns_falls_back_to_the_full_layout():
    """n = m: every column is basic, nothing to condense; the solver answers 'optimal' at once like the reference."""
    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver
    from linprog_amd.tableau import PrimalTableauSimplexSolver

    rng = np.random.default_rng(3)
    A = np.eye(6) + 0.1 * rng.random((6, 6))
    b = A @ rng.uniform(1, 2, 6)
    c = rng.uniform(-1, 1, 6)
    basis = np.arange(6)
    t = PrimalTableauSimplexSolver(c, A, b, basis).solve()
    r = PrimalRevisedSimplexSolver(c, A, b, basis).solve()
    assert t.optimum and r.optimum and t.iters == r.iters == 1
    assert np.allclose(t.x, r.x, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("m,n", [(4, 8), (5, 12), (16, 32), (37, 100), (40, 104), (64, 128), (61, 125), (64, 70),
                                 (80, 170), (96, 192), (64, 256), (70, 80), (65, 130), (128, 256), (100, 340), (30, 200),
                                 (128, 384), (65, 66), (2, 200), (128, 129), (256, 300), (129, 257), (200, 300), (3, 131),
                                 (256, 384)])
def test_batched_tableau_kernel_bit_identical_to_the_tableau_oracle(m, n):
    """lpx_batched_tableau_solve (condensed tableau in registers, one wave per LP): per LP the pivot log, x, cost, basis and
    iters of the oracle's tableau loop, bit for bit, on full solves and on solves cut off by maxiters; the revised-form
    batched kernel takes the same pivots and agrees in x to 1e-9."""
    from linprog_amd.batched import BatchedPrimalSolver, BatchedTableauSolver
    from oracle import oracle as O

    probs = [gen.gen_primal(m, n, 300 + s) for s in range(6)]
    args = [np.stack([p[k] for p in probs]) for k in range(4)]
    for maxiters in (100000, 7):
        tab = BatchedTableauSolver(*args, log_cap=4096)
        rt = tab.solve(maxiters)
        from linprog_amd.exceptions import LpxBackendError

        try:  # beyond one wave the multi-wave kernel also takes shapes that the LDS-resident revised kernels cannot hold
            rev = BatchedPrimalSolver(*args, log_cap=4096)
            rr = rev.solve(maxiters)
        except LpxBackendError:
            assert m > 64 or n > 256
            rev = rr = None
        for i, p in enumerate(probs):
            o = O.OracleTableau(*p, log_cap=4096).solve(maxiters)
            assert tab.pivot_logs()[i].tolist() == o.log.tolist()
            if rev is None:
                assert rt[i].iters == o.iters and np.array_equal(rt[i].x, o.x) and rt[i].cost == o.cost
                continue
            assert o.log.tolist() == rev.pivot_logs()[i].tolist()
            assert rt[i].iters == o.iters and rt[i].optimum == (o.status == O.OPTIMAL)
            assert np.array_equal(rt[i].x, o.x) and rt[i].cost == o.cost
            assert rt[i].basis.tolist() == o.basis.tolist()
            assert_close(rt[i].x, rr[i].x, "x vs the revised-form kernel")


def test_batched_tableau_kernel_reference_sequences_and_unbounded():
    from linprog_amd.batched import BatchedTableauSolver
    from linprog_amd import _lib

    golden = {cs["seed"]: cs["ref"] for cs in load("random.json")
              if cs.get("gen") == "gen_primal" and cs["m"] == 64 and cs["n"] == 128 and cs["maxiters"] == 100000
              and cs["solver"] == "PrimalRevisedSimplexSolver"}
    assert golden
    seeds = sorted(golden)
    probs = [gen.gen_primal(64, 128, s) for s in seeds]
    solver = BatchedTableauSolver(*[np.stack([p[k] for p in probs]) for k in range(4)], log_cap=2048)
    res = solver.solve(100000)
    for i, s in enumerate(seeds):
        assert solver.pivot_logs()[i].tolist() == golden[s]["pivots"]
        assert res[i].iters == golden[s]["iters"] and res[i].optimum
        assert_close(res[i].x, golden[s]["x"], "x")
        assert_close(res[i].cost, golden[s]["cost"], "cost")
    # an unbounded LP in the batch (its structural columns negated) does not disturb its neighbours
    c, A, b, basis = [np.stack([p[k] for p in probs[:3]]) for k in range(4)]
    A[1, :, :64] *= -1
    sol = BatchedTableauSolver(c, A, b, basis)
    out = sol.solve(100000)
    assert sol.raw["status"].tolist() == [_lib.OPTIMAL, _lib.PRIMAL_UNBOUNDED, _lib.OPTIMAL]
    assert np.array_equal(out[0].x, res[0].x) and np.array_equal(out[2].x, res[2].x)
    with pytest.raises(ValueError):
        BatchedTableauSolver(*[np.stack([p[k] for p in [gen.gen_primal(130, 260, 0)]]) for k in range(4)])  # 4 x 3 waves
    # the same in the multi-wave kernel (80 x 170: four waves per LP)
    probs = [gen.gen_primal(80, 170, s) for s in range(3)]
    c, A, b, basis = [np.stack([p[k] for p in probs]) for k in range(4)]
    ref = BatchedTableauSolver(c, A, b, basis).solve(100000)
    A[1, :, :90] *= -1
    sol = BatchedTableauSolver(c, A, b, basis)
    out = sol.solve(100000)
    assert sol.raw["status"].tolist() == [_lib.OPTIMAL, _lib.PRIMAL_UNBOUNDED, _lib.OPTIMAL]
    assert np.array_equal(out[0].x, ref[0].x) and np.array_equal(out[2].x, ref[2].x)


@pytest.mark.parametrize("m,k", [(8, 8), (24, 40), (64, 64), (37, 60), (80, 90), (64, 192), (128, 128)])
def test_batched_tableau_inequality_form_equals_materialised_problem(m, k):
    """lpx_batched_tableau_solve_ineq (G x <= h; the slack identity is never stored) returns the same bits as the
    standard-form entry point on [G | I]."""
    from linprog_amd.batched import BatchedTableauSolver

    probs = [gen.gen_primal(m, k + m, 70 + s) for s in range(4)]
    c = np.stack([p[0] for p in probs]); A = np.stack([p[1] for p in probs]); b = np.stack([p[2] for p in probs])
    basis = np.stack([p[3] for p in probs])
    full = BatchedTableauSolver(c, A, b, basis, log_cap=2048)
    rf = full.solve(100000)
    ineq = BatchedTableauSolver.from_inequalities(c[:, :k], A[:, :, :k], b, log_cap=2048)
    ri = ineq.solve(100000)
    for i in range(4):
        assert ineq.pivot_logs()[i].tolist() == full.pivot_logs()[i].tolist()
        assert np.array_equal(ri[i].x, rf[i].x) and ri[i].cost == rf[i].cost and ri[i].iters == rf[i].iters
        assert ri[i].basis.tolist() == rf[i].basis.tolist() and ri[i].optimum


@pytest.mark.parametrize("m,n", [(8, 16), (13, 40), (32, 64), (37, 100), (64, 128), (60, 124), (80, 170), (65, 130),
                                 (64, 256), (128, 256), (100, 340), (200, 300), (2, 200), (129, 257)])
def test_batched_dual_tableau_kernel_bit_identical_to_the_dual_tableau_oracle(m, n):
    """lpx_batched_tableau_dual_solve (one wave per LP, a grid of waves beyond 64 x 64 slots): per LP the pivot log, x, cost,
    basis, iters and status of the oracle's dual tableau loop, bit for bit; the revised-form dual kernel takes the same
    pivots (shapes it holds)."""
    from linprog_amd.batched import BatchedDualSolver, BatchedDualTableauSolver
    from linprog_amd.exceptions import LpxBackendError
    from oracle import oracle as O

    probs = [gen.gen_dual(m, n, 400 + s) for s in range(5)]
    args = [np.stack([p[k] for p in probs]) for k in range(4)]
    for maxiters in (100000, 9):
        tab = BatchedDualTableauSolver(*args, log_cap=2048)
        rt = tab.solve(maxiters)
        try:
            rev = BatchedDualSolver(*args, log_cap=2048)
            rev.solve(maxiters)
        except (LpxBackendError, ValueError):
            assert m > 64 or n > 256
            rev = None
        for i, p in enumerate(probs):
            o = O.OracleTableau(*p, kind="dual", log_cap=2048).solve(maxiters)
            assert tab.pivot_logs()[i].tolist() == o.log.tolist()
            assert rev is None or o.log.tolist() == rev.pivot_logs()[i].tolist()
            assert rt[i].iters == o.iters and rt[i].optimum is (True if o.status == O.OPTIMAL else None)
            assert np.array_equal(rt[i].x, o.x) and rt[i].cost == o.cost
            assert rt[i].basis.tolist() == o.basis.tolist()


def test_batched_tableau_kernel_on_degenerate_goldens():
    """Ties in the ratio test and equal reduced costs (small-integer data, zeros in b): "first index" decides the pivots
    (primal_solvers.py:137, :166).  Every degenerate golden through the tableau batched kernel: the reference's pivot
    sequence, basis, iters, x and cost; unbounded cases report the status."""
    from linprog_amd import _lib
    from linprog_amd.batched import BatchedTableauSolver

    cases = [cs for cs in load("degenerate.json") if cs["solver"] == "PrimalRevisedSimplexSolver"]
    shapes = sorted({(cs["m"], cs["n"]) for cs in cases})
    assert len(cases) >= 12
    for m, n in shapes:
        grp = [cs for cs in cases if (cs["m"], cs["n"]) == (m, n)]
        probs = [problem_of(cs) for cs in grp]
        solver = BatchedTableauSolver(*[np.stack([p[k] for p in probs]) for k in range(4)], log_cap=8192)
        res = solver.solve(grp[0].get("maxiters", 100))
        for i, cs in enumerate(grp):
            ref = cs["ref"]
            assert cs.get("maxiters", 100) == grp[0].get("maxiters", 100)
            assert solver.pivot_logs()[i].tolist() == ref["pivots"], (m, n, cs["seed"])
            if ref["exception"] == "PrimalIsUnboundedError":
                assert solver.raw["status"][i] == _lib.PRIMAL_UNBOUNDED
                continue
            assert not ref["exception"]
            assert res[i].iters == ref["iters"] and res[i].optimum == ref["optimum"] and res[i].basis.tolist() == ref["basis"]
            assert_close(res[i].x, ref["x"], "x")
            assert_close(res[i].cost, ref["cost"], "cost")


@pytest.mark.parametrize("m,n", [(5, 12), (16, 32), (37, 100), (64, 128), (80, 170), (64, 256), (130, 250)])
def test_batched_tableau_kernel_from_a_general_starting_basis(m, n):
    """lpx_batched_tableau_solve_from: any primal feasible starting basis (here the basis after 9 oracle pivots; one LP of
    the batch starts from a permuted slack basis).  The float32 constructor state of the reference (primal_solvers.py:37-38,
    :64) is carried as a condensed tableau: same arithmetic as the unbatched condensed tableau solver (bit-identical x,
    cost, pivots), the reference's (revised-form) pivot sequence, x to float32 accuracy of the start."""
    from linprog_amd.batched import BatchedTableauSolver
    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    probs = []
    for s_ in range(5):
        c, A, b, basis = gen.gen_primal(m, n, 500 + s_)
        mid = basis[::-1].copy() if s_ == 0 else O.OracleRevised(c, A, b, basis, log_cap=16).solve(9).basis.copy()
        probs.append((c, A, b, mid))
    solver = BatchedTableauSolver(*[np.stack([p[k] for p in probs]) for k in range(4)], log_cap=4096)
    assert solver.start is not None
    for maxiters in (100000, 6):
        res = solver.solve(maxiters)
        for i, p in enumerate(probs):
            ref = O.OracleRevised(*p, log_cap=4096).solve(maxiters)
            one = type("L", (PrimalTableauSimplexSolver,), {"_LOG_CAP": 4096})(*p, condensed=True)
            r1 = one.solve(maxiters)
            assert solver.pivot_logs()[i].tolist() == one.pivot_log().tolist() == ref.log.tolist()
            assert np.array_equal(res[i].x, r1.x) and res[i].iters == r1.iters and res[i].basis.tolist() == r1.basis.tolist()
            assert abs(res[i].cost - r1.cost) <= 1e-13 * max(1.0, abs(r1.cost))  # tree dot there, sequential here
            assert res[i].optimum == (ref.status == O.OPTIMAL)
            assert np.allclose(res[i].x, ref.x, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("m,n,count", [(16, 32, 40), (64, 128, 24), (33, 67, 9), (128, 256, 6)])
def test_batched_bounded_bit_identical_to_oracle(m, n, count):
    """BatchedBoundedSolver (one workgroup per LP, lpx_revised_solve_small_batch): step logs (basis changes and bound
    flips), x, x_B-derived basis, per-variable bound states and iteration counts bit-identical to the oracle's bounded
    loop LP by LP -- whose own logs are pinned to the reference's by tests/golden/bounded.json -- and the golden cases
    themselves (seeds 0, 1) reproduce the reference's steps."""
    from linprog_amd import _lib
    from linprog_amd.batched import BatchedBoundedSolver
    from oracle import oracle as O

    probs = [gen.gen_box(m, n, s) for s in range(count)]
    c, A, b, lb, ub, basis = [np.stack([p[k] for p in probs]) for k in range(6)]
    vs = np.full((count, n), _lib.VS_LOWER, dtype=np.int32)  # gen_box: every structural variable starts at its lower bound
    s = BatchedBoundedSolver(c, A, b, lb, ub, basis, vs, log_cap=2048)
    res = s.solve(100000)
    logs = s.pivot_logs()
    golden = {cs["seed"]: cs["ref"] for cs in load("bounded.json") if cs["m"] == m and cs["n"] == n}
    for i, p in enumerate(probs):
        o = O.OracleBounded(*p).solve(100000)
        assert res[i].optimum and res[i].iters == o.iters and res[i].basis.tolist() == o.basis.tolist()
        assert logs[i].tolist() == o.log.tolist()
        assert np.array_equal(res[i].x, o.x) and np.array_equal(s.raw["var_state"][i], o.var_state)
        assert_close(res[i].cost, o.cost, "cost")
        if i in golden:
            ref = golden[i]
            assert [list(map(int, e)) for e in logs[i].tolist()] == [e[:3] for e in ref["steps"]]
            assert res[i].iters == ref["iters_total"]
            assert_close(res[i].x, ref["x"], "x")
            assert_close(res[i].cost, ref["cost"], "cost")


def test_batched_bounded_reentrant_and_upper_bound_start():
    """Calls of a few iterations each continue from the state the previous one left (solve(maxiters=1) repeated is how
    the goldens were traced), and a start with variables at their UPPER bound prices them with the opposite sign."""
    from linprog_amd import _lib
    from linprog_amd.batched import BatchedBoundedSolver
    from oracle import oracle as O

    m, n, count = 24, 48, 8
    probs = [gen.gen_box(m, n, 100 + s) for s in range(count)]
    c, A, b, lb, ub, basis = [np.stack([p[k] for p in probs]) for k in range(6)]
    vs = np.full((count, n), _lib.VS_LOWER, dtype=np.int32)
    vs[:, 1:5] = _lib.VS_UPPER
    s = BatchedBoundedSolver(c, A, b, lb, ub, basis, vs)
    oracles = [O.OracleBounded(p[0], p[1], p[2], p[3], p[4], p[5], [0] + list(range(5, n - m)), [1, 2, 3, 4]) for p in probs]
    for k in (1, 3, 2, 100000):
        res = s.solve(k)
        for i, o in enumerate(oracles):
            w = o.solve(k)
            assert res[i].iters == w.iters and res[i].basis.tolist() == w.basis.tolist() and np.array_equal(res[i].x, w.x)
            assert bool(res[i].optimum) == (w.status == O.OPTIMAL)


@pytest.mark.parametrize("condensed", [True, False], ids=["condensed", "full"])
@pytest.mark.parametrize("gen_name,m,n,seed", [("gen_primal", 64, 128, 1), ("gen_primal", 300, 700, 2), ("gen_primal", 530, 1061, 3),
                                                ("gen_primal_degenerate", 32, 64, 0), ("gen_primal_degenerate", 32, 64, 1),
                                                ("gen_primal_degenerate", 64, 128, 2)])
def test_harris_ratio_test_opt_in(gen_name, m, n, seed, condensed):
    """ratio="harris" (north_star's "Harris-style ratio test"; the reference has only the textbook test,
    primal_solvers.py:149-167, so this is PARITY UNPINNED against it and never the default): two passes with a 1e-9
    tolerance, largest pivot among the admissible rows.  Bit-exact against the oracle's restatement of the same rule
    (lpo_ratio_harris: pivot log, whole tableau), x_B >= -tol-scale, and the optimum is the LP's (scipy HiGHS)."""
    from scipy.optimize import linprog

    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    c, A, b, basis = getattr(gen, gen_name)(m, n, seed)
    cls = type("L", (PrimalTableauSimplexSolver,), {"_LOG_CAP": 8192})
    s = cls(c, A, b, basis, condensed=condensed, ratio="harris")
    o = O.OracleTableau(c, A, b, basis, condensed=condensed, log_cap=8192, ratio="harris")
    for k in (7, 100000):
        res, want = s.solve(k), o.solve(k)
        assert s.pivot_log().tolist() == o.log[:len(s.pivot_log())].tolist() and res.iters == want.iters
        w = o.nslots + 1
        assert np.array_equal(s._engine.T.cpu().numpy()[:, :w], o.T[:, :w])
    assert res.optimum and np.array_equal(res.x, want.x) and res.x.min() >= -1e-7
    A32, b32, c32 = (np.asarray(v, dtype=np.float32).astype(float) for v in (A, b, c))
    hs = linprog(c32, A_eq=A32, b_eq=b32, method="highs")
    assert hs.status == 0 and abs(res.cost - hs.fun) <= 1e-7 * max(1.0, abs(hs.fun))
    # the default rule on the same LP ends on the same optimum (by another path on the degenerate inputs)
    t = PrimalTableauSimplexSolver(c, A, b, basis, condensed=condensed).solve(100000)
    assert t.optimum and abs(t.cost - res.cost) <= 1e-7 * max(1.0, abs(t.cost))


def test_harris_is_refused_outside_the_unpartitioned_primal_loop():
    import ctypes as C

    from linprog_amd import _lib
    from linprog_amd.tableau import DualTableauSimplexSolver, PrimalTableauSimplexSolver

    c, A, b, basis = gen.gen_primal(40, 90, 1)
    eng = PrimalTableauSimplexSolver(c, A, b, basis, ratio="harris")._engine
    lib = _lib.load()
    assert lib.lpx_tableau_dual_iterate(C.byref(eng.s), 4, None) == -1 and b"Harris" in lib.lpx_last_error()
    eng.s.ratio_tol = -1.0
    assert lib.lpx_tableau_iterate(C.byref(eng.s), 4, None) == -1
