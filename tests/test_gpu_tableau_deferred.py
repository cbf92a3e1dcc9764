"""GPU parity of the tableau loop with deferred updates (lpx_tableau_blocked.hpp): KB pivots are decided from the stale
tableau plus the pending (eta column, pivot row) pairs, then one kernel applies the KB rank-1 updates in order.  Same
arithmetic in the same order per element, so everything must be bit-identical to the eager loops and the oracle."""
import numpy as np
import pytest

from tests._util import assert_close, case_id, gen, load, problem_of

pytestmark = pytest.mark.gpu

PRIMAL = [cs for cs in load("random.json") + load("textbook.json") + load("random_large.json") + load("random_xlarge.json") + load("degenerate.json")
          if cs["solver"] == "PrimalRevisedSimplexSolver" and not cs["ref"]["exception"]]


def logged(cls, cap):
    return type(cls.__name__ + "Logged", (cls,), {"_LOG_CAP": cap})


def deferred(monkeypatch, kb, rows=None, pipe=False):
    """pipe: the PIPELINED chain (lpx_tableau_pipelined: the pass of a block writes the other buffer while the next block
    is decided from the old one plus the pairs of both blocks -- chains of up to 2 kb), forced on at any size; else the
    in-place chain."""
    monkeypatch.setenv("LPX_TABLEAU_BLOCK", str(kb))
    monkeypatch.setenv("LPX_TABLEAU_BLOCK_MIN_MB", "0")  # at any size
    monkeypatch.setenv("LPX_TABLEAU_PIPELINE_MIN_MB", "0" if pipe else "-1")
    if rows is not None:
        monkeypatch.setenv("LPX_TB_ROWS", str(rows))


PIPE = pytest.mark.parametrize("pipe", [False, True], ids=["inplace", "pipelined"])


@PIPE
@pytest.mark.parametrize("kb", [8, 32])
@pytest.mark.parametrize("condensed", [True, False], ids=["condensed", "full"])
@pytest.mark.parametrize("cs", PRIMAL, ids=case_id)
def test_deferred_chain_matches_reference(cs, condensed, kb, pipe, monkeypatch):
    """Both layouts through the deferred-update chain against the reference's golden pivot sequences and results."""
    from linprog_amd.tableau import PrimalTableauSimplexSolver

    deferred(monkeypatch, kb, pipe=pipe)
    ref = cs["ref"]
    c, A, b, basis = problem_of(cs)
    s = logged(PrimalTableauSimplexSolver, max(16, len(ref["pivots"]) + 1))(c, A, b, basis, condensed=condensed)
    res = s.solve(cs.get("maxiters", 100))
    assert s.pivot_log().tolist() == ref["pivots"]
    assert res.iters == ref["iters"] and res.optimum == ref["optimum"] and res.basis.tolist() == ref["basis"]
    if "x" in ref:
        assert_close(res.x, ref["x"], "x")
    assert_close(res.cost, ref["cost"], "cost")


@PIPE
@pytest.mark.parametrize("kb,rows", [(8, 1), (8, 16), (16, 4), (32, 16), (32, 5)])
@pytest.mark.parametrize("m,n,condensed", [(300, 700, True), (300, 700, False), (530, 1061, True), (64, 1200, True),
                                           (700, 760, True), (96, 200, False)])
def test_deferred_chain_bit_exact_vs_oracle(m, n, condensed, kb, rows, pipe, monkeypatch):
    """Calls that end inside a block, on a block boundary and after several blocks (graph replays of whole blocks): pivot
    log, the whole tableau, basis and column maps bit-identical to the oracle's eager loop on the same layout."""
    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    deferred(monkeypatch, kb, rows, pipe)
    c, A, b, basis = gen.gen_primal(m, n, 5)
    s = logged(PrimalTableauSimplexSolver, 1024)(c, A, b, basis, condensed=condensed)
    o = O.OracleTableau(c, A, b, basis, condensed=condensed, log_cap=1024)
    done = 0
    for k in (1, 2, 5, 64, 33, 100, 96):
        res = s.solve(k)
        r = o.solve(k)
        done += min(k, r.iters)
        assert s.pivot_log().tolist() == o.log[:len(s.pivot_log())].tolist()
        w = o.nslots + 1
        T = s._engine.T.cpu().numpy()
        assert np.array_equal(T[:, :w], o.T[:, :w])
        assert res.basis.tolist() == r.basis.tolist() and res.iters == r.iters and np.array_equal(res.x, r.x)
        assert_close(res.cost, r.cost, "cost")  # dot(c, x): the device sums it as a tree, the oracle in index order
        if condensed:
            assert np.array_equal(s._engine.slot_var.cpu().numpy(), o.slot_var)


@PIPE
@pytest.mark.parametrize("rows", [None, 7, 64])
@pytest.mark.parametrize("m,n,condensed", [(300, 700, True), (530, 1061, True), (700, 760, True), (300, 700, False)])
def test_deferred_pass_large_tableau_instances_bit_exact_vs_oracle(m, n, condensed, rows, pipe, monkeypatch):
    """The instances the pass uses on tableaus beyond 192 MiB (non-temporal streams; blocks of 32: the condensed layout's
    full blocks in a kernel of their own, short blocks and the full layout in its twin -- lpx_tb_flush32.hip), forced on at
    a size the oracle finishes in seconds: full blocks, a short last block and calls that end inside a block."""
    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    deferred(monkeypatch, 32, rows, pipe)
    monkeypatch.setenv("LPX_RANK1_NT", "1")
    c, A, b, basis = gen.gen_primal(m, n, 11)
    s = logged(PrimalTableauSimplexSolver, 1024)(c, A, b, basis, condensed=condensed)
    o = O.OracleTableau(c, A, b, basis, condensed=condensed, log_cap=1024)
    for k in (64, 32, 5, 27, 100, 7, 128):
        res = s.solve(k)
        r = o.solve(k)
        assert s.pivot_log().tolist() == o.log[:len(s.pivot_log())].tolist()
        w = o.nslots + 1
        assert np.array_equal(s._engine.T.cpu().numpy()[:, :w], o.T[:, :w])
        assert res.basis.tolist() == r.basis.tolist() and res.iters == r.iters and np.array_equal(res.x, r.x)


@PIPE
@pytest.mark.parametrize("kb", [8, 32])
def test_deferred_chain_equals_the_eager_chains_bit_for_bit(kb, pipe, monkeypatch):
    """The same LP through the deferred chain and through the in-place three-kernel chain: identical tableaus."""
    from linprog_amd.tableau import PrimalTableauSimplexSolver

    c, A, b, basis = gen.gen_primal(400, 900, 3)
    monkeypatch.setenv("LPX_TABLEAU_BLOCK", "0")
    monkeypatch.setenv("LPX_TABLEAU_FUSED", "0")
    eager = logged(PrimalTableauSimplexSolver, 512)(c, A, b, basis)
    r0 = eager.solve(150)
    deferred(monkeypatch, kb, pipe=pipe)
    s = logged(PrimalTableauSimplexSolver, 512)(c, A, b, basis)
    r1 = s.solve(150)
    assert s.pivot_log().tolist() == eager.pivot_log().tolist()
    assert bool((s._engine.T == eager._engine.T).all())
    assert np.array_equal(r0.x, r1.x) and r0.cost == r1.cost and r0.iters == r1.iters


@PIPE
@pytest.mark.parametrize("kb", [8, 32])
def test_deferred_chain_runs_to_the_optimum_and_stays_there(kb, pipe, monkeypatch):
    """Full solve (termination in the middle of a block, inside a replayed graph), then a second solve() on the optimal
    state: iters == 1, nothing moves (tests/test_primal_base_solvers.py:24-34's re-entrancy)."""
    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    deferred(monkeypatch, kb, pipe=pipe)
    c, A, b, basis = gen.gen_primal(128, 256, 0)
    s = logged(PrimalTableauSimplexSolver, 1024)(c, A, b, basis)
    res = s.solve(100000)
    o = O.OracleTableau(c, A, b, basis, condensed=True, log_cap=1024)
    want = o.solve(100000)
    assert res.optimum and res.iters == want.iters and s.pivot_log().tolist() == want.log.tolist()
    assert np.array_equal(res.x, want.x)
    assert_close(res.cost, want.cost, "cost")
    w = o.nslots + 1
    assert np.array_equal(s._engine.T.cpu().numpy()[:, :w], o.T[:, :w])
    T0 = s._engine.T.clone()
    again = s.solve(100)
    assert again.optimum and again.iters == 1 and bool((s._engine.T == T0).all())


@PIPE
def test_deferred_chain_unbounded_inside_a_block(pipe, monkeypatch):
    """LPs that turn out unbounded after some pivots: the pending updates of the unfinished block are applied, the
    state equals the oracle's at the point where it stops."""
    from linprog_amd import exceptions
    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    deferred(monkeypatch, 8, pipe=pipe)
    found = 0
    for seed in range(60):
        c, A, b, basis = gen.gen_primal(24, 60, seed)
        A, c = A.copy(), c.copy()
        j = 30 + seed % 6  # a late structural column that can never block: entered only after the ones before it
        A[:, j] = -np.abs(A[:, j]) - 0.5
        c[j] = -abs(c[j]) - 1.0
        o = O.OracleTableau(c, A, b, basis, condensed=True, log_cap=256)
        r = o.solve(1000)
        if r.status != O.PRIMAL_UNBOUNDED or r.pivots < 3:
            continue
        s = logged(PrimalTableauSimplexSolver, 256)(c, A, b, basis)
        with pytest.raises(exceptions.PrimalIsUnboundedError):
            s.solve(1000)
        assert s.pivot_log().tolist() == r.log.tolist()
        w = o.nslots + 1
        assert np.array_equal(s._engine.T.cpu().numpy()[:, :w], o.T[:, :w])
        found += 1
        if found == 3:
            break
    assert found > 0, "no generated LP went unbounded after a few pivots"


@PIPE
@pytest.mark.parametrize("m,n,seed", [(64, 128, 1), (200, 520, 2), (600, 1300, 3)])
def test_deferred_chain_dantzig_opt_in_matches_oracle(m, n, seed, pipe, monkeypatch):
    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    deferred(monkeypatch, 16, pipe=pipe)
    c, A, b, basis = gen.gen_primal(m, n, seed)
    s = logged(PrimalTableauSimplexSolver, 4096)(c, A, b, basis, pricing="dantzig")
    o = O.OracleTableau(c, A, b, basis, condensed=True, log_cap=4096, pricing=1)
    res, want = s.solve(300), o.solve(300)
    assert s.pivot_log().tolist() == want.log.tolist() and res.iters == want.iters
    w = o.nslots + 1
    assert np.array_equal(s._engine.T.cpu().numpy()[:, :w], o.T[:, :w])


def test_block_entry_points_reject_more_pending_updates_than_the_largest_block(monkeypatch):
    """lpx_tableau_block_propose / _commit refuse a pending count beyond the largest block (32), _flush a block size it has no
    instance for.  The block itself is the caller's to agree on (column-partitioned: one value for all ranks, which may be
    larger than lpx_tableau_block_size() of a rank's own ragged slice -- ADVICE round 2)."""
    import ctypes as C

    from linprog_amd import _lib
    from linprog_amd.tableau import PrimalTableauSimplexSolver

    deferred(monkeypatch, 8)
    c, A, b, basis = gen.gen_primal(40, 90, 1)
    eng = PrimalTableauSimplexSolver(c, A, b, basis)._engine
    lib = _lib.load()
    assert lib.lpx_tableau_block_size(C.byref(eng.s)) == 8 and eng.kb == 8
    buf = eng.torch.zeros(int(lib.lpx_tableau_msg_len(eng.m)), dtype=eng.torch.float64, device="cuda")
    assert lib.lpx_tableau_block_propose(C.byref(eng.s), 32, buf.data_ptr(), None) == -1
    assert lib.lpx_tableau_block_commit(C.byref(eng.s), 33, buf.data_ptr(), 1, None) == -1
    assert lib.lpx_tableau_block_flush(C.byref(eng.s), 12, None) == -1
    assert lib.lpx_tableau_block_flush(C.byref(eng.s), 16, None) == 0  # a larger block than this slice's own default
