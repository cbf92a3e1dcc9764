"""GPU parity of the chain for launch-bound tableaus (lpx_tableau_small.hip: one launch per block of 8 pivots, one
workgroup decides the block while the others apply the previous one; BASELINE config 2).  Checked against the oracle's
loop (pivot log, carried tableau, basis, column maps: bit-identical) and against the per-pivot chain of the same library
on the same inputs (the whole control block as well)."""
import ctypes as C

import numpy as np
import pytest

from tests._util import assert_close, gen

pytestmark = pytest.mark.gpu


def logged(cls, cap):
    return type(cls.__name__ + "Logged", (cls,), {"_LOG_CAP": cap})


def chain_kind(solver):
    from linprog_amd import _lib

    return int(_lib.load().lpx_tableau_chain_kind(C.byref(solver._engine.s)))


CTRL_FIELDS = ("status", "f32_phase", "counter", "pivots", "enter_j", "leave_r", "leave_var", "price_val", "ratio_val",
               "aux_val", "flip", "enter_slot")


def ctrl_of(solver):
    c = solver._engine.read_ctrl()
    return {f: getattr(c, f) for f in CTRL_FIELDS}


@pytest.mark.parametrize("pricing", ["first_negative", "dantzig"])
@pytest.mark.parametrize("m,n,condensed,seed", [(300, 700, True, 5), (300, 700, False, 5), (530, 1061, True, 6), (1000, 2000, True, 0),
                                                (1023, 2045, True, 2), (7, 20, True, 1), (64, 1020, False, 3), (40, 1063, True, 4)])
def test_small_chain_bit_exact(m, n, condensed, seed, pricing, monkeypatch):
    """Shapes up to the chain's limits (1024 rows with the cost row, 1024 carried columns with x_B: 40 x 1063 carries 1023), both layouts, both
    pricing rules, calls that are no multiple of the block, end on either buffer and follow one another."""
    from linprog_amd import _lib
    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    c, A, b, basis = gen.gen_primal(m, n, seed)
    s = logged(PrimalTableauSimplexSolver, 512)(c, A, b, basis, condensed=condensed, pricing=pricing)
    assert chain_kind(s) == _lib.CHAIN_SMALL
    monkeypatch.setenv("LPX_TABLEAU_SMALL_CHAIN", "0")
    monkeypatch.setenv("LPX_TABLEAU_FUSED", "0")
    e = logged(PrimalTableauSimplexSolver, 512)(c, A, b, basis, condensed=condensed, pricing=pricing)
    assert chain_kind(e) == _lib.CHAIN_EAGER
    o = O.OracleTableau(c, A, b, basis, condensed=condensed, log_cap=512, pricing={"first_negative": 0, "dantzig": 1}[pricing])
    done = 0
    for k in (1, 2, 5, 64, 33, 8, 16, 9):
        monkeypatch.setenv("LPX_TABLEAU_SMALL_CHAIN", "1")
        monkeypatch.setenv("LPX_TABLEAU_FUSED", "1")
        res = s.solve(k)
        monkeypatch.setenv("LPX_TABLEAU_SMALL_CHAIN", "0")
        monkeypatch.setenv("LPX_TABLEAU_FUSED", "0")
        ref = e.solve(k)
        r = o.solve(k)
        done += k
        log = s.pivot_log().tolist()  # (the smallest case reaches its optimum inside the third call)
        assert len(log) == ctrl_of(s)["pivots"] <= done and log == o.log[:len(log)].tolist() and log == e.pivot_log().tolist()
        w = o.nslots + 1
        T = s._engine.T.cpu().numpy()
        assert np.array_equal(T[:, :w], o.T[:, :w])
        assert np.array_equal(T[:, :w], e._engine.T.cpu().numpy()[:, :w])
        assert res.basis.tolist() == r.basis.tolist() and res.iters == r.iters and np.array_equal(res.x, r.x)
        assert res.iters == ref.iters and res.optimum == ref.optimum
        assert_close(res.cost, r.cost, "cost")  # dot(c, x): the device sums it as a tree, the oracle in index order
        assert ctrl_of(s) == ctrl_of(e)
        if condensed:
            assert np.array_equal(s._engine.slot_var.cpu().numpy(), o.slot_var)
            assert np.array_equal(s._engine.var_slot.cpu().numpy(), e._engine.var_slot.cpu().numpy())
        else:
            assert np.array_equal(s._engine.var_state.cpu().numpy(), e._engine.var_state.cpu().numpy())


@pytest.mark.parametrize("m,n,seed", [(128, 256, 0), (200, 520, 2), (40, 100, 3)])
def test_small_chain_ends_inside_a_block_and_stays_there(m, n, seed, monkeypatch):
    """The loop ends (optimum) in the middle of a block and of a call: the launches that follow do nothing, the control
    block equals the per-pivot chain's, and a second solve() on the optimal state prices once and moves nothing."""
    from linprog_amd import _lib
    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    c, A, b, basis = gen.gen_primal(m, n, seed)
    s = logged(PrimalTableauSimplexSolver, 4096)(c, A, b, basis)
    assert chain_kind(s) == _lib.CHAIN_SMALL
    res = s.solve(100000)
    want = O.OracleTableau(c, A, b, basis, condensed=True, log_cap=4096).solve(100000)
    assert res.optimum and res.iters == want.iters and s.pivot_log().tolist() == want.log.tolist()
    assert np.array_equal(res.x, want.x)
    got = ctrl_of(s)
    monkeypatch.setenv("LPX_TABLEAU_SMALL_CHAIN", "0")
    monkeypatch.setenv("LPX_TABLEAU_FUSED", "0")
    e = logged(PrimalTableauSimplexSolver, 4096)(c, A, b, basis)
    e.solve(100000)
    assert got == ctrl_of(e)
    assert np.array_equal(s._engine.T.cpu().numpy(), e._engine.T.cpu().numpy())
    monkeypatch.setenv("LPX_TABLEAU_SMALL_CHAIN", "1")
    T0 = s._engine.T.clone()
    again = s.solve(100)
    assert again.optimum and again.iters == 1 and bool((s._engine.T == T0).all())


def test_small_chain_unbounded_inside_a_block(monkeypatch):
    """An unbounded direction met after some pivots: same exception, same control block as the per-pivot chain."""
    from linprog_amd import exceptions
    from linprog_amd.tableau import PrimalTableauSimplexSolver

    rng = np.random.default_rng(7)
    m, n = 40, 120
    c, A, b, basis = gen.gen_primal(m, n, 4)
    j = int(np.setdiff1d(np.arange(n), basis)[-1])  # a late nonbasic column (Bland meets it after other pivots): a free ray
    A[:, j] = -np.abs(rng.standard_normal(m))
    c[j] = -1.0
    ctrls = []
    for small in ("1", "0"):
        monkeypatch.setenv("LPX_TABLEAU_SMALL_CHAIN", small)
        monkeypatch.setenv("LPX_TABLEAU_FUSED", small)
        s = logged(PrimalTableauSimplexSolver, 512)(c, A, b, basis)
        with pytest.raises(exceptions.PrimalIsUnboundedError):
            s.solve(10000)
        ctrls.append((ctrl_of(s), s.pivot_log().tolist(), s._engine.T.cpu().numpy().copy()))
    assert ctrls[0][0] == ctrls[1][0] and ctrls[0][1] == ctrls[1][1] and np.array_equal(ctrls[0][2], ctrls[1][2])
    assert ctrls[0][0]["pivots"] > 0


def test_small_chain_limits():
    """One row or one carried column beyond what a workgroup holds: the library takes another chain."""
    from linprog_amd import _lib
    from linprog_amd.tableau import PrimalTableauSimplexSolver

    c, A, b, basis = gen.gen_primal(1024, 2050, 1)
    assert chain_kind(PrimalTableauSimplexSolver(c, A, b, basis)) != _lib.CHAIN_SMALL
    c, A, b, basis = gen.gen_primal(64, 1088, 1)  # 1024 carried columns: one too many
    assert chain_kind(PrimalTableauSimplexSolver(c, A, b, basis)) != _lib.CHAIN_SMALL
