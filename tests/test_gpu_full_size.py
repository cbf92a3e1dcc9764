"""GPU tests at BASELINE.json's full sizes, through size-independent properties (task rule 3).

config 4: m=20000, n=40000 tableau (6.4 GB): after K pivots the carried tableau must still satisfy
          T[:, basis] = I exactly, rc[basis] = 0 exactly, x_B >= 0, A x = b, cost = c.x, cost non-increasing,
          and the first pivots must equal the oracle's on the same workload.
config 3: 4096 LPs of 64 x 128: all optimal, A x = b, x >= 0, cost = c.x, reduced costs >= -1e-8 at the
          reported basis (optimality certificate computed independently in float64 on the host), plus a sample
          of LPs bit-identical to the oracle.
"""
import numpy as np
import pytest

from tests._util import gen

pytestmark = pytest.mark.gpu


def test_config4_tableau_invariants():
    import torch

    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    m, n, K = 20000, 40000, 24
    A, b, c, basis = gen.gen_primal_columns(m, n, 0, 0, n)
    cls = type("L", (PrimalTableauSimplexSolver,), {"_LOG_CAP": K + 1})
    s = cls(c, A, b, basis, columns=(0, n), n_global=n)
    costs = []
    for k in (8, 8, 8):
        res = s.solve(k)
        costs.append(float(res.cost))
        assert res.iters == k and res.optimum is False
    assert costs[0] >= costs[1] >= costs[2] and costs[2] < 0
    log = s.pivot_log()
    assert len(log) == K
    eng = s._engine
    T = eng.T
    if eng.slot_var is None:
        bas = torch.from_numpy(res.basis).cuda()
        # basic columns are exact unit vectors, basic reduced costs exactly 0
        cols = T[:, bas]                                   # (m+1) x m gather
        assert bool((cols[:m].diagonal() == 1.0).all())
        assert int((cols[:m] != 0).sum().item()) == m
        assert bool((cols[m] == 0.0).all())
    else:
        # condensed form (default): only nonbasic columns are carried; slots and basis partition the variables, and a
        # sample of carried columns equals B^-1 A_j recomputed in float64 on the host
        slot_var, var_slot = eng.slot_var.cpu().numpy(), eng.var_slot.cpu().numpy()
        assert eng.n == n - m and np.array_equal(np.sort(np.concatenate([slot_var, res.basis])), np.arange(n))
        assert np.array_equal(var_slot[slot_var], np.arange(eng.n)) and np.all(var_slot[res.basis] == -1)
        Bm = A[:, res.basis].astype(np.float64)
        for k in (0, 777, eng.n - 1):
            col = np.linalg.solve(Bm, A[:, slot_var[k]].astype(np.float64))
            assert np.abs(T[:m, k].cpu().numpy() - col).max() < 1e-8
    xB = T[:m, eng.n].cpu().numpy()
    assert xB.min() >= 0.0
    x = res.x
    assert np.array_equal(x[res.basis], xB) and np.count_nonzero(x) <= m
    # A x = b and cost = c.x against the ORIGINAL float32 inputs, in float64 on the host
    resid = A[:, np.flatnonzero(x)].astype(np.float64) @ x[np.flatnonzero(x)] - b.astype(np.float64)
    assert np.abs(resid).max() < 1e-9
    assert abs(float(np.dot(c.astype(np.float64), x)) - res.cost) <= 1e-9 * max(1.0, abs(res.cost))
    assert len(set(res.basis.tolist())) == m
    # the oracle takes the same first pivots on the same workload (OpenMP update: ~0.1 s per pivot)
    kc = 6
    o = O.OracleTableau(c, A, b, basis, log_cap=kc + 1, threads=16).solve(kc)
    assert o.log.tolist() == log[:kc].tolist()


def test_config4_headline_path_bit_exact_vs_oracle():
    """The instances that carry the headline, at the headline's size: one solve(96) -- three full blocks of 32 through the
    PIPELINED chain (decision kernels with the previous block's pairs in their chains, the pass in its hot instance
    k_tb_flush2<32, nt, 8, 256, 1, 1> writing the other buffer, the last one in place) -- then a ragged solve(40) (a full
    block and a short one: the twin instance).  The WHOLE pivot log against the 16-thread oracle (condensed layout), and
    the WHOLE tableau -- every carried column, x_B, the reduced-cost row -- bit for bit against the oracle's."""
    import ctypes as C

    from linprog_amd import _lib
    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    m, n = 20000, 40000
    A, b, c, basis = gen.gen_primal_columns(m, n, 0, 0, n)
    cls = type("L", (PrimalTableauSimplexSolver,), {"_LOG_CAP": 256})
    s = cls(c, A, b, basis, columns=(0, n), n_global=n)
    eng = s._engine
    lib = _lib.load()
    assert eng.slot_var is not None and lib.lpx_tableau_block_size(C.byref(eng.s)) == 32
    assert lib.lpx_tableau_pipelined(C.byref(eng.s)) == 1 and eng.T_alt is not None
    o = O.OracleTableau(c, A, b, basis, log_cap=256, threads=16, condensed=True)
    w = o.nslots + 1
    for k in (96, 40):
        res = s.solve(k)
        r = o.solve(k)
        assert res.iters == k == r.iters and res.optimum is False
        log = s.pivot_log()
        assert log.tolist() == o.log[:len(log)].tolist()
        assert np.array_equal(eng.slot_var.cpu().numpy(), o.slot_var) and res.basis.tolist() == r.basis.tolist()
        for r0 in range(0, m + 1, 4000):  # by row blocks: bounded temporaries
            assert np.array_equal(eng.T[r0:r0 + 4000, :w].cpu().numpy(), o.T[r0:r0 + 4000, :w]), f"rows {r0}.."
        assert np.array_equal(res.x, r.x)
    assert len(s.pivot_log()) == 136


@pytest.mark.parametrize("form", ["tableau", "revised"])
def test_config3_batch_full_size(form):
    from linprog_amd.batched import BatchedPrimalSolver, BatchedTableauSolver
    from oracle import oracle as O

    B = 4096
    probs = [gen.gen_primal(64, 128, s) for s in range(B)]
    c = np.stack([p[0] for p in probs]); A = np.stack([p[1] for p in probs])
    b = np.stack([p[2] for p in probs]); basis = np.stack([p[3] for p in probs])
    solver = (BatchedTableauSolver if form == "tableau" else BatchedPrimalSolver)(c, A, b, basis)
    res = solver.solve(100000)
    raw = solver.raw
    assert (raw["status"] == 1).all() and (raw["iters"] == raw["pivots"] + 1).all()
    A32 = A.astype(np.float32).astype(np.float64); b32 = b.astype(np.float32).astype(np.float64)
    c32 = c.astype(np.float32).astype(np.float64)
    x = raw["x"]
    assert x.min() >= -1e-12
    assert np.abs(np.einsum("bij,bj->bi", A32, x) - b32).max() < 1e-9
    assert np.abs(np.einsum("bj,bj->b", c32, x) - raw["cost"]).max() < 1e-9
    # independent optimality certificate: rc = c - c_B B^-1 A >= -1e-8 at the reported basis
    Bm = np.take_along_axis(A32, raw["basis"][:, None, :], axis=2)
    cB = np.take_along_axis(c32, raw["basis"], axis=1)
    y = np.linalg.solve(np.transpose(Bm, (0, 2, 1)), cB[..., None])[..., 0]
    rc = c32 - np.einsum("bi,bij->bj", y, A32)
    assert rc.min() >= -1e-8 - 1e-9
    Oracle = O.OracleTableau if form == "tableau" else O.OracleRevised
    for i in range(0, B, 293):  # a spread sample is bit-identical to the oracle of the same form
        o = Oracle(*probs[i], log_cap=4).solve(100000)
        assert np.array_equal(res[i].x, o.x) and res[i].cost == o.cost and res[i].iters == o.iters


def test_tableau_beyond_32bit_element_indices():
    """A tableau of 4.6e9 doubles (36.9 GB, built on the device): element offsets exceed 2^32, so every index in K1 and
    in the gather / copy kernels must be 64-bit.  Each pivot's decision and the updated tableau are checked against a
    torch restatement of the same pivot kept in a second buffer (decisions exact, values to 1e-12: torch's addcmul is
    not a fused multiply-add)."""
    import torch

    from linprog_amd import _lib
    from linprog_amd.tableau import TableauEngine

    m, n = 48000, 96000
    ld = n + 2
    assert (m + 1) * ld > 2 ** 32
    g = torch.Generator(device="cuda").manual_seed(7)
    T = torch.zeros((m + 1, ld), dtype=torch.float64, device="cuda")
    k = n - m
    for r0 in range(0, m, 4000):  # G ~ U(0,1) in float32-valued doubles, by row blocks
        blk = torch.rand((min(4000, m - r0), k), generator=g, device="cuda", dtype=torch.float32)
        T[r0:r0 + blk.shape[0], :k] = blk.to(torch.float64)
    T[:m, k:n] = torch.eye(m, dtype=torch.float64, device="cuda")
    T[:m, n] = 1.0 + torch.rand(m, generator=g, device="cuda", dtype=torch.float32).to(torch.float64)
    c = torch.zeros(n, dtype=torch.float32, device="cuda")
    c[:k] = -torch.rand(k, generator=g, device="cuda", dtype=torch.float32)
    T[m, :n] = c.to(torch.float64)
    basis = torch.arange(k, n, dtype=torch.int64, device="cuda")
    vs = torch.full((n,), _lib.VS_LOWER, dtype=torch.int32, device="cuda")
    vs[k:] = _lib.VS_BASIC
    ref = T.clone()
    eng = TableauEngine(T, m, n, basis, vs, c, log_cap=16)
    pivots = 3
    ctrl = eng.run(pivots)
    assert ctrl.status == _lib.RUNNING and ctrl.pivots == pivots
    log = eng.host_log(pivots)
    f32 = True
    for it in range(pivots):
        rc = ref[m, :n]
        j = int(torch.nonzero(rc < 0)[0])
        d = ref[:, j].clone()
        xB = ref[:m, n]
        pos = d[:m] > 0
        if f32:
            th = torch.where(pos, (xB.to(torch.float32) / d[:m].to(torch.float32)).to(torch.float64), torch.inf)
        else:
            th = torch.where(pos, xB / d[:m], torch.inf)
        r = int(torch.argmin(th))
        assert [r, j] == log[it][:2].tolist(), f"pivot {it}"
        e = -d / d[r]
        e[r] = 1.0 / d[r]
        rowr = ref[r].clone()
        for r0 in range(0, m + 1, 8000):  # row blocks keep the temporaries small
            r1 = min(m + 1, r0 + 8000)
            ref[r0:r1].addcmul_(e[r0:r1, None], rowr[None, :])
        ref[r] = e[r] * rowr
        ref[:, j] = 0.0
        ref[r, j] = 1.0
        f32 = False
    # compare by row blocks (a full-size temporary would be another 37 GB)
    worst = 0.0
    for r0 in range(0, m + 1, 8000):
        r1 = min(m + 1, r0 + 8000)
        diff = (T[r0:r1, :n + 1] - ref[r0:r1, :n + 1]).abs() / ref[r0:r1, :n + 1].abs().clamp_min(1.0)
        worst = max(worst, float(diff.max()))
    assert worst <= 1e-12, worst
    # the last rows / columns (offsets > 2^32) were really written
    assert float((T[m, :n] - ref[m, :n]).abs().max()) <= 1e-12 and bool((T[:, log[-1][1]] != 0).sum() == 1)


def test_revised_form_10000x20000_invariants():
    """The revised loop at the size where its large-matrix paths engage (non-temporal streams, K1 fused with the next y
    pass, staged pricing): after 60 pivots B^-1 is the inverse of the current basis matrix, x_B = B^-1 b >= 0, the cost
    never rose, A x = b; the tableau-form engine takes the same 60 pivots on the same data and the oracle the same
    first 6 (m = 10000: a quarter of a second per oracle pivot)."""
    import torch

    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver
    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O

    m, n, K = 10000, 20000, 60
    A, b, c, basis = gen.gen_primal_columns(m, n, 0, 0, n)
    cls = type("L", (PrimalRevisedSimplexSolver,), {"_LOG_CAP": K + 1})
    s = cls(c, A, b, basis)
    costs = []
    for k in (20, 20, 20):
        res = s.solve(k)
        costs.append(float(res.cost))
        assert res.iters == k and res.optimum is False
    assert costs[0] >= costs[1] >= costs[2] and costs[2] < 0
    log = s.pivot_log()
    assert len(log) == K
    eng = s._engine
    bas = torch.from_numpy(res.basis).cuda()
    A_dev = torch.from_numpy(A).cuda()
    Bm = A_dev[:, bas].to(torch.float64)
    Binv = eng.Binv[:, :m]
    resid = Bm @ Binv
    resid.diagonal().sub_(1.0)
    assert float(resid.abs().max()) < 1e-10
    xB = eng.xB[:m]
    want = Binv @ torch.from_numpy(b.astype(np.float64)).cuda()
    assert float((xB - want).abs().max()) < 1e-9 and float(xB.min()) >= 0.0
    x = res.x
    nz = np.flatnonzero(x)
    assert np.abs(A[:, nz].astype(np.float64) @ x[nz] - b.astype(np.float64)).max() < 1e-9
    assert abs(float(np.dot(c.astype(np.float64), x)) - res.cost) <= 1e-9 * max(1.0, abs(res.cost))
    del resid, Bm, A_dev
    tcls = type("LT", (PrimalTableauSimplexSolver,), {"_LOG_CAP": K + 1})
    t = tcls(c, A, b, basis, columns=(0, n), n_global=n)
    rt = t.solve(K)
    assert t.pivot_log().tolist() == log.tolist()
    assert abs(rt.cost - res.cost) <= 1e-9 * max(1.0, abs(res.cost))
    o = O.OracleRevised(c, A, b, basis, log_cap=8).solve(6)
    assert o.log.tolist() == log[:6].tolist()
