"""GPU parity of the individually exported kernels (through the C ABI) against the oracle.

Bar: bit-exact for the elementwise rank-1 update and the eta column (same fma rule as the oracle), exact
indices and exact extreme values for every (value, index) reduction.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    from linprog_amd import _lib
    from oracle import oracle as O

    L = _lib.load()
    scratch = torch.zeros(int(L.lpx_reduce_scratch_bytes()), dtype=torch.uint8, device="cuda")
    return dict(torch=torch, L=L, lib=_lib, O=O, scratch=scratch)


def dev(env, a):
    return env["torch"].from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("rows,cols,pad", [(1, 1, 1), (3, 9, 1), (64, 64, 0), (65, 129, 1), (257, 1000, 0),
                                           (1001, 2001, 1), (2500, 777, 3)])
def test_pivot_update_bit_exact(env, rows, cols, pad):
    torch, L, O = env["torch"], env["L"], env["O"]
    rng = np.random.default_rng(rows * 1000 + cols)
    ld = cols + pad + ((cols + pad) % 2)
    M = np.zeros((rows, ld))
    M[:, :cols] = rng.standard_normal((rows, cols))
    d = rng.standard_normal(rows)
    r = int(rng.integers(rows))
    e = O.eta(d, r)
    want = O.rank1_update(M.copy(), r, e, cols=cols)
    Md, ed = dev(env, M), dev(env, e)
    scr = torch.zeros(ld + 2, dtype=torch.float64, device="cuda")
    env["lib"].check(L.lpx_pivot_update(Md.data_ptr(), rows, cols, ld, r, ed.data_ptr(), scr.data_ptr(), None))
    got = Md.cpu().numpy()
    assert np.array_equal(got[:, :cols], want[:, :cols])
    # eta column itself
    dd, out = dev(env, d), torch.zeros(rows, dtype=torch.float64, device="cuda")
    env["lib"].check(L.lpx_eta(dd.data_ptr(), rows, r, out.data_ptr(), None))
    assert np.array_equal(out.cpu().numpy(), e)


def test_pivot_update_rejects_bad_arguments(env):
    L, torch = env["L"], env["torch"]
    M = torch.zeros((4, 4), dtype=torch.float64, device="cuda")
    e = torch.zeros(4, dtype=torch.float64, device="cuda")
    assert L.lpx_pivot_update(M.data_ptr(), 4, 4, 3, 0, e.data_ptr(), e.data_ptr(), None) == -1  # ld < cols
    assert L.lpx_pivot_update(M.data_ptr(), 4, 4, 4, 7, e.data_ptr(), e.data_ptr(), None) == -1  # r out of range
    assert b"lpx_pivot_update" in L.lpx_last_error()


def _outs(env):
    t = env["torch"]
    return t.zeros(2, dtype=t.float64, device="cuda"), t.zeros(2, dtype=t.int64, device="cuda")


@pytest.mark.parametrize("n", [1, 5, 64, 257, 4097, 100003])
def test_reductions_match_oracle(env, n):
    L, O, lib = env["L"], env["O"], env["lib"]
    rng = np.random.default_rng(n)
    sp = env["scratch"].data_ptr()
    for trial in range(4):
        rc = rng.standard_normal(n)
        if trial == 1:
            rc = np.abs(rc)  # optimal: no negative entry
        if trial == 2:
            rc = np.round(rc, 1)  # many exact ties
        if trial == 3:
            rc[rng.integers(n)] = -1e-9  # inside the isclose band only
            rc = np.where(rc < 0, rc, np.abs(rc))
        vs = rng.integers(0, 4, n).astype(np.int32) if trial % 2 else None
        ov, oi = _outs(env)
        # keep every device tensor referenced while kernels use it (temporaries would be recycled)
        rc_d = dev(env, rc)
        vs_d = None if vs is None else dev(env, vs)
        vs_p = None if vs is None else vs_d.data_ptr()
        lib.check(L.lpx_price_first_negative(rc_d.data_ptr(), vs_p, n, ov.data_ptr(), oi.data_ptr(), sp, None))
        opt, mn, j = O.first_negative(rc, vs)
        assert ov.cpu().numpy()[0] == mn and oi.cpu().numpy()[0] == j
        lib.check(L.lpx_price_argmax(rc_d.data_ptr(), n, ov.data_ptr(), oi.data_ptr(), sp, None))
        _, mx, j = O.first_argmax(rc)
        assert ov.cpu().numpy()[0] == mx and oi.cpu().numpy()[0] == j
        # ratio tests (float64 and the float32 first-iteration variant)
        x = np.abs(rng.standard_normal(n)).astype(np.float32).astype(np.float64)
        d = np.round(rng.standard_normal(n), 2 if trial == 2 else 6).astype(np.float32).astype(np.float64)
        x_d, d_d = dev(env, x), dev(env, d)
        for f32 in (0, 1):
            lib.check(L.lpx_ratio_primal(x_d.data_ptr(), d_d.data_ptr(), n, f32, ov.data_ptr(), oi.data_ptr(), sp, None))
            r, th, dm = O.ratio_primal(x, d, f32)
            v = ov.cpu().numpy()
            assert oi.cpu().numpy()[0] == r and v[0] == th and v[1] == dm
            lib.check(L.lpx_ratio_dual(x_d.data_ptr(), d_d.data_ptr(), vs_p, n, f32, ov.data_ptr(), oi.data_ptr(), sp, None))
            j, th, rm = O.ratio_dual(x, d, f32, vs)
            v = ov.cpu().numpy()
            if vs is None or np.any(vs != 3):
                assert oi.cpu().numpy()[0] == j and v[0] == th and v[1] == rm
        lbB = x - np.abs(rng.standard_normal(n))
        ubB = x + np.abs(rng.standard_normal(n))
        lb_d, ub_d = dev(env, lbB), dev(env, ubB)
        for sign in (1.0, -1.0):
            lib.check(L.lpx_ratio_bounded(x_d.data_ptr(), d_d.data_ptr(), lb_d.data_ptr(), ub_d.data_ptr(), n, sign,
                                          ov.data_ptr(), oi.data_ptr(), sp, None))
            g1, r1, g2, r2 = O.ratio_bounded(x, d, lbB, ubB, sign)
            assert ov.cpu().numpy().tolist() == [g1, g2] and oi.cpu().numpy().tolist() == [r1, r2]
        lib.check(L.lpx_dual_row_select(rc_d.data_ptr(), n, ov.data_ptr(), oi.data_ptr(), sp, None))
        opt, mn, j = O.first_negative(rc)
        assert ov.cpu().numpy()[0] == mn and oi.cpu().numpy()[0] == j


def test_float32_ratio_tie_breaks_like_reference(env):
    """Two float64 quotients that round to the same float32 must tie (lowest index wins) in the f32 phase."""
    L, lib = env["L"], env["lib"]
    x = np.array([1.0 + 2.0 ** -30, 1.0])
    d = np.array([3.0, 3.0])
    ov, oi = _outs(env)
    sp = env["scratch"].data_ptr()
    x_d, d_d = dev(env, x), dev(env, d)
    lib.check(L.lpx_ratio_primal(x_d.data_ptr(), d_d.data_ptr(), 2, 1, ov.data_ptr(), oi.data_ptr(), sp, None))
    assert oi.cpu().numpy()[0] == 0
    lib.check(L.lpx_ratio_primal(x_d.data_ptr(), d_d.data_ptr(), 2, 0, ov.data_ptr(), oi.data_ptr(), sp, None))
    assert oi.cpu().numpy()[0] == 1


@pytest.mark.parametrize("transa", [0, 1])
@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (100, 70, 33), (129, 257, 65), (1000, 1000, 1000), (3000, 3000, 1500)])
def test_dgemm_mfma_vs_numpy(env, M, N, K, transa):
    """lpx_dgemm (fp64 MFMA, both tile sizes, ragged edges, op(A) = A or A^T, beta*C and gamma*I terms) against numpy.
    Floating point: |err| <= 1e-13 * K * max|a||b| (summation order differs from BLAS)."""
    lib, torch = env["L"], env["torch"]
    rng = np.random.default_rng(M * 7 + N * 3 + K + transa)
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((K, N))
    Cm = rng.standard_normal((M, N))
    alpha, beta, gamma = -1.25, 0.5, 2.0
    want = alpha * (A @ B) + beta * Cm + gamma * np.eye(M, N)
    As = np.ascontiguousarray(A.T) if transa else A
    dA, dB, dC = (torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (As, B, Cm))
    dD = torch.zeros((M, N), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.lpx_dgemm(transa, M, N, K, alpha, dA.data_ptr(), As.shape[1], dB.data_ptr(), N, beta, dC.data_ptr(), N, gamma,
                       dD.data_ptr(), N, st)
    assert rc == 0
    got = dD.cpu().numpy()
    assert np.max(np.abs(got - want)) <= 1e-13 * K * 16
    # C = NULL path
    rc = lib.lpx_dgemm(transa, M, N, K, 1.0, dA.data_ptr(), As.shape[1], dB.data_ptr(), N, 0.0, None, 0, 0.0, dD.data_ptr(), N, st)
    assert rc == 0
    assert np.max(np.abs(dD.cpu().numpy() - A @ B)) <= 1e-13 * K * 16


def _dinverse(env, A, ldi_pad=0):
    lib, torch = env["L"], env["torch"]
    m = A.shape[0]
    dA = torch.from_numpy(np.ascontiguousarray(A)).cuda()
    out = torch.zeros((m, m + ldi_pad), dtype=torch.float64, device="cuda")
    nbytes = lib.lpx_invert_workspace_bytes(m)
    work = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    info = torch.full((1,), -7, dtype=torch.int64, device="cuda")
    rc = lib.lpx_dinverse(m, dA.data_ptr(), m, out.data_ptr(), m + ldi_pad, work.data_ptr(), nbytes, info.data_ptr(),
                          torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return out.cpu().numpy()[:, :m], int(info.item())


@pytest.mark.parametrize("m", [1, 2, 5, 63, 64, 65, 128, 130, 200, 1000, 2100])
def test_dinverse_blocked_gauss_jordan_vs_numpy(env, m):
    """lpx_dinverse (blocked Gauss-Jordan, partial pivoting, MFMA trailing updates) against np.linalg.inv: block edges at
    64, ragged last block, strided output.  Floating point: residual |A X - I| <= 1e-13 * m * cond-ish factor and
    |X - inv(A)| <= 1e-10 * max|inv(A)| for these well-conditioned matrices (tolerances stated here)."""
    rng = np.random.default_rng(100 + m)
    A = rng.standard_normal((m, m)) + 0.5 * np.sqrt(m) * np.eye(m)
    X, info = _dinverse(env, A, ldi_pad=3)
    assert info == 0
    want = np.linalg.inv(A)
    assert np.max(np.abs(A @ X - np.eye(m))) <= 1e-12 * max(1, m)
    assert np.max(np.abs(X - want)) <= 1e-10 * np.max(np.abs(want))


def test_dinverse_needs_its_row_swaps(env):
    """A permuted, scaled identity plus noise: every pivot is off the diagonal, several across block boundaries."""
    m = 150
    rng = np.random.default_rng(7)
    perm = rng.permutation(m)
    A = 1e-3 * rng.standard_normal((m, m))
    A[np.arange(m), perm] += rng.uniform(1.0, 2.0, m) * rng.choice([-1.0, 1.0], m)
    X, info = _dinverse(env, A)
    assert info == 0
    assert np.max(np.abs(A @ X - np.eye(m))) <= 1e-12
    assert np.max(np.abs(X - np.linalg.inv(A))) <= 1e-11


def test_dinverse_flags_a_singular_matrix(env):
    """A zero column 70: info = 1 + 70, as LAPACK's getrf reports the first zero pivot.  A repeated column leaves
    rounding dust instead of zeros: flagged by the m * eps * max|A| threshold.  Nothing faults either way."""
    m = 100
    rng = np.random.default_rng(8)
    A = rng.standard_normal((m, m))
    A[:, 70] = 0.0
    _, info = _dinverse(env, A)
    assert info == 71
    A[:, 70] = A[:, 3]
    _, info = _dinverse(env, A)
    assert info == 71


def test_plain_c_client_of_the_abi():
    """examples/c_abi_demo (built by __graft_entry__.build() with gcc against include/lpx.h and liblpx.so): the C ABI used
    without Python or PyTorch -- one LP through lpx_revised_iterate, four through lpx_batched_primal_solve_ineq."""
    import os
    import subprocess

    from tests._util import ROOT

    exe = os.path.join(ROOT, "examples", "c_abi_demo")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "c abi demo ok" in r.stdout, r.stdout + r.stderr


def test_graph_capture_failure_unwinds_and_the_next_call_works(env, monkeypatch):
    """lpx_graph.hpp: an error inside a captured chain is returned (LPX_E_ARG) with the capture ended and the graph
    destroyed, and a failing capture / instantiation falls back to eager launches; in both cases the per-device capture
    stream is usable again, so the next call captures and replays as usual.  LPX_GRAPH_TEST_FAIL is the test hook."""
    import ctypes as C

    from linprog_amd.tableau import PrimalTableauSimplexSolver
    from oracle import oracle as O
    from tests._util import gen

    lib, L = env["lib"], env["L"]
    c, A, b, basis = gen.gen_primal(40, 100, 3)
    want = O.OracleTableau(c, A, b, basis, condensed=True, log_cap=512)
    s = type("L", (PrimalTableauSimplexSolver,), {"_LOG_CAP": 512})(c, A, b, basis)
    eng = s._engine
    st = env["torch"].cuda.current_stream().cuda_stream
    lib.check(L.lpx_ctrl_begin_solve(eng.s.ctrl, st))
    monkeypatch.setenv("LPX_GRAPH_UNROLL", "6")  # a key of its own: nothing cached from other tests
    monkeypatch.setenv("LPX_TABLEAU_SMALL_CHAIN", "0")  # the graph-captured chain (one launch per pivot), not one launch per 8 pivots
    monkeypatch.setenv("LPX_GRAPH_TEST_FAIL", "arg")
    assert L.lpx_tableau_iterate(C.byref(eng.s), 12, st) == -1  # LPX_E_ARG out of the captured chain
    assert b"captured chain" in L.lpx_last_error()
    env["torch"].cuda.synchronize()
    done = int(eng.read_ctrl().pivots)  # whatever ran before the failure is a valid prefix of the solve
    monkeypatch.setenv("LPX_GRAPH_TEST_FAIL", "hip")
    lib.check(L.lpx_tableau_iterate(C.byref(eng.s), 12, st))  # capture "fails": eager fallback, all 12 iterations run
    assert int(eng.read_ctrl().pivots) == done + 12
    monkeypatch.delenv("LPX_GRAPH_TEST_FAIL")
    lib.check(L.lpx_tableau_iterate(C.byref(eng.s), 12, st))  # captures and replays
    total = done + 24
    assert int(eng.read_ctrl().pivots) == total
    want.solve(total)
    assert s.pivot_log().tolist() == want.log[:total].tolist()


def test_selfcheck_resolves_every_kernel_instance():
    """lpx_selfcheck: every kernel handle of liblpx.so resolves to device code on this GPU (hipFuncGetAttributes); the list
    is the library's own (generated from the host stubs of its objects), so it cannot drift from the launchers."""
    from linprog_amd import _lib

    n = _lib.selfcheck()
    assert n >= 250, n
    assert _lib.selfcheck() == n  # repeatable, nothing left behind
