"""Worker of the column-partition tests (one process per rank, launched by torch.distributed.run).

mode "cpu": the host protocol (ProposalExchange + column_range) driven by a NumPy stand-in of the device kernels,
            gloo on CPU tensors -- checks message layout, winner selection and the replicated ratio test.
mode "gpu": the product path (PrimalTableauSimplexSolver with a process group) with every rank on cuda:0 and the
            gloo backend staging the all-gather through the host (2-rank rehearsal on the one-GPU box).
Both compare against the single-process oracle on the same problem.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import linprog_amd  # noqa: E402,F401
from linprog_amd import problems  # noqa: E402
from linprog_amd.tableau import ProposalExchange, column_range  # noqa: E402
from oracle import oracle as O  # noqa: E402

NOIDX = float(2 ** 62)


def numpy_rank(c, A, b, basis, rank, world, ex, maxiters):
    """Column-partitioned tableau simplex with NumPy standing in for lpx_tableau_propose/commit."""
    c32, A32, b32 = O.preprocess(c, A, b)
    m, n = A32.shape
    j0, j1 = column_range(n, rank, world)
    nl = j1 - j0
    T = np.zeros((m + 1, nl + 1))
    T[:m, :nl] = A32[:, j0:j1]
    T[:m, nl] = b32
    T[m, :nl] = c32[j0:j1]
    basis = np.array(basis, dtype=np.int64)
    log = []
    f32 = True
    for it in range(maxiters):
        rc = T[m, :nl]
        neg = np.flatnonzero(rc < 0)
        send = np.zeros(ex.msg)
        send[0] = rc.min() if nl else np.inf
        send[1] = j0 + neg[0] if neg.size else NOIDX
        if neg.size:
            send[2:2 + m + 1] = T[:, neg[0]]
        ex.sendbuf.copy_(torch.from_numpy(send))
        recv = ex.all_gather().numpy().reshape(world, ex.msg)
        mn = recv[:, 0].min()
        if mn >= 0 or abs(mn) <= 1e-8:
            return log, basis, T[:m, nl].copy(), True
        win = int(np.argmin(recv[:, 1]))
        j = int(recv[win, 1])
        d = recv[win, 2:2 + m + 1]
        r, theta, dmax = O.ratio_primal(T[:m, nl].copy(), d[:m].copy(), f32)
        e = (-d) / d[r]
        e[r] = 1.0 / d[r]
        log.append([r, j, int(basis[r])])
        basis[r] = j
        T = O.rank1_update(np.ascontiguousarray(T), r, e)
        if j0 <= j < j1:
            T[:, j - j0] = 0.0
            T[r, j - j0] = 1.0
        f32 = False
    return log, basis, T[:m, nl].copy(), False


def numpy_rank_deferred(c, A, b, basis, rank, world, ex, maxiters, kb=5):
    """The same protocol with DEFERRED updates (lpx_tableau_block_*): the slice T0 stays stale for up to kb pivots; the
    proposal's column and the commit's pivot row are formed from T0 plus the pending (r, e, pivot row) triples with the
    update's own arithmetic (the oracle's rank-1 helper on the thin piece), the reduced-cost row and x_B are kept current,
    and a "flush" applies the pending updates to T0 in order.  Same messages, same decisions, same bits as numpy_rank."""
    c32, A32, b32 = O.preprocess(c, A, b)
    m, n = A32.shape
    j0, j1 = column_range(n, rank, world)
    nl = j1 - j0
    T0 = np.zeros((m + 1, nl + 1))
    T0[:m, :nl] = A32[:, j0:j1]
    T0[:m, nl] = b32
    T0[m, :nl] = c32[j0:j1]
    basis = np.array(basis, dtype=np.int64)
    log, f32, pend = [], True, []  # pend: (r, e, old pivot row over the local columns + x_B, local rewritten column or -1)
    rc, xb = T0[m, :nl].copy(), T0[:m, nl].copy()

    def column(k):  # column k of the slice as the pending updates leave it
        d = T0[:, k:k + 1].copy()
        for r, e, row, kk in pend:
            if kk == k:
                d[:] = 0.0
                d[r] = 1.0
            else:
                d = O.rank1_update(np.ascontiguousarray(d), r, e)  # d[r] is the old pivot-row entry of this column
        return d[:, 0]

    def pivot_row(r):  # row r of the slice (and its x_B entry) as the pending updates leave it
        x = T0[r:r + 1, :].copy()
        for ru, e, row, kk in pend:
            if ru == r:
                x = O.rank1_update(np.ascontiguousarray(row[None, :].copy()), 0, np.array([e[ru]]))
            else:
                x = O.rank1_update(np.ascontiguousarray(np.vstack([x, row[None, :]])), 1, np.array([e[r], e[ru]]))[:1]
            if kk >= 0:
                x[0, kk] = 1.0 if ru == r else 0.0
        return x[0]

    def flush():
        nonlocal T0
        for r, e, row, kk in pend:
            T0 = O.rank1_update(np.ascontiguousarray(T0), r, e)
            if kk >= 0:
                T0[:, kk] = 0.0
                T0[r, kk] = 1.0
        pend.clear()

    for it in range(maxiters):
        neg = np.flatnonzero(rc < 0)
        send = np.zeros(ex.msg)
        send[0] = rc.min() if nl else np.inf
        send[1] = j0 + neg[0] if neg.size else NOIDX
        if neg.size:
            send[2:2 + m + 1] = column(int(neg[0]))
        ex.sendbuf.copy_(torch.from_numpy(send))
        recv = ex.all_gather().numpy().reshape(world, ex.msg)
        mn = recv[:, 0].min()
        if mn >= 0 or abs(mn) <= 1e-8:
            flush()
            assert np.array_equal(T0[m, :nl], rc) and np.array_equal(T0[:m, nl], xb)  # the eager pieces are the tableau's
            return log, basis, xb.copy(), True
        win = int(np.argmin(recv[:, 1]))
        j = int(recv[win, 1])
        d = recv[win, 2:2 + m + 1]
        r, theta, dmax = O.ratio_primal(xb.copy(), d[:m].copy(), f32)
        e = (-d) / d[r]
        e[r] = 1.0 / d[r]
        row = pivot_row(r)
        kk = j - j0 if j0 <= j < j1 else -1
        # eager pieces, with the update's own arithmetic: x_B (column nl) and the reduced-cost row (row m)
        xb = O.rank1_update(np.ascontiguousarray(np.append(xb, 0.0)[:, None]), r, e)[:m, 0]
        rc = O.rank1_update(np.ascontiguousarray(np.vstack([rc[None, :], row[None, :nl]])), 1, np.array([e[m], e[r]]))[0]
        if kk >= 0:
            rc[kk] = 0.0
        pend.append((r, e.copy(), row, kk))
        log.append([r, j, int(basis[r])])
        basis[r] = j
        f32 = False
        if len(pend) == kb:
            flush()
    flush()
    return log, basis, xb.copy(), False


def numpy_rank_dual(c, A, b, basis, rank, world, ex, maxiters):
    """Column-partitioned DUAL tableau simplex with NumPy standing in for lpx_tableau_dual_propose/commit."""
    c32, A32, b32 = O.preprocess(c, A, b)
    m, n = A32.shape
    j0, j1 = column_range(n, rank, world)
    nl = j1 - j0
    basis = np.array(basis, dtype=np.int64)
    inv32 = np.linalg.inv(A32[:, basis])
    T = np.zeros((m + 1, nl + 1))
    T[:m, :nl] = (inv32 @ A32)[:, j0:j1]
    T[:m, nl] = inv32 @ b32
    T[m, :nl] = (c32 - c32[basis] @ inv32 @ A32)[j0:j1]
    isbasic = np.zeros(n, dtype=bool)
    isbasic[basis] = True
    T[m, :nl][isbasic[j0:j1]] = 0.0
    log = []
    f32 = True
    for it in range(maxiters):
        xB = T[:m, nl]
        mn = xB.min()
        if mn >= 0 or abs(mn) <= 1e-8:
            return log, basis, xB.copy(), "optimal"
        r = int(np.flatnonzero(xB < 0)[0])  # replicated decision
        row = np.where(isbasic[j0:j1], 0.0, T[r, :nl])
        rc = T[m, :nl]
        with np.errstate(divide="ignore", invalid="ignore"):
            if f32:
                th = np.where(row < 0, ((-rc.astype(np.float32)) / row.astype(np.float32)).astype(np.float64), np.inf)
            else:
                th = np.where(row < 0, (-rc) / row, np.inf)
        send = np.zeros(ex.msg)
        k = int(np.argmin(th)) if nl else 0
        send[0] = th[k] if nl else np.inf
        send[1] = j0 + k
        if nl:
            send[2:2 + m + 1] = T[:, k]
        ex.sendbuf.copy_(torch.from_numpy(send))
        recv = ex.all_gather().numpy().reshape(world, ex.msg)
        if np.all(np.isinf(recv[:, 0])):
            return log, basis, xB.copy(), "dual_unbounded"
        win = min(range(world), key=lambda g: (recv[g, 0], recv[g, 1]))
        j = int(recv[win, 1])
        d = recv[win, 2:2 + m + 1]
        e = (-d) / d[r]
        e[r] = 1.0 / d[r]
        log.append([r, j, int(basis[r])])
        isbasic[basis[r]] = False
        isbasic[j] = True
        basis[r] = j
        T = O.rank1_update(np.ascontiguousarray(T), r, e)
        if j0 <= j < j1:
            T[:, j - j0] = 0.0
            T[r, j - j0] = 1.0
        f32 = False
    return log, basis, T[:m, nl].copy(), "running"


def numpy_rank_dual_ahead(c, A, b, basis, rank, world, ex, maxiters):
    """The same with the lookahead ordering of TableauEngine.enqueue (lpx_tableau_dual_commit_begin / _commit_update): the
    leaving row and the proposal of pivot t+1 are formed, and gathered into the OTHER slot, before pivot t's update is
    applied to the carried tableau; the decision is adopted when pivot t+1 begins."""
    c32, A32, b32 = O.preprocess(c, A, b)
    m, n = A32.shape
    j0, j1 = column_range(n, rank, world)
    nl = j1 - j0
    basis = np.array(basis, dtype=np.int64)
    inv32 = np.linalg.inv(A32[:, basis])
    T = np.zeros((m + 1, nl + 1))
    T[:m, :nl] = (inv32 @ A32)[:, j0:j1]
    T[:m, nl] = inv32 @ b32
    T[m, :nl] = (c32 - c32[basis] @ inv32 @ A32)[j0:j1]
    isbasic = np.zeros(n, dtype=bool)
    isbasic[basis] = True
    T[m, :nl][isbasic[j0:j1]] = 0.0

    def propose(Tv, f32):  # (optimal?, leaving row, message) from a tableau view
        xB = Tv[:m, nl]
        mn = xB.min()
        if mn >= 0 or abs(mn) <= 1e-8:
            return True, 0, np.zeros(ex.msg)
        r = int(np.flatnonzero(xB < 0)[0])
        row = np.where(isbasic[j0:j1], 0.0, Tv[r, :nl])
        rc = Tv[m, :nl]
        with np.errstate(divide="ignore", invalid="ignore"):
            if f32:
                th = np.where(row < 0, ((-rc.astype(np.float32)) / row.astype(np.float32)).astype(np.float64), np.inf)
            else:
                th = np.where(row < 0, (-rc) / row, np.inf)
        send = np.zeros(ex.msg)
        k = int(np.argmin(th)) if nl else 0
        send[0] = th[k] if nl else np.inf
        send[1] = j0 + k
        if nl:
            send[2:2 + m + 1] = Tv[:, k]
        return False, r, send

    log = []
    cur = 0
    opt, r, send = propose(T, True)
    ex.sendbufs[cur].copy_(torch.from_numpy(send))
    ex.all_gather(cur)
    pend = (opt, r)
    for it in range(maxiters):
        opt, r = pend  # adopted at the start of the pivot (k_td_commit)
        if opt:
            return log, basis, T[:m, nl].copy(), "optimal"
        recv = ex.recvbufs[cur].numpy().reshape(world, ex.msg)
        if np.all(np.isinf(recv[:, 0])):
            return log, basis, T[:m, nl].copy(), "dual_unbounded"
        win = min(range(world), key=lambda g: (recv[g, 0], recv[g, 1]))
        j = int(recv[win, 1])
        d = recv[win, 2:2 + m + 1].copy()
        e = (-d) / d[r]
        e[r] = 1.0 / d[r]
        log.append([r, j, int(basis[r])])
        isbasic[basis[r]] = False
        isbasic[j] = True
        basis[r] = j
        # what the update is about to write (the device forms only the thin pieces it needs of it: td_next_entry)
        Tn = O.rank1_update(np.ascontiguousarray(T), r, e)
        if j0 <= j < j1:
            Tn[:, j - j0] = 0.0
            Tn[r, j - j0] = 1.0
        nxt = cur ^ 1
        if it + 1 < maxiters:
            pend = propose(Tn, False)[:2]
            ex.sendbufs[nxt].copy_(torch.from_numpy(propose(Tn, False)[2]))
            ex.all_gather(nxt)  # travels while the update runs
        T = Tn  # lpx_tableau_commit_update
        cur = nxt
    return log, basis, T[:m, nl].copy(), "running"


def main_dual(mode, m, n, seed, maxiters):
    rank, world = dist.get_rank(), dist.get_world_size()
    c, A, b, basis = problems.gen_dual(m, n, seed)
    want = O.OracleTableau(c, A, b, basis, kind="dual", log_cap=maxiters + 1).solve(maxiters)
    assert len(want.log) > 3
    if mode in ("cpu-dual", "cpu-ahead-dual"):
        ex = ProposalExchange(dist.group.WORLD, m + 3 + ((m + 3) % 2), "cpu")
        model = numpy_rank_dual if mode == "cpu-dual" else numpy_rank_dual_ahead
        log, bas, xB, how = model(c, A, b, basis, rank, world, ex, maxiters)
        assert log == want.log.tolist(), f"rank {rank}: pivot sequence differs"
        assert bas.tolist() == want.basis.tolist() and (how == "optimal") == (want.status == O.OPTIMAL)
    else:
        from linprog_amd.tableau import DualTableauSimplexSolver

        torch.cuda.set_device(0)
        cls = type("L", (DualTableauSimplexSolver,), {"_LOG_CAP": maxiters + 1})
        s = cls(c, A, b, basis, group=dist.group.WORLD)
        res = s.solve(maxiters)
        assert s.pivot_log().tolist() == want.log.tolist(), f"rank {rank}: pivot sequence differs"
        assert res.basis.tolist() == want.basis.tolist() and (res.optimum is True) == (want.status == O.OPTIMAL)
        assert np.allclose(res.x, want.x, rtol=1e-9, atol=1e-9) and abs(res.cost - want.cost) <= 1e-9 * max(1, abs(want.cost))


def main():
    mode = sys.argv[1]
    m, n, seed, maxiters = (int(v) for v in sys.argv[2:6])
    # LPX_TEST_BACKEND=nccl (one rank per GPU: world size 1 on a one-GPU box): the library's own loop around RCCL
    # (lpx_tableau_dist_iterate) instead of the Python loop around torch.distributed that the gloo rehearsals run
    backend = os.environ.get("LPX_TEST_BACKEND", "gloo")
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if mode.endswith("-dual"):
        main_dual(mode, m, n, seed, maxiters)
        dist.barrier()
        dist.destroy_process_group()
        print(f"rank {rank} ok")
        return
    c, A, b, basis = problems.gen_primal(m, n, seed)
    want = O.OracleRevised(c, A, b, basis, log_cap=maxiters + 1).solve(maxiters)
    if mode in ("cpu", "cpu-deferred"):
        ex = ProposalExchange(dist.group.WORLD, m + 3 + ((m + 3) % 2), "cpu")
        log, bas, xB, opt = (numpy_rank if mode == "cpu" else numpy_rank_deferred)(c, A, b, basis, rank, world, ex, maxiters)
        assert log == want.log.tolist(), f"rank {rank}: pivot sequence differs"
        assert bas.tolist() == want.basis.tolist() and opt == (want.status == O.OPTIMAL)
    else:
        from linprog_amd.tableau import PrimalTableauSimplexSolver

        torch.cuda.set_device(0)
        if os.environ.get("LPX_TEST_RAGGED_KB"):
            # ranks whose own slices would pick DIFFERENT block sizes (a ragged partition at a size threshold): rank 0
            # alone sees deferred updates on; the engine must settle on one value for all ranks
            os.environ["LPX_TABLEAU_BLOCK_MIN_MB"] = "0" if rank == 0 else "1000000"
        cls = type("L", (PrimalTableauSimplexSolver,), {"_LOG_CAP": maxiters + 1})
        s = cls(c, A, b, basis, group=dist.group.WORLD)
        if os.environ.get("LPX_TEST_RAGGED_KB"):
            kbs = [None] * world
            dist.all_gather_object(kbs, s._engine.kb)
            assert len(set(kbs)) == 1 and kbs[0] == int(os.environ["LPX_TABLEAU_BLOCK"]), f"ranks disagree on the block: {kbs}"
        if os.environ.get("LPX_TEST_BACKEND") == "nccl":  # the library's own loop is what runs, pipelined where asked for
            import ctypes as C

            from linprog_amd import _lib

            assert s._engine.dist_comm is not None and s._engine.kb == int(os.environ["LPX_TABLEAU_BLOCK"])
            want_pipe = os.environ.get("LPX_TABLEAU_PIPELINE_MIN_MB") == "0" and cls.CONDENSED
            assert bool(_lib.load().lpx_tableau_pipelined(C.byref(s._engine.s))) == want_pipe
        res = s.solve(maxiters)
        assert s.pivot_log().tolist() == want.log.tolist(), f"rank {rank}: pivot sequence differs"
        assert res.basis.tolist() == want.basis.tolist() and res.optimum == (want.status == O.OPTIMAL)
        assert np.allclose(res.x, want.x, rtol=1e-9, atol=1e-9) and abs(res.cost - want.cost) <= 1e-9 * max(1, abs(want.cost))
        # a non-slack starting basis (float32 inverse start, primal_solvers.py:37-38): the nonbasic variables are then a
        # mix of structural and slack columns, partitioned by position in the sorted nonbasic list
        mid = O.OracleRevised(c, A, b, basis, log_cap=16).solve(9).basis.copy()
        # From a float32-inverted start the carried basic columns of the FULL layout are inv32 @ A[:, basis], an identity
        # only to float32 accuracy, and its pivots follow the oracle's tableau loop; the condensed layout carries no basic
        # columns -- they are exact unit vectors by construction, as in the reference, which re-derives everything from
        # B^-1 -- and follows the reference's (revised) sequence.  x agrees with the other form to float32 accuracy.
        Same, Other = (O.OracleRevised, O.OracleTableau) if cls.CONDENSED else (O.OracleTableau, O.OracleRevised)
        w3 = Same(c, A, b, mid, log_cap=maxiters + 1).solve(maxiters)
        s3 = cls(c, A, b, mid, group=dist.group.WORLD)
        r3 = s3.solve(maxiters)
        assert s3.pivot_log().tolist() == w3.log.tolist(), f"rank {rank}: pivot sequence differs (non-slack start)"
        assert r3.optimum == (w3.status == O.OPTIMAL) and r3.basis.tolist() == w3.basis.tolist()
        assert np.allclose(r3.x, w3.x, rtol=1e-5, atol=1e-5)
        w4 = Other(c, A, b, mid, log_cap=4).solve(maxiters)
        assert abs(r3.cost - w4.cost) <= 1e-5 * max(1.0, abs(w4.cost))
        # (condensed form, the default: only nonbasic columns are carried -- with the slack basis the n - m structural ones)
        ncarry = n - m if cls.CONDENSED else n
        Ab, bb, cb, basb = problems.gen_primal_columns(m, n, seed, *column_range(ncarry, rank, world))
        s2 = cls(cb, Ab, bb, basb, group=dist.group.WORLD, columns=column_range(ncarry, rank, world), n_global=n)
        r2 = s2.solve(maxiters)
        full = problems.gen_primal_columns(m, n, seed, 0, n)
        w2 = O.OracleTableau(full[2], full[0], full[1], full[3], log_cap=maxiters + 1).solve(maxiters)
        assert s2.pivot_log().tolist() == w2.log.tolist()
        assert np.allclose(r2.x, w2.x, rtol=1e-9, atol=1e-9)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
