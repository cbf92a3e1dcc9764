#!/usr/bin/env python3
"""Random shapes through the carried-tableau solvers (primal and dual rules, condensed and full layout, the chain of one
launch per 8 pivots, the one-launch chain, the in-place chains and the deferred-update chains with blocks of 8 / 16 / 32 --
in place and pipelined --, solve() in pieces) against the oracle's tableau loops: pivot log, x, iteration counts, status.
GPU box: python3 tools/fuzz_tableau.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from linprog_amd import exceptions, problems
    from linprog_amd.tableau import DualTableauSimplexSolver, PrimalTableauSimplexSolver
    from oracle import oracle as O

    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for it in range(cases):
        m = int(rng.integers(1, 600))
        n = m + int(rng.integers(1, 600))
        kind = "dual" if rng.random() < 0.5 else "primal"
        condensed = bool(rng.random() < 0.6)
        fused = bool(rng.random() < 0.7)
        os.environ["LPX_TABLEAU_FUSED"] = "1" if fused else "0"
        small = bool(fused and rng.random() < 0.6)  # one launch per 8 pivots (primal rules, shapes that fit one workgroup)
        os.environ["LPX_TABLEAU_SMALL_CHAIN"] = "1" if small else "0"
        block = int(rng.choice([0, 8, 16, 32], p=[0.4, 0.2, 0.2, 0.2]))  # > 0: deferred updates, forced on at any size
        os.environ["LPX_TABLEAU_BLOCK"] = str(block)
        os.environ["LPX_TABLEAU_BLOCK_MIN_MB"] = "0" if block else "1000000"
        pipe = bool(block and rng.random() < 0.5)  # the pass beside the next block's decisions (second buffer forced on)
        os.environ["LPX_TABLEAU_PIPELINE_MIN_MB"] = "0" if pipe else "-1"
        if block:
            os.environ["LPX_TB_ROWS"] = str(int(rng.choice([0, 1, 7, 64, 200])))
        seed = 1000 + it
        if kind == "primal":
            c, A, b, basis = (problems.gen_primal_degenerate if rng.random() < 0.2 else problems.gen_primal)(m, n, seed)
            if rng.random() < 0.15:
                A = A.copy(); A[:, : n - m] *= -1  # unbounded
            cls = PrimalTableauSimplexSolver
        else:
            c, A, b, basis = problems.gen_dual(m, n, seed)
            if rng.random() < 0.15:
                A = A.copy(); A[:, : n - m] = np.abs(A[:, : n - m])  # rows with b < 0 cannot be repaired: dual unbounded
            cls = DualTableauSimplexSolver
        pieces = [int(v) for v in rng.choice([1, 2, 3, 17, 32, 40, 65], size=3)] + [100000]
        ok, note = True, ""
        try:
            s = type("L", (cls,), {"_LOG_CAP": 20000})(c, A, b, basis, condensed=condensed)
            o = O.OracleTableau(c, A, b, basis, kind=kind, log_cap=20000)
        except Exception as e:  # constructor errors must agree
            try:
                O.OracleTableau(c, A, b, basis, kind=kind, log_cap=8)
                ok, note = False, f"GPU constructor raised {type(e).__name__}, oracle did not"
            except Exception:
                note = f"both constructors raise ({type(e).__name__})"
            print(f"{it:3d}: {kind:6s} {m:3d}x{n:4d} cond={int(condensed)} fused={int(fused)} small={int(small)} block={block:2d} pipe={int(pipe)} {note} {'ok' if ok else 'MISMATCH'}", flush=True)
            bad += not ok
            continue
        done = 0
        for k in pieces:
            err = None
            try:
                res = s.solve(k)
            except (exceptions.PrimalIsUnboundedError, exceptions.DualIsUnboundedError) as e:
                err = type(e).__name__
            r = o.solve(k)
            gl = s.pivot_log().tolist()
            ok = ok and gl == o.log[:len(gl)].tolist()
            if err:
                ok = ok and r.status in (O.PRIMAL_UNBOUNDED, O.DUAL_UNBOUNDED)
                note = err
                break
            ok = ok and res.iters == r.iters and np.array_equal(res.x, r.x) and res.basis.tolist() == r.basis.tolist()
            ok = ok and bool(res.optimum) == (r.status == O.OPTIMAL)
            if r.status != O.RUNNING:
                break
        print(f"{it:3d}: {kind:6s} {m:3d}x{n:4d} cond={int(condensed)} fused={int(fused)} small={int(small)} block={block:2d} pipe={int(pipe)} pivots={len(s.pivot_log()):5d} {note} "
              f"{'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
    print("failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
