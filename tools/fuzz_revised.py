#!/usr/bin/env python3
"""Random shapes through the drop-in solver classes in revised form (PrimalRevisedSimplexSolver, DualRevisedSimplexSolver,
BoundedVariablePrimalSimplexSolver: the one-launch small-LP kernel up to m = 128 / 96, the multi-kernel chain beyond, solve()
in pieces) against the oracle's loops: pivot log identical, x / cost within 1e-9, iteration counts, status.
GPU box: python3 tools/fuzz_revised.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def close(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return bool(np.all(np.abs(a - b) <= 1e-9 * np.maximum(1.0, np.abs(b))))


def main():
    from linprog_amd import exceptions, problems
    from linprog_amd.dual_solvers import DualRevisedSimplexSolver
    from linprog_amd.primal_solvers import PrimalRevisedSimplexSolver
    from linprog_amd.special_solvers import BoundedVariablePrimalSimplexSolver
    from oracle import oracle as O

    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for it in range(cases):
        kind = str(rng.choice(["primal", "dual", "bounded"]))
        m = int(rng.integers(2, 170 if kind == "bounded" else 400))
        n = m + int(rng.integers(1, 400))
        seed = 2000 + it
        pieces = [int(v) for v in rng.choice([1, 2, 5, 33, 64], size=2)] + [100000]
        if kind == "bounded":
            box = problems.gen_box(m, n, seed)
            s = type("L", (BoundedVariablePrimalSimplexSolver,), {"_LOG_CAP": 20000})(*box)
            o = O.OracleBounded(*box)
            pieces = [100000]
        else:
            c, A, b, basis = (problems.gen_dual if kind == "dual" else problems.gen_primal)(m, n, seed)
            if kind == "primal" and rng.random() < 0.15:
                A = A.copy(); A[:, : n - m] *= -1
            cls = DualRevisedSimplexSolver if kind == "dual" else PrimalRevisedSimplexSolver
            s = type("L", (cls,), {"_LOG_CAP": 20000})(c, A, b, basis)
            o = O.OracleRevised(c, A, b, basis, kind=kind, log_cap=20000)
        ok, note = True, ""
        for k in pieces:
            err = None
            try:
                res = s.solve(k)
            except (exceptions.PrimalIsUnboundedError, exceptions.DualIsUnboundedError) as e:
                err = type(e).__name__
            r = o.solve(k)
            gl = s.pivot_log().tolist()
            ok = ok and gl == np.asarray(o.log)[:len(gl)].tolist()
            if err:
                ok = ok and r.status in (O.PRIMAL_UNBOUNDED, O.DUAL_UNBOUNDED)
                note = err
                break
            ok = ok and res.iters == r.iters and close(res.x, r.x) and close(res.cost, r.cost)
            ok = ok and sorted(np.asarray(res.basis).tolist()) == sorted(np.asarray(r.basis).tolist())
            ok = ok and bool(res.optimum) == (r.status == O.OPTIMAL)
            if r.status != O.RUNNING:
                break
        print(f"{it:3d}: {kind:7s} {m:3d}x{n:4d} pivots={len(s.pivot_log()):5d} {note} {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
    print("failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
