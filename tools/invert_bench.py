#!/usr/bin/env python3
"""Time lpx_dinverse (blocked Gauss-Jordan, fp64 MFMA trailing updates) against torch.linalg.inv (rocSOLVER) and, for
small sizes, numpy on the host.  GPU box: python3 tools/invert_bench.py [m ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from linprog_amd import _lib

    lib = _lib.load()
    sizes = [int(a) for a in sys.argv[1:]] or [1000, 2000, 4000, 10000]
    st = torch.cuda.current_stream().cuda_stream
    for m in sizes:
        g = torch.Generator(device="cuda").manual_seed(m)
        A = torch.randn((m, m), generator=g, device="cuda", dtype=torch.float64) + 0.5 * m ** 0.5 * torch.eye(
            m, device="cuda", dtype=torch.float64)
        X = torch.empty_like(A)
        nbytes = lib.lpx_invert_workspace_bytes(m)
        work = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        info = torch.zeros(1, dtype=torch.int64, device="cuda")

        def run():
            _lib.check(lib.lpx_dinverse(m, A.data_ptr(), m, X.data_ptr(), m, work.data_ptr(), nbytes, info.data_ptr(), st))

        run()
        torch.cuda.synchronize()
        reps = 3 if m >= 4000 else 10
        t = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        ours = (time.perf_counter() - t) / reps
        resid = float((A @ X - torch.eye(m, device="cuda", dtype=torch.float64)).abs().max())
        try:
            torch.linalg.inv(A)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                torch.linalg.inv(A)
            torch.cuda.synchronize()
            theirs = (time.perf_counter() - t) / reps
        except RuntimeError as e:  # rocSOLVER's workspace allocation fails at m = 10000 on this image
            theirs = float("nan")
            print("torch.linalg.inv failed:", str(e).splitlines()[0][:100])
        host = None
        if m <= 4000:
            Ah = A.cpu().numpy()
            t = time.perf_counter()
            np.linalg.inv(Ah)
            host = time.perf_counter() - t
        print(f"m={m}: lpx_dinverse {ours * 1e3:.2f} ms ({2 * m ** 3 / ours / 1e12:.2f} TFLOP/s, info {int(info.item())}, "
              f"|AX-I| {resid:.2e}); torch.linalg.inv {theirs * 1e3:.2f} ms"
              + (f"; numpy host {host * 1e3:.1f} ms" if host is not None else ""), flush=True)


if __name__ == "__main__":
    main()
