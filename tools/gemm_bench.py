"""TFLOP/s of lpx_dgemm (fp64 MFMA) and of one refinement step: python tools/gemm_bench.py [m ...] (GPU box)."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

import linprog_amd  # noqa: F401
from linprog_amd import _lib

L = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for m in [int(v) for v in sys.argv[1:]] or [1000, 4096, 10000]:
    A = torch.randn((m, m), dtype=torch.float64, device="cuda")
    B = torch.randn((m, m), dtype=torch.float64, device="cuda")
    D = torch.zeros((m, m), dtype=torch.float64, device="cuda")
    for transa in (0, 1):
        reps = 3 if m >= 8000 else 20
        for _ in range(2):
            L.lpx_dgemm(transa, m, m, m, 1.0, A.data_ptr(), m, B.data_ptr(), m, 0.0, None, 0, 0.0, D.data_ptr(), m, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            L.lpx_dgemm(transa, m, m, m, 1.0, A.data_ptr(), m, B.data_ptr(), m, 0.0, None, 0, 0.0, D.data_ptr(), m, st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"m={m} transa={transa}: {dt * 1e3:.3f} ms  {2.0 * m ** 3 / dt / 1e12:.2f} TFLOP/s "
              f"({2.0 * m ** 3 / dt / 78.6e12 * 100:.1f} % of 78.6)", flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    C = A @ B
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    C = A @ B
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"m={m} torch.matmul (rocBLAS/hipBLASLt): {dt * 1e3:.3f} ms  {2.0 * m ** 3 / dt / 1e12:.2f} TFLOP/s", flush=True)
