"""K1 (default launch plan) over matrix shapes: does the ragged last strip or the row length cost bandwidth?
    python tools/k1_shapes.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import linprog_amd  # noqa: E402,F401
from linprog_amd import _lib  # noqa: E402

L = _lib.load()
REP = 10
SHAPES = [(20001, 40001, 0), (20001, 20001, 0), (20001, 19968, 0), (20001, 20480, 0), (40001, 20001, 0), (20001, 10001, 0),
          (20001, 5001, 0), (20001, 2501, 0), (20001, 2501, 512), (20001, 2501, 1536), (20001, 40001, 512), (20001, 20001, 1024)]
for rows, cols, extra in SHAPES:  # extra: additional row padding (doubles) on top of the 4-KiB alignment
    ld = -(-cols // 512) * 512 + extra
    M = torch.rand((rows, ld), dtype=torch.float64, device="cuda")
    e = (torch.rand(rows, dtype=torch.float64, device="cuda") - 0.5) * 1e-3
    scr = torch.zeros(ld + 2, dtype=torch.float64, device="cuda")
    _lib.check(L.lpx_pivot_update(M.data_ptr(), rows, cols, ld, 7, e.data_ptr(), scr.data_ptr(), None))
    torch.cuda.synchronize()
    _lib.check(L.lpx_profile_k1(1))
    for _ in range(REP):
        _lib.check(L.lpx_pivot_update(M.data_ptr(), rows, cols, ld, 7, e.data_ptr(), scr.data_ptr(), None))
    tot, n = C.c_double(), C.c_int64()
    _lib.check(L.lpx_profile_k1_read(C.byref(tot), C.byref(n)))
    _lib.check(L.lpx_profile_k1(0))
    ms = tot.value / n.value
    print(f"{rows} x {cols} (ld {ld}): {16.0 * rows * cols / ms / 1e6:7.0f} GB/s  {ms:.4f} ms", flush=True)
    del M
