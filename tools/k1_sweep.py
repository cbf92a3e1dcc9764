"""Sweep the launch parameters of the rank-1 pivot-update kernel (K1) on the GPU box.

    python tools/k1_sweep.py [rows cols [ld]]   (default 20001 x 40001, the BASELINE config-4 tableau, 4-KiB-aligned rows)

Prints achieved algorithmic GB/s (16 B per element) per variant, from HIP events around K1 only, and two
torch reference points measured the same way (in-place add = read+write stream, clone = copy).
"""
import ctypes as C
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import linprog_amd  # noqa: E402,F401
from linprog_amd import _lib  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20001
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 40001
ld = int(sys.argv[3]) if len(sys.argv) > 3 else -(-cols // 512) * 512
L = _lib.load()
M = torch.rand((rows, ld), dtype=torch.float64, device="cuda")
e = (torch.rand(rows, dtype=torch.float64, device="cuda") - 0.5) * 1e-3
scr = torch.zeros(ld + 2, dtype=torch.float64, device="cuda")
nbytes = 16.0 * rows * cols
REP = 10


def timed(fn):
    fn()
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REP):
        fn()
    t.record()
    torch.cuda.synchronize()
    return s.elapsed_time(t) / REP


ms = timed(lambda: M.add_(1e-9))
print(f"torch in-place add (read+write)   {nbytes / ms / 1e6:8.0f} GB/s  {ms:.3f} ms")
M2 = torch.empty_like(M)
ms = timed(lambda: M2.copy_(M))
print(f"torch copy (read A, write B)      {nbytes / ms / 1e6:8.0f} GB/s  {ms:.3f} ms")
del M2

results = []
for nt, vpt, un, rpb, thr in itertools.product((0, 1), (1, 2), (1, 2), (1, 2), (64, 128, 256, 512, 1024)):
    if rpb != un:
        continue
    os.environ.update(LPX_RANK1_THREADS=str(thr), LPX_RANK1_NT=str(nt), LPX_RANK1_VPT=str(vpt), LPX_RANK1_UNROLL=str(un), LPX_RANK1_ROWS=str(rpb))
    _lib.check(L.lpx_pivot_update(M.data_ptr(), rows, cols, ld, 7, e.data_ptr(), scr.data_ptr(), None))
    torch.cuda.synchronize()
    _lib.check(L.lpx_profile_k1(1))
    for _ in range(REP):
        _lib.check(L.lpx_pivot_update(M.data_ptr(), rows, cols, ld, 7, e.data_ptr(), scr.data_ptr(), None))
    tot, n = C.c_double(), C.c_int64()
    _lib.check(L.lpx_profile_k1_read(C.byref(tot), C.byref(n)))
    _lib.check(L.lpx_profile_k1(0))
    ms = tot.value / n.value
    results.append((nbytes / ms / 1e6, nt, vpt, un, rpb, thr, ms))
    print(f"nt={nt} vpt={vpt} unroll={un} rows/block={rpb} thr={thr:4d}  {nbytes / ms / 1e6:8.0f} GB/s  {ms:.3f} ms", flush=True)
best = max(results)
print("BEST", best)
