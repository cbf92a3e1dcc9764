"""The allocation-free entry point (lpx_pivot_update_static) against the entry point it is a variant of
(lpx_pivot_update), through the C ABI: bit-identical output, two calls in a row, the second on a smaller matrix (the
static pivot-row buffer keeps the first call's contents).  Collected last on purpose (file name)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(torch, lib, L, rows, cols, pad, seed):
    rng = np.random.default_rng(seed)
    ld = cols + pad + ((cols + pad) % 2)
    M = np.zeros((rows, ld))
    M[:, :cols] = rng.standard_normal((rows, cols))
    d = rng.standard_normal(rows)
    r = int(rng.integers(rows))
    e = np.where(np.arange(rows) == r, 1.0 / d[r], -d / d[r])
    ed = torch.from_numpy(e).cuda()
    want = torch.from_numpy(M).cuda()
    scr = torch.zeros(ld + 2, dtype=torch.float64, device="cuda")
    lib.check(L.lpx_pivot_update(want.data_ptr(), rows, cols, ld, r, ed.data_ptr(), scr.data_ptr(), None))
    got = torch.from_numpy(M).cuda()
    warm = torch.from_numpy(M).cuda()
    # one warm-up call: the symbol look-up and the module load happen once per device; then the DRIVER's free-memory
    # figure (hipMemGetInfo -- torch.cuda.memory_allocated() only sees torch's own caching allocator, a hipMalloc inside
    # liblpx would not move it) must not change across a call
    lib.check(L.lpx_pivot_update_static(warm.data_ptr(), rows, cols, ld, r, ed.data_ptr(), None))
    torch.cuda.synchronize()
    before, before_torch = torch.cuda.mem_get_info()[0], torch.cuda.memory_allocated()
    lib.check(L.lpx_pivot_update_static(got.data_ptr(), rows, cols, ld, r, ed.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == before and torch.cuda.memory_allocated() == before_torch
    assert np.array_equal(got.cpu().numpy()[:, :cols], want.cpu().numpy()[:, :cols])
    assert np.array_equal(warm.cpu().numpy()[:, :cols], want.cpu().numpy()[:, :cols])


def test_static_pivot_update_matches_pivot_update_twice():
    import torch

    from linprog_amd import _lib

    L = _lib.load()
    _both(torch, _lib, L, 1001, 2001, 1, 1)   # first call: fills 2002 doubles of the static row buffer
    _both(torch, _lib, L, 65, 129, 1, 2)      # second call, smaller: stale tail of the buffer must not matter
    _both(torch, _lib, L, 3, 9, 1, 3)


def test_static_pivot_update_rejects_what_it_cannot_hold():
    import torch

    from linprog_amd import _lib

    L = _lib.load()
    cap = 65536
    M = torch.zeros((2, cap + 2), dtype=torch.float64, device="cuda")
    e = torch.zeros(2, dtype=torch.float64, device="cuda")
    assert L.lpx_pivot_update_static(M.data_ptr(), 2, cap + 2, cap + 2, 0, e.data_ptr(), None) == -3  # LPX_E_SIZE
    assert b"static pivot-row buffer" in L.lpx_last_error()
    assert L.lpx_pivot_update_static(M.data_ptr(), 2, 4, 3, 0, e.data_ptr(), None) == -1  # ld < cols
    # the limit itself is accepted
    M2 = torch.ones((2, cap), dtype=torch.float64, device="cuda")
    e2 = torch.tensor([2.0, 3.0], dtype=torch.float64, device="cuda")
    _lib.check(L.lpx_pivot_update_static(M2.data_ptr(), 2, cap, cap, 0, e2.data_ptr(), None))
    out = M2.cpu().numpy()
    assert np.all(out[0] == 2.0) and np.all(out[1] == 4.0)  # row 0 = e_0 * row 0 ; row 1 = 1 + 3 * 1
