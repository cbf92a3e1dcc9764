"""Device-buffer plumbing: torch (PyTorch-ROCm) owns HBM allocations and streams; liblpx owns the math."""
import numpy as np

from .exceptions import LpxBackendError


def torch_cuda():
    try:
        import torch
    except Exception as e:  # pragma: no cover
        raise LpxBackendError(f"PyTorch-ROCm is required for device buffers: {e}") from e
    if not torch.cuda.is_available():
        raise LpxBackendError("no HIP device visible (torch.cuda.is_available() is False); "
                              "liblpx has no CPU fallback")
    return torch


def stream_ptr():
    torch = torch_cuda()
    return torch.cuda.current_stream().cuda_stream


def roundup(a, b):
    return (a + b - 1) // b * b


def to_device_2d(host, ld, dtype):
    """Copy a 2-D NumPy array into a zero-padded (rows, ld) device tensor."""
    torch = torch_cuda()
    rows, cols = host.shape
    out = torch.zeros((rows, ld), dtype=dtype, device="cuda")
    out[:, :cols].copy_(torch.from_numpy(np.ascontiguousarray(host)).to("cuda", non_blocking=False))
    return out


def to_device(host, dtype=None):
    torch = torch_cuda()
    t = torch.from_numpy(np.ascontiguousarray(host)).to("cuda")
    return t if dtype is None else t.to(dtype)
