"""Device plumbing: one C-ABI context per GPU, torch used only for HBM buffers and streams."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _cabi

_CTX = {}


def _torch():
    import torch
    return torch


_GPU_OK = False


def require_gpu():
    global _GPU_OK
    torch = _torch()
    if _GPU_OK:
        return torch
    if not torch.cuda.is_available():
        raise _cabi.IsoError("isochrones_amd needs an AMD GPU (HIP device); none is visible and there "
                             "is no CPU fallback")
    _GPU_OK = True
    return torch


def current_device() -> int:
    return require_gpu().cuda.current_device()


def context(device: int | None = None):
    """The (cached) iso_ctx* of a device."""
    torch = require_gpu()
    if device is None:
        device = torch.cuda.current_device()
    if device not in _CTX:
        h = C.c_void_p()
        _cabi.check(_cabi.lib().iso_ctx_create(C.byref(h), int(device)))
        _CTX[device] = h
    return _CTX[device]


def stream_ptr(device: int | None = None):
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def is_tensor(x) -> bool:
    try:
        import torch
    except Exception:  # pragma: no cover
        return False
    return isinstance(x, torch.Tensor)


def to_device_f64(x, device: int):
    """numpy / sequence / tensor -> contiguous float64 tensor on `device`."""
    torch = _torch()
    dev = torch.device("cuda", device)
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=torch.float64).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=dev)


def empty_f64(shape, device: int):
    torch = _torch()
    return torch.empty(shape, dtype=torch.float64, device=torch.device("cuda", device))


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def i32_array(values):
    a = np.ascontiguousarray(values, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))
