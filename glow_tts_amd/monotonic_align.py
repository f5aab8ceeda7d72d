"""Drop-in for the reference's `monotonic_align` package (monotonic_align/__init__.py:6-21):
`maximum_path(value, mask) -> path`, same dtype/device as `value`, computed on the GPU by
glowtts_mas_dp_f32 + glowtts_mas_path_from_idx (no device->host->device round trip)."""
import torch

from . import _lib


def maximum_path_idx(value, t_x, t_y, max_neg_val=-1e9, want_q=False):
    """value [B,Tx,Ty] f32 (pre-masked), t_x/t_y [B] i32 on device -> idx [B,Ty] i32 (-1 past t_y)."""
    B, Tx, Ty = value.shape
    idx = torch.empty((B, Ty), dtype=torch.int32, device=value.device)
    q = value.clone() if want_q else None
    _lib.check(_lib.lib().glowtts_mas_dp_f32(_lib.ptr(value), _lib.ptr(t_x), _lib.ptr(t_y), _lib.ptr(idx),
                                             _lib.ptr(q), B, Tx, Ty, max_neg_val, _lib.stream()), "glowtts_mas_dp_f32")
    return (idx, q) if want_q else idx


def path_from_idx(idx, Tx, dtype=torch.float32):
    B, Ty = idx.shape
    kind = {torch.int32: 0, torch.float32: 1}[dtype]
    path = torch.empty((B, Tx, Ty), dtype=dtype, device=idx.device)
    _lib.check(_lib.lib().glowtts_mas_path_from_idx(_lib.ptr(idx), _lib.ptr(path), B, Tx, Ty, kind, _lib.stream()),
               "glowtts_mas_path_from_idx")
    return path


def maximum_path(value, mask, max_neg_val=-1e9):
    """value, mask: [b, t_x, t_y] device tensors.  Returns the 0/1 path in value's dtype."""
    dtype = value.dtype
    value = (value * mask).to(torch.float32).contiguous()
    t_x = mask.sum(1)[:, 0].to(torch.int32)          # __init__.py:18
    t_y = mask.sum(2)[:, 0].to(torch.int32)          # __init__.py:19
    idx = maximum_path_idx(value, t_x.contiguous(), t_y.contiguous(), max_neg_val)
    path = path_from_idx(idx, value.shape[1], torch.float32)
    return path if dtype == torch.float32 else path.to(dtype)


def maximum_path_host(value, mask, max_neg_val=-1e9, num_threads=0):
    """The reference wrapper (monotonic_align/__init__.py:6-21) for HOST tensors / arrays, through the library's host twin
    `glowtts_mas_f32_host` (same arithmetic as core.pyx; utterances over host threads).  Not a fallback of the GPU path: `maximum_path`
    rejects CPU tensors; call this one explicitly when the score matrix lives in host memory."""
    import ctypes
    import numpy as np
    is_t = torch.is_tensor(value)
    dtype = value.dtype if is_t else None
    v = (value * mask).detach().cpu().numpy() if is_t else np.asarray(value) * np.asarray(mask)
    mk = mask.detach().cpu().numpy() if torch.is_tensor(mask) else np.asarray(mask)
    v = np.ascontiguousarray(v, dtype=np.float32)
    t_x = np.ascontiguousarray(mk.sum(1)[:, 0], dtype=np.int32)
    t_y = np.ascontiguousarray(mk.sum(2)[:, 0], dtype=np.int32)
    path = np.zeros(v.shape, dtype=np.int32)
    L = _lib.lib()
    L.glowtts_mas_f32_host.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_int]
    _lib.check(L.glowtts_mas_f32_host(v.ctypes.data, path.ctypes.data, t_x.ctypes.data, t_y.ctypes.data, v.shape[0], v.shape[1], v.shape[2],
                                      max_neg_val, int(num_threads)), "glowtts_mas_f32_host")
    return torch.from_numpy(path).to(dtype) if is_t else path
