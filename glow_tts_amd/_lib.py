"""ctypes binding of libglowtts_hip.so (C ABI in include/glowtts_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails, this module
raises.  PyTorch is used only for device memory and streams (tensor.data_ptr(), current stream)."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libglowtts_hip.so")
_lib = None

c_int, c_float, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p


class GlowTTSHipError(RuntimeError):
    pass


def lib():
    """Loads the shared library (once).  Raises if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GlowTTSHipError(
                f"{LIB_PATH} not found: build it with `make -C glow_tts_amd/csrc` "
                "(or __graft_entry__.build()); the HIP path has no CPU fallback")
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(L):
    L.glowtts_abi_version.restype = c_int
    L.glowtts_device_arch.argtypes = [ctypes.c_char_p, c_int]
    L.glowtts_mas_dp_f32.argtypes = [c_void_p] * 5 + [c_int] * 3 + [c_float, c_void_p]
    L.glowtts_mas_path_from_idx.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    L.glowtts_mas_f32.argtypes = [c_void_p] * 5 + [c_int] * 3 + [c_float, c_void_p]
    L.glowtts_pack_weight.argtypes = [c_void_p] + [c_int] * 7 + [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_void_p]
    L.glowtts_conv_cl.argtypes = [c_void_p, c_void_p]


def check(rc, what):
    if rc != 0:
        raise GlowTTSHipError(f"{what} failed with code {rc}")


def ptr(t):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise GlowTTSHipError("expected a device tensor (the HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise GlowTTSHipError("expected a contiguous tensor")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream (as an integer for ctypes).  Called once per kernel launch: the raw accessor avoids
    building a torch.cuda.Stream object each time (eager launches are host-bound)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
