"""ORACLE - test infrastructure.  ctypes binding of oracle/mas_ref.c (C restatement of
monotonic_align/core.pyx:9-45).  Built by `make -C oracle` / __graft_entry__.build()."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libmas_ref.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "libmas_ref.so"], stdout=subprocess.DEVNULL)
        _LIB = ctypes.CDLL(so)
        _LIB.mas_ref_maximum_path_c.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_float]
        _LIB.mas_ref_maximum_path_c.restype = None
    return _LIB


def maximum_path_c(values, t_xs, t_ys, max_neg_val=-1e9, return_q=False):
    """values float32 [B,Tx,Ty] (already multiplied by the mask, as __init__.py:11 does).
    Returns int32 path [B,Tx,Ty]; with return_q also the clobbered cumulative values."""
    q = np.ascontiguousarray(values, dtype=np.float32).copy()
    B, Tx, Ty = q.shape
    path = np.zeros((B, Tx, Ty), dtype=np.int32)
    t_xs = np.ascontiguousarray(t_xs, dtype=np.int32)
    t_ys = np.ascontiguousarray(t_ys, dtype=np.int32)
    _lib().mas_ref_maximum_path_c(path.ctypes.data, q.ctypes.data, t_xs.ctypes.data, t_ys.ctypes.data,
                                  B, Tx, Ty, ctypes.c_float(max_neg_val))
    return (path, q) if return_q else path


def reference_core():
    """The reference's own compiled core.pyx (oracle/_ref, built by oracle/build_ref.sh) or None."""
    import importlib.machinery
    import importlib.util
    d = os.path.join(_HERE, "_ref", "monotonic_align")
    if not os.path.isdir(d):
        return None
    so = [f for f in os.listdir(d) if f.startswith("core.") and f.endswith(".so")]
    if not so:
        return None
    loader = importlib.machinery.ExtensionFileLoader("core", os.path.join(d, so[0]))
    spec = importlib.util.spec_from_loader("core", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod
