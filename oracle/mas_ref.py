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


def maximum_path_python(values, t_xs, t_ys):
    """The reference's pure-Python twin, Modules.py:951-980 `Maximum_Path_Generater.calc_path` (what BASELINE config 1 runs when
    Use_Cython_Alignment is false): same recurrence with the sentinel -1e7 instead of core.pyx's -1e9, plain Python loops over numpy
    scalars.  values float32 [B,Tx,Ty] pre-multiplied by the mask.  Identical to maximum_path_c unless a cumulative score falls between
    -1e9 and -1e7.  Slow by construction (~60-100 ms per 120 x 800 utterance): bench.py's cpu_baseline times it."""
    values = np.ascontiguousarray(values, dtype=np.float32).copy()
    paths = []
    for x, token_length, mel_length in zip(values, t_xs, t_ys):
        token_length, mel_length = int(token_length), int(mel_length)
        path = np.zeros_like(x).astype(np.int32)
        for mel_index in range(mel_length):                                                       # :958-972
            for token_index in range(max(0, token_length + mel_index - mel_length), min(token_length, mel_index + 1)):
                current_q = -1e+7 if mel_index == token_index else x[token_index, mel_index - 1]
                if token_index == 0:
                    prev_q = 0.0 if mel_index == 0 else -1e+7
                else:
                    prev_q = x[token_index - 1, mel_index - 1]
                x[token_index, mel_index] = max(current_q, prev_q) + x[token_index, mel_index]
        token_index = token_length - 1                                                            # :974-978
        for mel_index in range(mel_length - 1, -1, -1):
            path[token_index, mel_index] = 1
            if token_index == mel_index or x[token_index, mel_index - 1] < x[token_index - 1, mel_index - 1]:
                token_index = max(0, token_index - 1)
        paths.append(path)
    return np.stack(paths, axis=0)
