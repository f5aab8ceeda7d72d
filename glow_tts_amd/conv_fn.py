"""torch.autograd.Function wrappers around the channels-last MFMA convolution (glowtts_conv_cl), its data gradient
(same kernel, transposed weight image) and its weight gradient (glowtts_wgrad_cl).

Activations are "rows" tensors [R, C] (R = B * (T + 2*ROW_PAD), channels contiguous, zero pad rows around every
utterance - include/glowtts_hip.h).  Used by the text encoder; the flow decoder drives the same kernels through the
per-flow C entry points instead (decoder.py)."""
import ctypes

import torch

from . import _lib, ops

_decl = False


class WgradArgs(ctypes.Structure):
    """Mirror of `glowtts_wgrad_args`."""
    _fields_ = [("dy", ctypes.c_void_p), ("lddy", ctypes.c_int64), ("x", ctypes.c_void_p), ("ldx", ctypes.c_int64),
                ("xpro", ctypes.c_int), ("xmask", ctypes.c_void_p),
                ("rows", ctypes.c_int), ("m", ctypes.c_int), ("ca", ctypes.c_int), ("taps", ctypes.c_int), ("pad", ctypes.c_int),
                ("perm", ctypes.c_int), ("perm_h", ctypes.c_int), ("precision", ctypes.c_int),
                ("splits", ctypes.c_int), ("accumulate", ctypes.c_int),
                ("dw", ctypes.c_void_p), ("dbias", ctypes.c_void_p)]


def _L():
    global _decl
    L = _lib.lib()
    if not _decl:
        L.glowtts_wgrad_cl.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _decl = True
    return L


def wgrad(dy, x, O, ca, taps, precision, want_bias=True, splits=0):
    """dW [O, ca, taps], db [O] from dy rows [R, >=O] and x rows [R, >=ca]."""
    R = dy.shape[0]
    dw = torch.zeros(O, ca, taps, device=dy.device)
    db = torch.zeros(O, device=dy.device) if want_bias else None
    a = WgradArgs()
    a.dy, a.lddy, a.x, a.ldx = dy.data_ptr(), dy.shape[1], x.data_ptr(), x.shape[1]
    a.rows, a.m, a.ca, a.taps, a.pad = R, O, ca, taps, (taps - 1) // 2
    a.precision, a.splits, a.accumulate = precision, splits, 1
    a.dw, a.dbias = dw.data_ptr(), (db.data_ptr() if db is not None else None)
    _lib.check(_L().glowtts_wgrad_cl(ctypes.byref(a), _lib.stream()), "glowtts_wgrad_cl")
    return dw, db


class ConvRows(torch.autograd.Function):
    """y = [relu]( conv1d_same(x, w) + b ) [+ residual] [* rowmask]   on rows tensors.
    Mirrors torch.nn.Conv1d(k, padding=(k-1)//2) + the elementwise tail the reference applies after it."""

    @staticmethod
    def forward(ctx, x, w, b, rowmask, residual, relu, mask_out, precision):
        x = x.contiguous()
        R, Cin = x.shape
        O, Ci2, k = w.shape
        assert Ci2 <= Cin and Cin % 4 == 0
        pw = ops.pack_weight(w.detach(), precision=precision)
        out = torch.empty(R, O, device=x.device)
        flags = (ops.F_BIAS if b is not None else 0) | (ops.F_RELU if relu else 0) | (ops.F_MASK if mask_out else 0) | \
                (ops.F_ADD_IN0 if residual is not None else 0)
        ops.conv_cl(x, pw, Ci2, R, lda=Cin, pad=(k - 1) // 2, epi=ops.EPI_LINEAR, flags=flags, n=O,
                    bias=b.detach().contiguous() if b is not None else None, rowmask=rowmask,
                    in0=residual.contiguous() if residual is not None else None, ldi0=O, out0=out, ld0=O)
        ctx.save_for_backward(x, w, out if relu else None, rowmask)
        ctx.cfg = (relu, mask_out, precision, b is not None, residual is not None, Ci2)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w, out, rowmask = ctx.saved_tensors
        relu, mask_out, precision, has_b, has_res, Ci2 = ctx.cfg
        R, Cin = x.shape
        O, _, k = w.shape
        dy = dy.contiguous()
        dres = None
        if mask_out:
            dy = dy * rowmask.unsqueeze(1)
        if has_res and ctx.needs_input_grad[4]:
            dres = dy
        dz = dy * (out > 0) if relu else dy                   # d(pre-activation)
        dx = None
        if ctx.needs_input_grad[0]:
            pwt = ops.pack_weight(w.detach(), transpose=True, precision=precision)
            dx = torch.empty(R, Cin, device=x.device) if Cin == Ci2 else torch.zeros(R, Cin, device=x.device)
            ops.conv_cl(dz, pwt, O, R, lda=O, pad=(k - 1) // 2, epi=ops.EPI_LINEAR, flags=0, n=Ci2, out0=dx, ld0=Cin)
        dw = db = None
        if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
            dw, db = wgrad(dz, x, O, Ci2, k, precision, want_bias=has_b, splits=min(4, max(1, R // 512)))
        return dx, dw, db, None, dres, None, None, None


def conv_rows(x, w, b, rowmask, relu=False, mask_out=False, residual=None, precision=ops.BF16):
    return ConvRows.apply(x, w, b, rowmask, residual, relu, mask_out, precision)
