"""torch.autograd.Function wrappers around the hand-written encoder / convolution kernels (C ABI in include/glowtts_hip.h):
  ConvRows       glowtts_conv_cl (+ transposed image for the data gradient) and glowtts_wgrad_cl
  LayerNormRows  glowtts_layernorm_fwd / _bwd            (LN + residual + ReLU + dropout + mask in one pass)
  EmbeddingRows  glowtts_embedding_fwd / _bwd
  RPRAttention   glowtts_rpr_attention_fwd / _bwd         (relative-position self-attention core)
Activations are "rows" tensors [R, C] (R = B * (T + 2*ROW_PAD), channels contiguous, zero pad rows around every utterance).
The flow decoder drives the same conv kernels through the per-flow C entry points instead (decoder.py)."""
import ctypes
import math

import torch

from . import _lib, ops

_decl = False
c_i64, c_int, c_f, c_u32, c_p = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_uint32, ctypes.c_void_p


class WgradArgs(ctypes.Structure):
    """Mirror of `glowtts_wgrad_args`."""
    _fields_ = [("dy", c_p), ("lddy", c_i64), ("x", c_p), ("ldx", c_i64),
                ("xpro", c_int), ("xmask", c_p),
                ("rows", c_int), ("m", c_int), ("ca", c_int), ("taps", c_int), ("pad", c_int),
                ("perm", c_int), ("perm_h", c_int), ("precision", c_int),
                ("splits", c_int), ("accumulate", c_int),
                ("dw", c_p), ("dbias", c_p), ("io_flags", c_int)]


def _L():
    global _decl
    L = _lib.lib()
    if not _decl:
        L.glowtts_wgrad_cl.argtypes = [c_p, c_p]
        L.glowtts_layernorm_fwd_io.argtypes = [c_p] * 8 + [c_i64, c_int, c_f, c_int, c_f, c_u32, c_p, c_p, c_p]
        L.glowtts_layernorm_qkv.argtypes = [c_p] * 10 + [c_int, c_p, c_p, c_i64, c_int, c_f, c_p]
        L.glowtts_proj_layernorm.argtypes = [c_p, c_i64, c_p, c_int] + [c_p] * 10 + [c_i64, c_int, c_f, c_f, c_u32, c_p, c_p]
        L.glowtts_layernorm_scratch_floats.argtypes = [c_i64, c_int]
        L.glowtts_layernorm_scratch_floats.restype = c_i64
        L.glowtts_layernorm_bwd_io.argtypes = [c_p] * 9 + [c_i64, c_int, c_int, c_f, c_p, c_p, c_f, c_p]
        L.glowtts_colsum_batched.argtypes = [c_p, c_p, c_int, c_int, c_int, c_i64, c_i64, c_p]
        L.glowtts_gate_bwd_io.argtypes = [c_p] * 4 + [c_i64, c_int, c_f, c_int, c_p]
        L.glowtts_embedding_fwd.argtypes = [c_p] * 4 + [c_int] * 3 + [c_f, c_p]
        L.glowtts_embedding_bwd.argtypes = [c_p] * 4 + [c_int] * 4 + [c_f, c_p]
        L.glowtts_rpr_attention_fwd_prec.argtypes = [c_p] * 6 + [c_int] * 5 + [c_f, c_u32, c_p, c_int, c_p]
        L.glowtts_rpr_attention_scratch_floats.argtypes = [c_int] * 5
        L.glowtts_rpr_attention_scratch_floats.restype = c_i64
        L.glowtts_rpr_attention_bwd_prec.argtypes = [c_p] * 11 + [c_int] * 5 + [c_f, c_u32, c_p, c_int, c_p]
        L.glowtts_rpr_attention_bwd_partial_rows.argtypes = [c_int] * 5
        L.glowtts_sum_slices.argtypes = [c_p, c_p, c_int, c_i64, c_p]
        L.glowtts_rpr_attention_bwd_partial_rows.restype = c_i64
        _decl = True
    return L


def wgrad(dy, x, O, ca, taps, precision, want_bias=True, splits=0, xpro=0, io_flags=0):
    """dW [O, ca, taps], db [O] from dy rows [R, >=O] and x rows [R, >=ca]."""
    R = dy.shape[0]
    buf = torch.zeros(O * ca * taps + (O if want_bias else 0), device=dy.device)      # one fill for both (split-K adds atomically)
    dw = buf[:O * ca * taps].view(O, ca, taps)
    db = buf[O * ca * taps:] if want_bias else None
    a = WgradArgs()
    a.dy, a.lddy, a.x, a.ldx = dy.data_ptr(), dy.shape[1], x.data_ptr(), x.shape[1]
    a.rows, a.m, a.ca, a.taps, a.pad = R, O, ca, taps, (taps - 1) // 2
    a.precision, a.splits, a.accumulate = precision, splits, 1
    a.xpro, a.io_flags = xpro, io_flags
    a.dw, a.dbias = dw.data_ptr(), (db.data_ptr() if db is not None else None)
    _lib.check(_L().glowtts_wgrad_cl(ctypes.byref(a), _lib.stream()), "glowtts_wgrad_cl")
    return dw, db


def _sp(t):
    return t.data_ptr() if t is not None else None


class WgradTape:
    """Weight-gradient problems of the convs of ONE forward pass.  ConvRows.backward only records (dz, x) and hands autograd an
    unfilled dW / db; ParamGate.backward - which autograd runs after every consumer of the gated parameters - fills all of them with
    one grouped launch per (taps, precision): no split-K atomics, no zero fills, 3 launches instead of ~30 for the text encoder."""

    def __init__(self):
        self.jobs = []
        self.ln_count, self.ln = 0, None           # LayerNorms of the block functions: their gamma / beta partials are reduced by ONE launch
        self.att_count, self.att = 0, None         # attention cores of the block functions: their relative-position gradients likewise
        # Row splits (round 5).  A weight gradient reduces over the rows, and the text encoder has few of them (B x (T + 4) = 3 968 at 32 x 120 tokens): a
        # launch of the grouped kernels is ~150 tiles, each a serial walk over all rows - 50-115 us per launch, six launches at the very end of the
        # encoder's backward, behind which the optimizer waits.  With `splits` = S > 1 every problem is issued S times over B / S utterances each (the
        # rows just outside a split are the neighbouring utterances' zero pad rows: exact) into S partial images; one more launch sums them, in a
        # fixed order, into the gradients - which for that purpose live in ONE arena (`out`), sized at forward time (`capacity`).
        self.splits, self.rows_per_utt, self.capacity = 1, 0, 0
        self.arena, self.used = None, 0

    def out(self, shape, device):
        """A gradient tensor of the taped convs: a 16-byte aligned slice of the tape's arena (or a tensor of its own when the tape keeps none)."""
        n = 1
        for d_ in shape:
            n *= int(d_)
        if self.capacity <= 0 or self.used + n > self.capacity:
            return torch.empty(shape, device=device)
        if self.arena is None:
            self.arena = torch.empty(self.capacity, device=device)
        t = self.arena[self.used:self.used + n].view(shape)
        self.used += (n + 3) & ~3
        return t

    def ln_slot(self, R, C, device):
        """-> (scratch [nfloats], gb [2C]) of the next LayerNorm backward: slices of two buffers that `flush` reduces with one
        glowtts_colsum_batched (every LayerNorm registered through `ln_count` in the forward has the same shape)."""
        if self.ln is None:
            nf = _L().glowtts_layernorm_scratch_floats(R, C)
            self.ln = {"scratch": torch.empty(self.ln_count, nf, device=device), "gb": torch.empty(self.ln_count, 2 * C, device=device),
                       "used": 0, "nblk": nf // (2 * C), "C": C, "R": R}
        st = self.ln
        assert st["used"] < self.ln_count and (st["R"], st["C"]) == (R, C)
        i = st["used"]
        st["used"] += 1
        return st["scratch"][i], st["gb"][i]

    def att_slot(self, B, Tp, H, D, win, device):
        """-> (scratch, drel [2][2 win + 1][D]) of the next attention backward: the kernel leaves its per-workgroup partial sums of (d relK | d relV) in
        `scratch`, `flush` reduces every layer's with one launch - the reduction is not on the chain to the next layer's gradient."""
        L = _L()
        if self.att is None:
            nf = L.glowtts_rpr_attention_scratch_floats(B, Tp, H, D, win)
            cols = 2 * (2 * win + 1) * D
            self.att = {"scratch": torch.empty(self.att_count, nf, device=device), "out": torch.empty(self.att_count, cols, device=device), "used": 0,
                        "rows": L.glowtts_rpr_attention_bwd_partial_rows(B, Tp, H, D, win), "cols": cols, "key": (B, Tp, H, D, win)}
        st = self.att
        assert st["used"] < self.att_count and st["key"] == (B, Tp, H, D, win)
        i = st["used"]
        st["used"] += 1
        return st["scratch"][i], st["out"][i].view(2, 2 * win + 1, D)

    def add(self, dz, x, O, ca, taps, precision, dw, db):
        self.jobs.append((dz, x, O, ca, taps, precision, dw, db))

    def flush(self):
        from .decoder import WgradGroup
        groups = {}
        S, part, base = 1, None, 0
        if self.splits > 1 and self.arena is not None and self.rows_per_utt > 0 and self.jobs:
            lo, hi = self.arena.data_ptr(), self.arena.data_ptr() + 4 * self.used
            R0 = self.jobs[0][0].shape[0]
            inside = all(lo <= dw.data_ptr() < hi and (db is None or lo <= db.data_ptr() < hi) and dz.shape[0] == R0 for dz, _, _, _, _, _, dw, db in self.jobs)
            nutt = R0 // self.rows_per_utt
            if inside and R0 == nutt * self.rows_per_utt and nutt % self.splits == 0:
                S, base = self.splits, lo
                part = torch.empty(S, self.used, device=self.arena.device)
        for dz, x, O, ca, taps, precision, dw, db in self.jobs:
            io = (ops.WIO_DY_BF16 if dz.dtype == torch.bfloat16 else 0) | (ops.WIO_X_BF16 if x.dtype == torch.bfloat16 else 0)
            rows = dz.shape[0] // S
            key = (rows, taps, precision, io)
            if key not in groups:
                groups[key] = WgradGroup(rows, taps, precision, io_flags=io, tag="enc")
            for s_ in range(S):
                shift = 0 if part is None else part[s_].data_ptr() - base          # this split's image of the arena
                groups[key].add(dz.data_ptr() + s_ * rows * dz.shape[1] * dz.element_size(), dz.shape[1], O,
                                x.data_ptr() + s_ * rows * x.shape[1] * x.element_size(), x.shape[1], ca, dw.data_ptr() + shift,
                                (db.data_ptr() + shift) if db is not None else None)
        # one host-to-device copy for the job tables of all groups, then the launches back to back (a copy in front of every launch was a memcpy
        # node + two dependency hops, ~13 us each, six times at the very end of the encoder's chain)
        for g in groups.values():
            g.end_segment()
        WgradGroup.upload_all(list(groups.values()), self.jobs[0][0].device)
        for g in groups.values():
            g.launch_segment(0)
        if part is not None:                                   # gradients = sum of the splits' partial images (one launch, fixed order)
            _lib.check(_L().glowtts_sum_slices(part.data_ptr(), self.arena.data_ptr(), S, self.used, _lib.stream()), "glowtts_sum_slices")
        self.jobs = []        # (dz / x stay referenced by the launched work's stream ordering: same stream, freed after)
        self.arena, self.used = None, 0
        if self.ln is not None:
            st = self.ln
            assert st["used"] == self.ln_count, "a LayerNorm of the block functions did not run its backward"
            _lib.check(_L().glowtts_colsum_batched(st["scratch"].data_ptr(), st["gb"].data_ptr(), st["nblk"], 2 * st["C"], st["used"],
                                                   st["scratch"].shape[1], 2 * st["C"], _lib.stream()), "glowtts_colsum_batched")
            self.ln = None
        if self.att is not None:
            st = self.att
            assert st["used"] == self.att_count, "an attention core of the block functions did not run its backward"
            _lib.check(_L().glowtts_colsum_batched(st["scratch"].data_ptr(), st["out"].data_ptr(), st["rows"], st["cols"], st["used"],
                                                   st["scratch"].shape[1], st["cols"], _lib.stream()), "glowtts_colsum_batched")
            self.att = None


class ParamGate(torch.autograd.Function):
    """Identity on the parameters that the taped convs consume; its backward launches the tape."""

    @staticmethod
    def forward(ctx, tape, *params):
        ctx.tape = tape
        ctx.set_materialize_grads(False)      # gradients of gated tensors nobody used stay None (no zero tensors to add downstream)
        tape.capacity = sum((p.numel() + 3) & ~3 for p in params) if tape.splits > 1 else 0
        return tuple(p.detach() for p in params)

    @staticmethod
    def backward(ctx, *grads):
        from .decoder import stamp
        stamp("enc_dgrads_done")
        ctx.tape.flush()
        stamp("enc_wgrads_done")
        return (None,) + grads


def bf16_of(t):
    """The bf16 copy a producer attached to its fp32 rows (`LayerNormRows` with want_bf16), or None."""
    return getattr(t, "_bf16", None)


class ConvRows(torch.autograd.Function):
    """y = dropout( relu?( conv1d_same(x, w) + b ) ) [+ residual] [* rowmask]   on rows tensors.
    Mirrors torch.nn.Conv1d(k, padding=(k-1)//2) + the elementwise tail the reference applies after it.

    bf16 mode, bf16-STORED operand (`xb`: a bf16 copy of x written by its producer, or x itself bf16): the conv is served by the LDS-DMA
    kernel (`conv_dma_kernel`, raw 16-byte staging), its data gradient too (the gate of relu / dropout writes d(pre-activation) as bf16), and
    the weight gradient stages both operands as raw bf16.  `out_bf16`: the output is only ever a conv operand - store it as bf16."""

    @staticmethod
    def forward(ctx, x, w, b, rowmask, residual, relu, mask_out, precision, drop_p, seed, seed_t, tape=None, packs=None, xb=None, out_bf16=False):
        ctx.tape = tape
        ctx.pwt = packs[1] if packs is not None else None         # packs: (forward, transposed) from an ops.PackSet already run this step
        x = x.contiguous()
        R, Cin = x.shape
        O, Ci2, k = w.shape
        assert Ci2 <= Cin and Cin % 4 == 0
        if x.dtype == torch.bfloat16:
            assert precision == ops.BF16, "bf16-stored rows exist in bf16 mode only"
            a = x
        else:
            a = xb.contiguous() if (xb is not None and precision == ops.BF16 and Ci2 % 32 == 0 and Ci2 == Cin and (k > 1 or Ci2 % 64 == 0)) else None
            assert a is None or a.shape == x.shape
        # (bf16 output comes from the bf16-operand kernels only: a shape they do not serve - e.g. 96 or 160 channels with k = 1 - keeps fp32 rows)
        out_bf16 = bool(out_bf16 and (a is not None) and residual is None)
        pw = packs[0] if packs is not None else ops.pack_weight(w.detach(), precision=precision)
        out = torch.empty(R, O, device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
        flags = (ops.F_BIAS if b is not None else 0) | (ops.F_RELU if relu else 0) | (ops.F_MASK if mask_out else 0) | \
                (ops.F_ADD_IN0 if residual is not None else 0) | (ops.F_DROPOUT if drop_p > 0 else 0)
        io = (ops.IO_A_BF16 if a is not None else 0) | (ops.IO_OUT0_BF16 if out_bf16 else 0)
        ops.conv_cl(a if a is not None else x, pw, Ci2, R, lda=Cin, pad=(k - 1) // 2, epi=ops.EPI_LINEAR, flags=flags, n=O,
                    bias=b.detach().contiguous() if b is not None else None, rowmask=rowmask,
                    in0=residual.contiguous() if residual is not None else None, ldi0=O, out0=out, ld0=O,
                    drop_p=drop_p, seed=seed, seed_t=seed_t, io_flags=io)
        gated = relu or drop_p > 0
        assert not (gated and residual is not None), "gate recovery from the output needs out = gated value"
        # what the backward reads of x is the weight gradient's operand: the bf16 rows when there are any
        ctx.save_for_backward(a if a is not None else x, w, out if gated else None, rowmask)
        ctx.cfg = (gated, mask_out, precision, b is not None, residual is not None, Ci2, drop_p, x.dtype, a is not None)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w, out, rowmask = ctx.saved_tensors
        gated, mask_out, precision, has_b, has_res, Ci2, drop_p, x_dtype, bfpath = ctx.cfg
        R, Cin = x.shape
        O, _, k = w.shape
        dy = dy.contiguous()
        bf = torch.bfloat16
        # the data gradient is a conv with K = O: the LDS-DMA kernel takes whole 32-channel chunks, its 1x1 form pairs of them (Project's
        # 160 outputs do not qualify: the register-staged kernel serves it, from fp32 rows - no conversion pass, same 6-7 us)
        bfpath = bfpath and O % 32 == 0 and (k > 1 or O % 64 == 0)
        if gated:                                              # d(pre-activation): relu / dropout cut exactly where out == 0
            dz = torch.empty(R, O, device=dy.device, dtype=bf if bfpath else torch.float32)
            io = (1 if dy.dtype == bf else 0) | (2 if out.dtype == bf else 0) | (4 if bfpath else 0)
            _lib.check(_L().glowtts_gate_bwd_io(dy.data_ptr(), out.data_ptr(), _sp(rowmask) if mask_out else None, dz.data_ptr(), R, O,
                                                1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0, io, _lib.stream()), "glowtts_gate_bwd_io")
        elif mask_out:
            if dy.dtype == torch.float32 and O % 4 == 0:           # the row mask alone (glowtts_gate_bwd_io without a gate tensor): one vectorised launch
                dz = torch.empty(R, O, device=dy.device, dtype=bf if bfpath else torch.float32)
                _lib.check(_L().glowtts_gate_bwd_io(dy.data_ptr(), None, _sp(rowmask), dz.data_ptr(), R, O, 1.0, 4 if bfpath else 0, _lib.stream()),
                           "glowtts_gate_bwd_io")
            else:
                dz = dy * rowmask.unsqueeze(1)
        else:
            dz = dy
        dres = dz if (has_res and ctx.needs_input_grad[4]) else None
        if bfpath and dz.dtype != bf:
            dz = dz.to(bf)
        dx = None
        if ctx.needs_input_grad[0]:
            pwt = ctx.pwt if ctx.pwt is not None else ops.pack_weight(w.detach(), transpose=True, precision=precision)
            dx = torch.empty(R, Cin, device=x.device, dtype=x_dtype) if Cin == Ci2 else torch.zeros(R, Cin, device=x.device, dtype=x_dtype)
            io = (ops.IO_A_BF16 if dz.dtype == bf else 0) | (ops.IO_OUT0_BF16 if x_dtype == bf else 0)
            ops.conv_cl(dz, pwt, O, R, lda=O, pad=(k - 1) // 2, epi=ops.EPI_LINEAR, flags=0, n=Ci2, out0=dx, ld0=Cin, io_flags=io)
        dw = db = None
        if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
            if ctx.tape is not None:                           # deferred: filled by ParamGate.backward (one grouped launch)
                dw = ctx.tape.out((O, Ci2, k), x.device)
                db = ctx.tape.out((O,), x.device) if has_b else None
                ctx.tape.add(dz, x, O, Ci2, k, precision, dw, db)
            else:
                # split-K slices: few for the text encoder's short row counts; 0 = the library's own choice (about two workgroups per CU) for
                # the long ones (patch matrices of the prosody encoder's conv stack: up to 512 k rows against a handful of output tiles)
                wio = (ops.WIO_DY_BF16 if dz.dtype == bf else 0) | (ops.WIO_X_BF16 if x.dtype == bf else 0)
                dw, db = wgrad(dz, x, O, Ci2, k, precision, want_bias=has_b, splits=0 if R > 65536 else min(4, max(1, R // 512)), io_flags=wio)
        return dx, dw, db, None, dres, None, None, None, None, None, None, None, None, None, None


def conv_rows(x, w, b, rowmask, relu=False, mask_out=False, residual=None, precision=ops.BF16, drop_p=0.0, seed=0, seed_t=None, tape=None,
              packs=None, xb=None, out_bf16=False):
    return ConvRows.apply(x, w, b, rowmask, residual, relu, mask_out, precision, float(drop_p), seed, seed_t, tape, packs, xb, out_bf16)


class LayerNormRows(torch.autograd.Function):
    """y = rowmask * dropout( relu?( LayerNorm(a [+ b]) * gamma + beta ) ), eps = 1e-4 (Modules.py:472-475)."""

    @staticmethod
    def forward(ctx, a, b, gamma, beta, rowmask, relu, drop_p, seed, seed_t, yb=None):
        a = a.contiguous()
        R, C = a.shape
        y = torch.empty_like(a)
        stats = torch.empty(R, 2, device=a.device)
        s = torch.empty_like(a) if b is not None else a
        _lib.check(_L().glowtts_layernorm_fwd_io(a.data_ptr(), _sp(b.contiguous() if b is not None else None), s.data_ptr() if b is not None else None,
                                                 gamma.data_ptr(), beta.data_ptr(), _sp(rowmask), y.data_ptr(), stats.data_ptr(), R, C, 1e-4,
                                                 int(relu), float(drop_p), int(seed) & 0xFFFFFFFF, _sp(seed_t), _sp(yb), _lib.stream()),
                   "glowtts_layernorm_fwd_io")
        gated = relu or drop_p > 0
        ctx.save_for_backward(s, stats, gamma, rowmask, y if gated else None)
        ctx.cfg = (gated, float(drop_p), b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        s, stats, gamma, rowmask, y = ctx.saved_tensors
        gated, drop_p, has_b = ctx.cfg
        R, C = s.shape
        dy = dy.contiguous()
        L = _L()
        ds = torch.empty_like(s)
        gb = torch.empty(2 * C, device=s.device)
        scratch = torch.empty(L.glowtts_layernorm_scratch_floats(R, C), device=s.device)
        _lib.check(L.glowtts_layernorm_bwd_io(dy.data_ptr(), _sp(y), s.data_ptr(), stats.data_ptr(), gamma.data_ptr(), _sp(rowmask), ds.data_ptr(),
                                              gb.data_ptr(), scratch.data_ptr(), R, C, int(gated), drop_p, None, None, 1.0, _lib.stream()), "glowtts_layernorm_bwd_io")
        return ds, (ds if has_b else None), gb[:C], gb[C:], None, None, None, None, None, None


def layernorm_rows(a, b, gamma, beta, rowmask, relu=False, drop_p=0.0, seed=0, seed_t=None, want_bf16=False):
    """want_bf16: the result also exists rounded to bf16 (`bf16_of(y)`), written by the same pass - the operand of the convs that read y."""
    if not want_bf16:
        return LayerNormRows.apply(a, b, gamma, beta, rowmask, relu, float(drop_p), seed, seed_t)
    yb = torch.empty(a.shape, device=a.device, dtype=torch.bfloat16)
    y = LayerNormRows.apply(a, b, gamma, beta, rowmask, relu, float(drop_p), seed, seed_t, yb)
    y._bf16 = yb
    return y



def _conv_launch(a, pw, ci, R, k, flags, n, bias, rowmask, out, in0=None, drop_p=0.0, seed=0, seed_t=None, a_bf=True):
    """in0: added to the result (residual) - or, with ops.F_GATE_IN0 in `flags`, the kept output of the relu / dropout layer that gates it."""
    io = (ops.IO_A_BF16 if a_bf else 0) | (ops.IO_OUT0_BF16 if out.dtype == torch.bfloat16 else 0) | \
        (ops.IO_IN0_BF16 if (in0 is not None and in0.dtype == torch.bfloat16) else 0)
    ops.conv_cl(a, pw, ci, R, lda=ci, pad=(k - 1) // 2, epi=ops.EPI_LINEAR,
                flags=flags | (ops.F_ADD_IN0 if (in0 is not None and not flags & ops.F_GATE_IN0) else 0), n=n,
                bias=bias, rowmask=rowmask, in0=in0, ldi0=n, out0=out, ld0=n, drop_p=drop_p, seed=seed, seed_t=seed_t, io_flags=io)


# measured design switches of the block functions (tests flip them; never read from the environment)
FUSE = {"proj_ln": True, "gate_in_dgrad": True, "defer_rel_sum": True, "wgrad_splits": 2}


class FFNBlock(torch.autograd.Function):
    """x2 = LayerNorm_1( Dropout(Conv_1(Dropout(ReLU(Conv_0(x1 * mask))) * mask)) * mask + x1 )   (Modules.py:565-571), bf16-stored rows.

    One autograd node instead of three: its backward is the hand-ordered chain  LayerNorm backward (which also writes Conv_1's gate
    gradient: no separate gate pass) -> Conv_1 data gradient -> gate -> Conv_0 data gradient + the residual branch's gradient in its
    epilogue (no separate add) - 4 launches on the encoder stream's dependent chain instead of 8; weight gradients and the LayerNorm
    parameter gradients are deferred to the tape (grouped launches at the end of the encoder's backward)."""

    @staticmethod
    def forward(ctx, x1, x1b, w0, b0, w1, b1, gamma, beta, rowmask, drop_p, seeds, seed_t, tape, packs0, packs1, next_qkv=None):
        """next_qkv = (packed Wqkv, bias) of the NEXT block's fused Q / K / V conv: the closing LayerNorm then also computes that conv's output (one
        launch, csrc/gemm_cl.hip ln_qkv_kernel) and the call returns it as a third, non-differentiable tensor for `AttentionBlock.forward(qkv_pre=...)`."""
        R, C = x1.shape
        O0, k = w0.shape[0], w0.shape[2]
        dev = x1.device
        bf = torch.bfloat16
        dflag = ops.F_DROPOUT if drop_p > 0 else 0
        h0 = torch.empty(R, O0, device=dev, dtype=bf)
        _conv_launch(x1b, packs0[0], C, R, k, ops.F_BIAS | ops.F_RELU | ops.F_MASK | dflag, O0, b0.detach(), rowmask, h0, drop_p=drop_p, seed=seeds[0], seed_t=seed_t)
        h1 = torch.empty(R, C, device=dev)
        _conv_launch(h0, packs1[0], O0, R, k, ops.F_BIAS | ops.F_MASK | dflag, C, b1.detach(), rowmask, h1, drop_p=drop_p, seed=seeds[1], seed_t=seed_t)
        y, yb, s_, stats = torch.empty_like(x1), torch.empty(R, C, device=dev, dtype=bf), torch.empty_like(x1), torch.empty(R, 2, device=dev)
        qkv = None
        if next_qkv is not None and C == 192 and next_qkv[0].precision == ops.BF16 and next_qkv[0].npad >= 3 * C and R * 3 * C * 4 < 2 ** 31:
            qkv = torch.empty(R, 3 * C, device=dev)
            _lib.check(_L().glowtts_layernorm_qkv(h1.data_ptr(), x1.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rowmask.data_ptr(), s_.data_ptr(),
                                                  stats.data_ptr(), y.data_ptr(), yb.data_ptr(), next_qkv[0].data.data_ptr(), next_qkv[0].npad,
                                                  next_qkv[1].detach().data_ptr(), qkv.data_ptr(), R, C, 1e-4, _lib.stream()), "glowtts_layernorm_qkv")
        else:
            _lib.check(_L().glowtts_layernorm_fwd_io(h1.data_ptr(), x1.data_ptr(), s_.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rowmask.data_ptr(),
                                                     y.data_ptr(), stats.data_ptr(), R, C, 1e-4, 0, 0.0, 0, None, yb.data_ptr(), _lib.stream()),
                       "glowtts_layernorm_fwd_io")
        tape.ln_count += 1
        ctx.save_for_backward(x1b, h0, h1 if drop_p > 0 else None, s_, stats, gamma, rowmask, w0, w1)
        ctx.misc = (tape, packs0[1], packs1[1], float(drop_p))
        # the non-differentiable outputs' "gradients" stay None: materialised, each is a zero-fill launch of [R, C] / [R, 3C] on the encoder's backward chain
        # (two to three 5-us launches per block, twelve blocks: profiles/r04_step_order.txt)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(yb)
        if next_qkv is None:
            return y, yb
        if qkv is None:
            qkv = torch.empty(0, device=dev)                  # (shape contract not met: the next block runs its own conv)
        ctx.mark_non_differentiable(qkv)
        return y, yb, qkv

    @staticmethod
    def backward(ctx, dy, _dyb, _dqkv=None):
        x1b, h0, h1, s_, stats, gamma, rowmask, w0, w1 = ctx.saved_tensors
        tape, pwt0, pwt1, drop_p = ctx.misc
        R, C = s_.shape
        O0, k = w0.shape[0], w0.shape[2]
        dev = s_.device
        bf = torch.bfloat16
        L = _L()
        scale = 1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0
        dy = dy.contiguous()
        ds, dz1 = torch.empty_like(s_), torch.empty(R, C, device=dev, dtype=bf)
        scratch, gb = tape.ln_slot(R, C, dev)
        # LayerNorm_1 backward; dz1 = d(Conv_1 pre-activation): through the conv's dropout gate when there is one, else ds * mask = ds
        _lib.check(L.glowtts_layernorm_bwd_io(dy.data_ptr(), None, s_.data_ptr(), stats.data_ptr(), gamma.data_ptr(), rowmask.data_ptr(), ds.data_ptr(),
                                              None, scratch.data_ptr(), R, C, 0, 0.0, dz1.data_ptr(), _sp(h1), scale, _lib.stream()), "glowtts_layernorm_bwd_io")
        dz0 = torch.empty(R, O0, device=dev, dtype=bf)
        if FUSE["gate_in_dgrad"]:
            # Conv_1 data gradient with the relu / dropout gate of Conv_0's output in its epilogue (h0 == 0 on cut elements and on masked rows)
            _conv_launch(dz1, pwt1, C, R, k, ops.F_GATE_IN0, O0, None, None, dz0, in0=h0, drop_p=drop_p)
        else:
            dh = torch.empty(R, O0, device=dev, dtype=bf)
            _conv_launch(dz1, pwt1, C, R, k, 0, O0, None, None, dh)                                 # Conv_1 data gradient
            _lib.check(L.glowtts_gate_bwd_io(dh.data_ptr(), h0.data_ptr(), rowmask.data_ptr(), dz0.data_ptr(), R, O0, scale, 7, _lib.stream()), "glowtts_gate_bwd_io")
        dx1 = torch.empty(R, C, device=dev)
        _conv_launch(dz0, pwt0, O0, R, k, 0, C, None, None, dx1, in0=ds)                            # Conv_0 data gradient + the residual branch (ds)
        dw0, db0 = tape.out(w0.shape, dev), tape.out((O0,), dev)
        dw1, db1 = tape.out(w1.shape, dev), tape.out((C,), dev)
        tape.add(dz0, x1b, O0, C, k, ops.BF16, dw0, db0)
        tape.add(dz1, h0, C, O0, k, ops.BF16, dw1, db1)
        return dx1, None, dw0, db0, dw1, db1, gb[:C], gb[C:], None, None, None, None, None, None, None, None


class AttentionBlock(torch.autograd.Function):
    """x1 = LayerNorm_0( Dropout(Projection(RPR_attention(QKV(x)))) + x )   (Modules.py:560-562, RPR_MHA.py:82-128), bf16-stored rows.
    Backward chain: LayerNorm backward (+ the projection's dropout gate) -> projection data gradient -> attention backward -> QKV data
    gradient + the residual branch's gradient in its epilogue."""

    @staticmethod
    def forward(ctx, x, xb, wqkv, bqkv, relk, relv, wp, bp, gamma, beta, rowmask, B, Tp, H, win, drop_p, seeds, seed_t, tape, packs_qkv, packs_p, qkv_pre=None,
                rel_gated=False):
        """rel_gated: relk / relv reached this call through the tape's ParamGate - their gradients may then be filled by the tape's flush."""
        R, C = x.shape
        dev = x.device
        bf = torch.bfloat16
        L = _L()
        D = C // H
        dflag = ops.F_DROPOUT if drop_p > 0 else 0
        if qkv_pre is not None and qkv_pre.numel():
            qkv = qkv_pre                                     # computed by the previous block's closing launch (FFNBlock.forward next_qkv)
        else:
            qkv = torch.empty(R, 3 * C, device=dev)
            _conv_launch(xb, packs_qkv[0], C, R, 1, ops.F_BIAS, 3 * C, bqkv.detach(), rowmask, qkv)
        att = torch.empty(R, C, device=dev)
        P = torch.empty(B, H, Tp, Tp, device=dev)
        rk, rv = relk.detach().contiguous(), relv.detach().contiguous()
        _lib.check(L.glowtts_rpr_attention_fwd_prec(qkv.data_ptr(), rk.data_ptr(), rv.data_ptr(), rowmask.data_ptr(), att.data_ptr(), P.data_ptr(),
                                                    B, Tp, H, D, win, float(drop_p), int(seeds[0]) & 0xFFFFFFFF, _sp(seed_t), ops.BF16, _lib.stream()),
                   "glowtts_rpr_attention_fwd_prec")
        y, yb, s_, stats = torch.empty_like(x), torch.empty(R, C, device=dev, dtype=bf), torch.empty_like(x), torch.empty(R, 2, device=dev)
        pw = packs_p[0]
        from .decoder import TUNE
        if FUSE["proj_ln"] and TUNE["enc_proj_ln"] and C == 192 and pw.precision == ops.BF16 and pw.npad >= C and R * C * 4 < 2 ** 31:
            # projection + dropout + residual + LayerNorm as ONE launch (csrc/gemm_cl.hip proj_ln_kernel): one link less in the encoder's forward chain
            proj = torch.empty(R, C, device=dev) if drop_p > 0 else None
            _lib.check(L.glowtts_proj_layernorm(att.data_ptr(), C, pw.data.data_ptr(), pw.npad, bp.detach().data_ptr(), x.data_ptr(), gamma.data_ptr(),
                                                beta.data_ptr(), rowmask.data_ptr(), _sp(proj), s_.data_ptr(),
                                                stats.data_ptr(), y.data_ptr(), yb.data_ptr(), R, C, 1e-4, float(drop_p), int(seeds[1]) & 0xFFFFFFFF,
                                                _sp(seed_t), _lib.stream()), "glowtts_proj_layernorm")
        else:
            proj = torch.empty(R, C, device=dev)
            _conv_launch(att, packs_p[0], C, R, 1, ops.F_BIAS | dflag, C, bp.detach(), rowmask, proj, drop_p=drop_p, seed=seeds[1], seed_t=seed_t, a_bf=False)
            _lib.check(L.glowtts_layernorm_fwd_io(proj.data_ptr(), x.data_ptr(), s_.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rowmask.data_ptr(),
                                                  y.data_ptr(), stats.data_ptr(), R, C, 1e-4, 0, 0.0, 0, None, yb.data_ptr(), _lib.stream()), "glowtts_layernorm_fwd_io")
        tape.ln_count += 1
        defer_rel = bool(FUSE["defer_rel_sum"] and rel_gated)
        if defer_rel:
            tape.att_count += 1
        ctx.save_for_backward(xb, qkv, rk, rv, P, att, proj if drop_p > 0 else None, s_, stats, gamma, rowmask, wqkv, wp, seed_t)
        ctx.misc = (tape, packs_qkv[1], packs_p[1], float(drop_p), B, Tp, H, win, int(seeds[0]) & 0xFFFFFFFF, defer_rel)
        ctx.set_materialize_grads(False)                      # (see FFNBlock.forward)
        ctx.mark_non_differentiable(yb)
        return y, yb

    @staticmethod
    def backward(ctx, dy, _dyb):
        xb, qkv, rk, rv, P, att, proj, s_, stats, gamma, rowmask, wqkv, wp, seed_t = ctx.saved_tensors
        tape, pwt_qkv, pwt_p, drop_p, B, Tp, H, win, aseed, defer_rel = ctx.misc
        R, C = s_.shape
        D = C // H
        dev = s_.device
        bf = torch.bfloat16
        L = _L()
        scale = 1.0 / (1.0 - drop_p) if drop_p > 0 else 1.0
        dy = dy.contiguous()
        ds, dzp = torch.empty_like(s_), torch.empty(R, C, device=dev, dtype=bf)
        scratch, gb = tape.ln_slot(R, C, dev)
        _lib.check(L.glowtts_layernorm_bwd_io(dy.data_ptr(), None, s_.data_ptr(), stats.data_ptr(), gamma.data_ptr(), rowmask.data_ptr(), ds.data_ptr(),
                                              None, scratch.data_ptr(), R, C, 0, 0.0, dzp.data_ptr(), _sp(proj), scale, _lib.stream()), "glowtts_layernorm_bwd_io")
        datt = torch.empty(R, C, device=dev)
        _conv_launch(dzp, pwt_p, C, R, 1, 0, C, None, None, datt)                                     # projection data gradient
        nw = 2 * win + 1
        dS = torch.empty(B, H, Tp, Tp, device=dev)
        dqkv = torch.empty_like(qkv)
        if defer_rel:
            ascr, drel = tape.att_slot(B, Tp, H, D, win, dev)          # (d relK | d relV): summed by the tape's flush, off this chain
        else:
            drel = torch.empty(2, nw, D, device=dev)
            ascr = torch.empty(L.glowtts_rpr_attention_scratch_floats(B, Tp, H, D, win) + 2 * nw * D, device=dev)
        _lib.check(L.glowtts_rpr_attention_bwd_prec(qkv.data_ptr(), rk.data_ptr(), rv.data_ptr(), rowmask.data_ptr(), P.data_ptr(), datt.data_ptr(),
                                                    dS.data_ptr(), dqkv.data_ptr(), None if defer_rel else drel[0].data_ptr(),
                                                    None if defer_rel else drel[1].data_ptr(), ascr.data_ptr(), B, Tp, H, D, win,
                                                    drop_p, aseed, _sp(seed_t), ops.BF16, _lib.stream()), "glowtts_rpr_attention_bwd_prec")
        dx = torch.empty(R, C, device=dev)
        _conv_launch(dqkv, pwt_qkv, 3 * C, R, 1, 0, C, None, None, dx, in0=ds, a_bf=False)            # QKV data gradient + the residual branch (ds)
        dwq, dbq = tape.out(wqkv.shape, dev), tape.out((3 * C,), dev)
        dwp, dbp = tape.out(wp.shape, dev), tape.out((C,), dev)
        tape.add(dqkv, xb, 3 * C, C, 1, ops.BF16, dwq, dbq)
        tape.add(dzp, att, C, C, 1, ops.BF16, dwp, dbp)               # (bf16-stored DY x fp32-stored X: the staged weight-gradient kernel converts X in its loop)
        return (dx, None, dwq, dbq, drel[0].view(1, nw, D), drel[1].view(1, nw, D), dwp, dbp, gb[:C], gb[C:]) + (None,) * 13


class EmbeddingRows(torch.autograd.Function):
    """rows = table[tokens] * scale * mask on the padded rows layout (Modules.py:267)."""

    @staticmethod
    def forward(ctx, tokens, table, rowmask, scale):
        B, T = tokens.shape
        V, C = table.shape
        rows = torch.empty(B * (T + 4), C, device=table.device)
        tokens = tokens.contiguous()
        _lib.check(_L().glowtts_embedding_fwd(tokens.data_ptr(), table.detach().contiguous().data_ptr(), rowmask.data_ptr(), rows.data_ptr(),
                                              B, T, C, scale, _lib.stream()), "glowtts_embedding_fwd")
        ctx.save_for_backward(tokens, rowmask)
        ctx.cfg = (V, C, scale)
        return rows

    @staticmethod
    def backward(ctx, d):
        from .decoder import stamp
        tokens, rowmask = ctx.saved_tensors
        V, C, scale = ctx.cfg
        B, T = tokens.shape
        dt = torch.empty(V, C, device=d.device)
        _lib.check(_L().glowtts_embedding_bwd(tokens.data_ptr(), d.contiguous().data_ptr(), rowmask.data_ptr(), dt.data_ptr(), V, B, T, C, scale,
                                              _lib.stream()), "glowtts_embedding_bwd")
        stamp("enc_bwd_end")
        return None, dt, None, None


class RPRAttention(torch.autograd.Function):
    """Attention core of RPR_MHA.py:95-128 on fused QKV rows [B*Tp, 3*H*D] -> [B*Tp, H*D].  precision: ops.F32 / ops.BF16 (bf16 MFMA
    contractions for Tp <= 128, fp32 softmax; see glowtts_rpr_attention_fwd_prec)."""

    @staticmethod
    def forward(ctx, qkv, relk, relv, rowmask, B, Tp, H, win, drop_p, seed, seed_t, precision=0):
        qkv = qkv.contiguous()
        D = qkv.shape[1] // (3 * H)
        out = torch.empty(B * Tp, H * D, device=qkv.device)
        P = torch.empty(B, H, Tp, Tp, device=qkv.device)
        rk, rv = relk.detach().contiguous(), relv.detach().contiguous()
        _lib.check(_L().glowtts_rpr_attention_fwd_prec(qkv.data_ptr(), rk.data_ptr(), rv.data_ptr(), rowmask.data_ptr(), out.data_ptr(), P.data_ptr(),
                                                       B, Tp, H, D, win, float(drop_p), int(seed) & 0xFFFFFFFF, _sp(seed_t), int(precision), _lib.stream()),
                   "glowtts_rpr_attention_fwd_prec")
        ctx.save_for_backward(qkv, rk, rv, rowmask, P, seed_t)
        ctx.cfg = (B, Tp, H, D, win, float(drop_p), int(seed) & 0xFFFFFFFF, int(precision))
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, rk, rv, rowmask, P, seed_t = ctx.saved_tensors
        B, Tp, H, D, win, drop_p, seed, precision = ctx.cfg
        L = _L()
        dev = qkv.device
        nw = 2 * win + 1
        dS = torch.empty(B, H, Tp, Tp, device=dev)
        dqkv = torch.empty_like(qkv)
        drel = torch.empty(2, nw, D, device=dev)
        drk, drv = drel[0], drel[1]
        scratch = torch.empty(L.glowtts_rpr_attention_scratch_floats(B, Tp, H, D, win) + 2 * nw * D, device=dev)
        _lib.check(L.glowtts_rpr_attention_bwd_prec(qkv.data_ptr(), rk.data_ptr(), rv.data_ptr(), rowmask.data_ptr(), P.data_ptr(),
                                                    dout.contiguous().data_ptr(), dS.data_ptr(), dqkv.data_ptr(), drk.data_ptr(), drv.data_ptr(),
                                                    scratch.data_ptr(), B, Tp, H, D, win, drop_p, seed, _sp(seed_t), precision, _lib.stream()),
                   "glowtts_rpr_attention_bwd_prec")
        return dqkv, drk.view(1, nw, D), drv.view(1, nw, D), None, None, None, None, None, None, None, None, None
