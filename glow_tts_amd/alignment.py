"""Alignment block of GlowTTS.forward (Modules.py:107-122) on the HIP path:
log-prior matrix (batched f32 MFMA GEMM, glowtts_conv_cl) -> Monotonic Alignment Search (glowtts_mas_dp_f32_t)
-> dense 0/1 attentions (glowtts_mas_path_from_idx).  Nothing leaves the device."""
import ctypes
import math

import torch

from . import _lib, ops
from .decoder import PackedBatch, _L

LOG_2PI = math.log(2.0 * math.pi)
_decl = False


def _lib2():
    global _decl
    L = _L()
    if not _decl:
        L.glowtts_logprior_prep.argtypes = [ctypes.c_void_p] * 9 + [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_int)] * 2 + [ctypes.c_void_p]
        L.glowtts_mas_dp_f32_t.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_void_p]
        L.glowtts_expand_fwd.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        L.glowtts_expand_bwd.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        L.glowtts_duration_targets.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
        L.glowtts_mle_loss_fwd.argtypes = [ctypes.c_void_p] * 8 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.glowtts_mle_loss_bwd.argtypes = [ctypes.c_void_p] * 8 + [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.glowtts_expand_pair_targets.argtypes = [ctypes.c_void_p] * 8 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        L.glowtts_prior_loss.argtypes = [ctypes.c_void_p] * 17 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
        L.glowtts_prior_loss_bwd.argtypes = [ctypes.c_void_p] * 10 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        L.glowtts_mse_loss_fwd.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p]
        L.glowtts_mse_loss_bwd.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _decl = True
    return L


@torch.no_grad()
def log_prior_prepare(mean, log_std, token_lengths, mel_lengths, Ty, mel_multiple=1):
    """The operands of the log-prior GEMM that depend on the ENCODER's outputs and the lengths only (csrc/loss_ops.hip logprior_prep_tile_kernel): the packed
    (sigma^-2 | mu sigma^-2) weight image, the per-token constant, the frame mask and the int32 lengths.  `GlowTTS.forward` calls it on the encoder's stream, right
    behind the projection - off the chain between the flow decoder's last launch and its backward - and hands the result to `log_prior_t`."""
    B, Cm, Tx = mean.shape
    L = _lib2()
    dev = mean.device
    mean, log_std = mean.contiguous(), log_std.contiguous()
    npad, kch = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(L.glowtts_logprior_prep(None, None, None, None, None, None, None, None, None, B, Cm, Tx, Ty, int(mel_multiple), ctypes.byref(npad), ctypes.byref(kch), None),
               "glowtts_logprior_prep(size)")
    stride = kch.value * npad.value * 64                                     # bytes per utterance of the packed (sigma^-2 | mu sigma^-2) image
    packed = torch.empty(B * stride, dtype=torch.uint8, device=dev)
    cb, fmask = torch.empty(B, Tx, device=dev), torch.empty(B, Ty, device=dev)
    tx, ty = torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev)
    _lib.check(L.glowtts_logprior_prep(_lib.ptr(mean), _lib.ptr(log_std), _lib.ptr(token_lengths.contiguous()), _lib.ptr(mel_lengths.contiguous()),
                                       _lib.ptr(packed), _lib.ptr(cb), _lib.ptr(fmask), _lib.ptr(tx), _lib.ptr(ty), B, Cm, Tx, Ty, int(mel_multiple), None, None,
                                       _lib.stream()), "glowtts_logprior_prep")
    return {"packed": packed, "cb": cb, "fmask": fmask, "tx": tx, "ty": ty, "npad": npad.value, "kch": kch.value, "stride": stride, "dims": (B, Cm, Tx, Ty)}


@torch.no_grad()
def log_prior_t(mean, log_std, z, token_lengths, mel_lengths, mel_multiple=1, return_lengths=False, z_rows=None, prepared=None):
    """Modules.py:108-114, transposed: returns value_t [B, T_mel, T_tok] = log N(z_y; mean_x, std_x) * mask.
    mean/log_std [B, Cm, Tx], z [B, Cm, Ty] (channel-first like the reference).  Always fp32 (MAS ties).
    mel_multiple: mel lengths are rounded down to a multiple of it on the device (Decoder.Num_Squeeze); return_lengths: also return the
    int32 (token, mel) lengths the kernels used, for maximum_path_t.  prepared: `log_prior_prepare`'s result for these tensors."""
    B, Cm, Tx = mean.shape
    Ty = z.shape[2]
    dev = z.device
    if prepared is None or prepared["dims"] != (B, Cm, Tx, Ty):
        prepared = log_prior_prepare(mean, log_std, token_lengths, mel_lengths, Ty, mel_multiple)
    packed, cb, fmask, tx, ty = (prepared[k] for k in ("packed", "cb", "fmask", "tx", "ty"))
    stride = prepared["stride"]

    class _V:                                                               # (the two sizes the launch below reads)
        def __init__(self, v):
            self.value = v
    npad, kch = _V(prepared["npad"]), _V(prepared["kch"])
    # frames x channels operand: the decoder's own output rows when the caller has them (z_rows [B*(Ty/ns + 2*ROW_PAD), ns*Cm]: a squeezed
    # row holds ns consecutive frames x Cm channels, ROW_PAD pad rows in front of every utterance), else a transposed copy of z
    from . import decoder as _D
    ns = int(mel_multiple)
    if z_rows is not None and Ty % ns == 0 and tuple(z_rows.shape) == (B * (Ty // ns + 2 * _D.ROW_PAD), ns * Cm) and z_rows.is_contiguous():
        zt_ptr, zt_bstride, keep = z_rows.data_ptr() + 4 * _D.ROW_PAD * ns * Cm, (Ty // ns + 2 * _D.ROW_PAD) * ns * Cm, z_rows
    else:
        keep = z.transpose(1, 2).contiguous()                                  # [B,Ty,Cm]
        zt_ptr, zt_bstride = keep.data_ptr(), Ty * Cm
    out = torch.empty(B, Ty, Tx, device=dev)
    a = ops.ConvArgs()
    a.a, a.lda, a.ca1, a.ca, a.apro, a.rows = zt_ptr, Cm, Cm, 2 * Cm, ops.APRO_SQNEG, Ty
    a.w, a.n, a.npad, a.kchunks, a.taps, a.pad, a.precision = packed.data_ptr(), Tx, npad.value, kch.value, 1, 0, ops.F32
    a.epi, a.flags = ops.EPI_LINEAR, ops.F_BIAS | ops.F_MASK | ops.F_COLMASK
    a.bias, a.rowmask, a.out0, a.ld0 = cb.data_ptr(), fmask.data_ptr(), out.data_ptr(), Tx
    a.batch, a.a_bstride, a.w_bstride, a.bias_bstride, a.out_bstride, a.mask_bstride = B, zt_bstride, stride, Tx, Ty * Tx, Ty
    a.ncols_valid = tx.data_ptr()
    a.rows_per_utt = Ty
    _lib.check(_lib.lib().glowtts_conv_cl(ctypes.byref(a), _lib.stream()), "glowtts_conv_cl(log_prior)")
    return (out, tx, ty) if return_lengths else out


@torch.no_grad()
def maximum_path_t(value_t, token_lengths, mel_lengths, max_neg_val=-1e9):
    """value_t [B,Ty,Tx] -> idx [B,Ty] i32 (token aligned to each frame, -1 past the utterance)."""
    B, Ty, Tx = value_t.shape
    idx = torch.empty(B, Ty, dtype=torch.int32, device=value_t.device)
    tx = token_lengths if token_lengths.dtype == torch.int32 else token_lengths.to(torch.int32).contiguous()
    ty = mel_lengths if mel_lengths.dtype == torch.int32 else mel_lengths.to(torch.int32).contiguous()
    _lib.check(_lib2().glowtts_mas_dp_f32_t(_lib.ptr(value_t), _lib.ptr(tx), _lib.ptr(ty), _lib.ptr(idx), None, B, Tx, Ty,
                                            max_neg_val, _lib.stream()), "glowtts_mas_dp_f32_t")
    return idx


class ExpandPrior(torch.autograd.Function):
    """src @ attentions for one-hot-per-frame attentions (Modules.py:120-121): gather by the MAS token index.

    bwd_stream: the stream `src` was produced on when that is not the caller's (the text encoder's, GlowTTS.forward).  The backward - a
    frames -> tokens reduction that only the encoder's backward consumes - is then launched THERE, behind the caller's stream: the flow
    decoder's backward, next on the caller's stream, does not queue behind it (2 x ~10 us per step), and the consumers, which autograd
    replays on that same stream, are ordered behind it by the stream itself."""

    @staticmethod
    def forward(ctx, src, idx, bwd_stream=None):
        src = src.contiguous()
        B, C, Tx = src.shape
        Ty = idx.shape[1]
        out = torch.empty(B, C, Ty, device=src.device)
        _lib.check(_lib2().glowtts_expand_fwd(_lib.ptr(src), _lib.ptr(idx), _lib.ptr(out), B, C, Tx, Ty, _lib.stream()), "glowtts_expand_fwd")
        ctx.save_for_backward(idx)
        ctx.Tx, ctx.bwd_stream = Tx, bwd_stream
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dout = dout.contiguous()
        B, C, Ty = dout.shape
        cur, s = torch.cuda.current_stream(), ctx.bwd_stream

        def run():
            dsrc = torch.empty(B, C, ctx.Tx, device=dout.device)
            _lib.check(_lib2().glowtts_expand_bwd(_lib.ptr(dout), _lib.ptr(idx), _lib.ptr(dsrc), B, C, ctx.Tx, Ty, _lib.stream()), "glowtts_expand_bwd")
            return dsrc
        if s is None or s == cur:
            return run(), None, None
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            dsrc = run()
        dout.record_stream(s)
        idx.record_stream(s)
        return dsrc, None, None


@torch.no_grad()
def duration_targets(idx, token_lengths, Tx):
    """log(sum_t attentions + 1e-7) * token_mask (Modules.py:122) -> [B,1,Tx]."""
    B, Ty = idx.shape
    out = torch.empty(B, Tx, device=idx.device)
    _lib.check(_lib2().glowtts_duration_targets(_lib.ptr(idx), _lib.ptr(token_lengths.contiguous()), _lib.ptr(out), B, Tx, Ty, _lib.stream()),
               "glowtts_duration_targets")
    return out.unsqueeze(1)


def _segment_sums(dout, idx, Tx, stream):
    """glowtts_expand_bwd on `stream` (behind the caller's), see ExpandPrior."""
    dout = dout.contiguous()
    B, C, Ty = dout.shape
    cur = torch.cuda.current_stream()

    def run():
        dsrc = torch.empty(B, C, Tx, device=dout.device)
        _lib.check(_lib2().glowtts_expand_bwd(_lib.ptr(dout), _lib.ptr(idx), _lib.ptr(dsrc), B, C, Tx, Ty, _lib.stream()), "glowtts_expand_bwd")
        return dsrc
    if stream is None or stream == cur:
        return run()
    stream.wait_stream(cur)
    with torch.cuda.stream(stream):
        dsrc = run()
    dout.record_stream(stream)
    idx.record_stream(stream)
    return dsrc


class ExpandPair(torch.autograd.Function):
    """Modules.py:116, 120-122 in ONE launch (round 6): mel_Mean = mean @ attentions, mel_Log_Std = log_Std @ attentions (gathers by the MAS token index),
    log_Duration_Targets = log(sum_t attentions + 1e-7) * token_mask and - want_path - the dense 0/1 attentions themselves -> (mel_mean [B, C, Ty],
    mel_log_std [B, C, Ty], targets [B, Tx], attentions [B, Tx, Ty] or None; targets and attentions carry no gradient, Modules.py:107).  The general backward is
    ExpandPrior's (two segment-sum passes); `MLE_Loss` on exactly these two outputs never runs it - it differentiates through the expansion itself
    (`PriorLoss`).  Ty % 4 != 0: the separate launches."""

    @staticmethod
    def forward(ctx, mean, log_std, idx, token_lengths, bwd_stream=None, want_path=False):
        mean, log_std = mean.contiguous(), log_std.contiguous()
        B, C, Tx = mean.shape
        Ty = idx.shape[1]
        dev = mean.device
        om, ol, tg = torch.empty(B, C, Ty, device=dev), torch.empty(B, C, Ty, device=dev), torch.empty(B, Tx, device=dev)
        path = None
        L = _lib2()
        if Ty % 4 == 0:
            if want_path:
                path = torch.empty(B, Tx, Ty, device=dev)
            _lib.check(L.glowtts_expand_pair_targets(_lib.ptr(mean), _lib.ptr(log_std), _lib.ptr(idx), _lib.ptr(token_lengths.contiguous()), _lib.ptr(om),
                                                     _lib.ptr(ol), _lib.ptr(tg), _lib.ptr(path), B, C, Tx, Ty, _lib.stream()), "glowtts_expand_pair_targets")
        else:
            _lib.check(L.glowtts_expand_fwd(_lib.ptr(mean), _lib.ptr(idx), _lib.ptr(om), B, C, Tx, Ty, _lib.stream()), "glowtts_expand_fwd")
            _lib.check(L.glowtts_expand_fwd(_lib.ptr(log_std), _lib.ptr(idx), _lib.ptr(ol), B, C, Tx, Ty, _lib.stream()), "glowtts_expand_fwd")
            _lib.check(L.glowtts_duration_targets(_lib.ptr(idx), _lib.ptr(token_lengths.contiguous()), _lib.ptr(tg), B, Tx, Ty, _lib.stream()),
                       "glowtts_duration_targets")
            if want_path:
                from .monotonic_align import path_from_idx
                path = path_from_idx(idx, Tx, torch.float32)
        ctx.save_for_backward(idx)
        ctx.Tx, ctx.bwd_stream = Tx, bwd_stream
        ctx.mark_non_differentiable(tg)
        if path is None:
            return om, ol, tg, None
        ctx.mark_non_differentiable(path)
        return om, ol, tg, path

    @staticmethod
    def backward(ctx, dm, dl, _dt, _dp=None):
        (idx,) = ctx.saved_tensors
        return (None if dm is None else _segment_sums(dm, idx, ctx.Tx, ctx.bwd_stream), None if dl is None else _segment_sums(dl, idx, ctx.Tx, ctx.bwd_stream),
                None, None, None, None)


class PriorTag:
    """What `GlowTTS.forward` remembers about the expanded prior it returns: the token-space tensors it was gathered from (with their autograd history) and the
    token index.  `MLE_Loss` finds it on its `mean` / `std` arguments and then differentiates through the expansion in its own backward launch."""

    def __init__(self, mean, log_std, idx):
        self.mean, self.log_std, self.idx = mean, log_std, idx


def tag_prior(mel_mean, mel_log_std, mean, log_std, idx):
    tag = PriorTag(mean, log_std, idx)
    mel_mean._glow_prior, mel_log_std._glow_prior = (tag, 0), (tag, 1)


def prior_tag_of(mel_mean, mel_log_std):
    a, b = getattr(mel_mean, "_glow_prior", None), getattr(mel_log_std, "_glow_prior", None)
    if a is None or b is None or a[0] is not b[0] or (a[1], b[1]) != (0, 1):
        return None
    tag = a[0]
    if tuple(mel_mean.shape[:2]) != tuple(tag.mean.shape[:2]) or mel_mean.shape[2] != tag.idx.shape[1] or tag.mean.shape[2] > 1024:
        return None
    return tag


class MLELoss(torch.autograd.Function):
    """Modules.py:1020-1029 as one reduction + one elementwise backward kernel."""

    @staticmethod
    def forward(ctx, z, mean, log_std, log_dets, lengths, n_squeeze, mel_dim):
        z, mean, log_std, log_dets = z.contiguous(), mean.contiguous(), log_std.contiguous(), log_dets.contiguous()
        dev = z.device
        loss, inv = torch.empty((), device=dev), torch.empty(1, device=dev)
        scratch = torch.empty(1024, device=dev)
        _lib.check(_lib2().glowtts_mle_loss_fwd(_lib.ptr(z), _lib.ptr(mean), _lib.ptr(log_std), _lib.ptr(log_dets), _lib.ptr(lengths.contiguous()),
                                                loss.data_ptr(), inv.data_ptr(), _lib.ptr(scratch), z.numel(), z.shape[0], n_squeeze, mel_dim,
                                                _lib.stream()), "glowtts_mle_loss_fwd")
        ctx.save_for_backward(z, mean, log_std, inv)
        ctx.B = z.shape[0]
        return loss

    @staticmethod
    def backward(ctx, dloss):
        z, mean, log_std, inv = ctx.saved_tensors
        dz, dm, dl = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        dl_ = dloss.contiguous().reshape(1)
        dlogdet = torch.empty(ctx.B, device=z.device)
        _lib.check(_lib2().glowtts_mle_loss_bwd(_lib.ptr(z), _lib.ptr(mean), _lib.ptr(log_std), _lib.ptr(dl_), _lib.ptr(inv), _lib.ptr(dz),
                                                _lib.ptr(dm), _lib.ptr(dl), z.numel(), _lib.ptr(dlogdet), ctx.B, _lib.stream()), "glowtts_mle_loss_bwd")
        return dz, dm, dl, dlogdet, None, None, None


class PriorLoss(torch.autograd.Function):
    """MLE_Loss (Modules.py:1020-1029) on the expanded prior of `GlowTTS.forward`, differentiated THROUGH the expansion (:120-121): the value is MLELoss's (same
    reduction, same bits), the gradients are d z per frame and the gradients of the TOKEN-space mean / log_std as sums over each token's contiguous run of
    frames (csrc/loss_ops.hip prior_bwd_block) - the expanded gradients (2 x 8 MB) and the two segment-sum passes behind them never exist.  Same bits as
    MLELoss + ExpandPrior.  apply(z, mean_tok, log_std_tok, log_dets, lengths, n_squeeze, mel_dim, mel_mean, mel_log_std, idx, seed, owner).

    `seed` (a 0-d device tensor, normally `_lib.one`): the value the caller's backward will be seeded with.  Given, ONE launch writes the loss AND the
    gradients for that seed (glowtts_prior_loss), and a backward that arrives with exactly this tensor launches nothing; any other seed: the separate
    backward launch (glowtts_prior_loss_bwd)."""

    @staticmethod
    def forward(ctx, z, mean_tok, ls_tok, log_dets, lengths, n_squeeze, mel_dim, mel_mean, mel_ls, idx, seed=None, owner=None):
        z, log_dets = z.contiguous(), log_dets.contiguous()
        mean_tok, ls_tok = mean_tok.contiguous(), ls_tok.contiguous()
        dev = z.device
        B, C, Ty = z.shape
        Tx = mean_tok.shape[2]
        loss, inv = torch.empty((), device=dev), torch.empty(1, device=dev)
        scratch = torch.empty(1024, device=dev)
        L = _lib2()
        counter = _lib.counter(dev, "prior_loss", owner) if seed is not None else None
        ctx.seed, ctx.stash = None, None
        if counter is not None:
            dz, dm, dl = torch.empty_like(z), torch.empty_like(mean_tok), torch.empty_like(ls_tok)
            dlogdet = torch.empty(B, device=dev)
            _lib.check(L.glowtts_prior_loss(_lib.ptr(z), _lib.ptr(mel_mean.contiguous()), _lib.ptr(mel_ls.contiguous()), _lib.ptr(mean_tok), _lib.ptr(ls_tok),
                                            _lib.ptr(idx), _lib.ptr(log_dets), _lib.ptr(lengths.contiguous()), _lib.ptr(seed), _lib.ptr(scratch), _lib.ptr(counter),
                                            loss.data_ptr(), inv.data_ptr(), _lib.ptr(dz), _lib.ptr(dm), _lib.ptr(dl), _lib.ptr(dlogdet), B, C, Tx, Ty,
                                            int(n_squeeze), int(mel_dim), _lib.stream()), "glowtts_prior_loss")
            ctx.seed, ctx.stash = seed, (dz, dm, dl, dlogdet)
        else:
            _lib.check(L.glowtts_mle_loss_fwd(_lib.ptr(z), _lib.ptr(mel_mean.contiguous()), _lib.ptr(mel_ls.contiguous()), _lib.ptr(log_dets),
                                              _lib.ptr(lengths.contiguous()), loss.data_ptr(), inv.data_ptr(), _lib.ptr(scratch), z.numel(), B, n_squeeze,
                                              mel_dim, _lib.stream()), "glowtts_mle_loss_fwd")
        ctx.save_for_backward(z, mean_tok, ls_tok, idx, inv)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        z, mean_tok, ls_tok, idx, inv = ctx.saved_tensors
        none = (None,) * 8
        if ctx.seed is not None and dloss.data_ptr() == ctx.seed.data_ptr() and dloss.numel() == 1:
            stash, ctx.stash = ctx.stash, None
            return stash + none
        ctx.stash = None
        B, C, Ty = z.shape
        Tx = mean_tok.shape[2]
        dz, dm, dl = torch.empty_like(z), torch.empty_like(mean_tok), torch.empty_like(ls_tok)
        dlogdet = torch.empty(B, device=z.device)
        _lib.check(_lib2().glowtts_prior_loss_bwd(_lib.ptr(z), _lib.ptr(mean_tok), _lib.ptr(ls_tok), _lib.ptr(idx), _lib.ptr(dloss.contiguous().reshape(1)), _lib.ptr(inv),
                                                  _lib.ptr(dz), _lib.ptr(dm), _lib.ptr(dl), _lib.ptr(dlogdet), B, C, Tx, Ty, _lib.stream()), "glowtts_prior_loss_bwd")
        return (dz, dm, dl, dlogdet) + none


FUSED = {"prior_loss": True,        # tests / A-B runs: False keeps MLELoss + the expansion's own backward
         "seeded": True}            # False: loss nodes never write their gradients in the forward launch (the backward launches of rounds 1-5)
SEEDS = {"mle": None, "rest": None}  # what `LossTerms.backward` will seed the terms with (None = `_lib.one`); a data-parallel trainer sets its frame weight / 1 / world here


def _seed(kind, device):
    if not FUSED["seeded"] or not torch.is_grad_enabled():
        return None
    s = SEEDS.get(kind)
    if s is not None:
        return s if (torch.is_tensor(s) and s.is_cuda and s.numel() == 1 and s.dtype == torch.float32) else None
    return _lib.one(device)


def mle_loss(z, mean, std, log_dets, lengths, n_squeeze, mel_dim, owner=None):
    """`MLE_Loss.forward`: through the expansion when (mean, std) are the expanded prior `GlowTTS.forward` returned (and carry its tag), else on the tensors as given."""
    tag = prior_tag_of(mean, std) if (FUSED["prior_loss"] and torch.is_grad_enabled() and z.is_cuda) else None
    if tag is not None and (tag.mean.requires_grad or tag.log_std.requires_grad):
        return PriorLoss.apply(z, tag.mean, tag.log_std, log_dets, lengths, n_squeeze, mel_dim, mean.detach(), std.detach(), tag.idx, _seed("mle", z.device), owner)
    return MLELoss.apply(z, mean, std, log_dets, lengths, n_squeeze, mel_dim)


class LossTerms:
    """The sum of loss terms a training step differentiates (Train.py:213-216 `loss = MLE + Length (+ Speaker)`, `loss.backward()`), kept as its terms: `backward()`
    seeds every term with its weight directly - no sum node, no ones_like fill in front of the backward - and with the very tensors the terms' forward launches
    were told about (`SEEDS`), so that nodes which wrote their gradients in the forward launch recognise the seed and launch nothing.  The weighted sum itself
    (for the log) is computed when `detach()` asks for it.  Duck-types the tensor methods the step loops use (`backward`, `detach`, `item`)."""

    def __init__(self, terms, seeds=None, after=None):
        self.terms = [t for t in terms if t is not None]
        seeds = list(seeds) if seeds is not None else [None] * len(terms)
        self.seeds = [s for t, s in zip(terms, seeds) if t is not None]
        self.after, self._total = after, None

    def _seed_of(self, t, s):
        if s is not None:
            return s.reshape(()) if s.dim() else s
        o = _lib.one(t.device) if t.is_cuda else None
        return o if o is not None else torch.ones_like(t)

    def backward(self):
        torch.autograd.backward(self.terms, [self._seed_of(t, s) for t, s in zip(self.terms, self.seeds)])

    def detach(self):
        """The weighted sum (computed on first use - the step loops ask for it behind the parameter update, so that its launches and `after`'s do not sit between
        the backward and the gradient norm)."""
        if self._total is None:
            tot = None
            for t, s in zip(self.terms, self.seeds):
                v = t.detach() if s is None else t.detach() * s
                tot = v if tot is None else tot + v
            self._total = tot
            if self.after is not None:
                self.after()
        return self._total

    def item(self):
        return self.detach().item()


class DurationMSE(torch.autograd.Function):
    """The duration loss (Train.py:203-211 `MSELoss()(log_Durations, log_Duration_Targets)`) as one launch per direction.  denom: None = the element count (torch's
    mean); token_lengths [B] i64 = B x the batch's own longest text (trainer.duration_loss: the token axis is padded to a shape bucket); extent = a 0-d device
    tensor holding the longest text of the GLOBAL batch (data parallel).  seed: see PriorLoss - with the constant 1 the forward launch also writes the gradient,
    and a backward seeded with that tensor launches nothing (any other seed scales it in one launch)."""

    @staticmethod
    def forward(ctx, a, target, token_lengths=None, extent=None, seed=None):
        a, target = a.contiguous(), target.contiguous()
        loss = torch.empty((), device=a.device)
        B = int(a.shape[0])
        ext = None if extent is None else extent.to(torch.float32).reshape(1).contiguous()
        tl = None if (token_lengths is None or ext is not None) else token_lengths.contiguous()
        one = _lib.one(a.device) if seed is not None else None
        unit = seed is not None and one is not None and seed.data_ptr() == one.data_ptr()        # (the forward writes the gradient for the seed 1 only)
        da = torch.empty_like(a) if unit else None
        _lib.check(_lib2().glowtts_mse_loss_fwd(_lib.ptr(a), _lib.ptr(target), loss.data_ptr(), a.numel(), 1.0 / a.numel(), _lib.ptr(tl), B, _lib.ptr(ext),
                                                _lib.ptr(da), _lib.stream()), "glowtts_mse_loss_fwd")
        ctx.save_for_backward(a, target, tl, ext)
        ctx.seed, ctx.stash = (seed if unit else None), da
        return loss

    @staticmethod
    def backward(ctx, dloss):
        a, target, tl, ext = ctx.saved_tensors
        stash, ctx.stash = ctx.stash, None
        if ctx.seed is not None and dloss.data_ptr() == ctx.seed.data_ptr() and dloss.numel() == 1:
            return stash, None, None, None, None
        da = torch.empty_like(a)
        _lib.check(_lib2().glowtts_mse_loss_bwd(_lib.ptr(a), _lib.ptr(target), _lib.ptr(dloss.contiguous().reshape(1)), _lib.ptr(da), a.numel(), 1.0 / a.numel(),
                                                _lib.ptr(tl), int(a.shape[0]), _lib.ptr(ext), _lib.stream()), "glowtts_mse_loss_bwd")
        return da, None, None, None, None


def duration_mse(log_durations, log_duration_targets, token_lengths=None, extent=None):
    if not log_durations.is_cuda:
        raise _lib.GlowTTSHipError("glow_tts_amd runs on the GPU only (no CPU fallback)")
    return DurationMSE.apply(log_durations, log_duration_targets.detach(), token_lengths, extent, _seed("rest", log_durations.device))


@torch.no_grad()
def align(mean, log_std, z, token_lengths, mel_lengths, mel_multiple=1):
    """-> (attentions [B,Tx,Ty] float 0/1, idx [B,Ty] i32, value_t).  mel_lengths are rounded down to a multiple of mel_multiple."""
    from .monotonic_align import path_from_idx
    value_t, tx, ty = log_prior_t(mean, log_std, z, token_lengths, mel_lengths, mel_multiple, return_lengths=True)
    idx = maximum_path_t(value_t, tx, ty)
    return path_from_idx(idx, mean.shape[2], torch.float32), idx, value_t
