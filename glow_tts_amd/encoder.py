"""Text encoder (Modules.py:232-284, 438-648; RPR_MHA.py:5-165) - functional form over the reference-named parameters,
channels-last "rows" layout ([B, T + 4, C] with two zero pad rows around every utterance).

Every stage runs in the HIP library: embedding (glowtts_embedding_*), convolutions incl. the fused Q|K|V projection
(glowtts_conv_cl / glowtts_wgrad_cl), LayerNorm with its fused residual / ReLU / dropout / mask (glowtts_layernorm_*), the
relative-position attention core (glowtts_rpr_attention_*), the split of the projection into mean / log_std and the duration predictor's
1-channel projection (glowtts_prior_split_*, glowtts_dur_proj_*).  Remaining PyTorch device ops here are glue only: the
concatenation of the Q/K/V weights and the speaker-vector broadcast of the duration predictor.  Nothing runs on the CPU.

Layout invariant: every stored activation is zero on padded frames and pad rows.  The reference multiplies by the mask only in
front of convolutions (Modules.py:484,565,568,616,643) and at block ends; masking earlier only changes rows the reference
discards as well (all other ops are per-frame)."""
import math

import torch
import torch.nn.functional as F

from . import ops
from .conv_fn import AttentionBlock, EmbeddingRows, FFNBlock, ParamGate, RPRAttention, WgradTape, bf16_of, conv_rows, layernorm_rows

ROW_PAD = 2


def _with_bf16(y, yb):
    """Attaches the bf16 copy of fp32 rows (see conv_fn.bf16_of)."""
    y._bf16 = yb
    return y


_decl = False


def _dur_lib():
    global _decl
    import ctypes
    from . import _lib
    L = _lib.lib()
    if not _decl:
        L.glowtts_dur_proj_supported.argtypes = [ctypes.c_int]
        L.glowtts_dur_proj_fwd.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        L.glowtts_dur_proj_bwd.argtypes = [ctypes.c_void_p] * 9 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        L.glowtts_prior_split_fwd.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        L.glowtts_prior_split_bwd.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        _decl = True
    return L


class DurProj(torch.autograd.Function):
    """Duration_Predictor's Projection (Modules.py:596-618: Conv1d(C -> 1, k = 1) on the masked features, times the mask) on rows, one launch per direction
    (csrc/dur_ops.hip; was mul / sum / add / mul forward and a dozen elementwise / reduce / fill launches at the head of the encoder's backward chain).
    apply(d [B (T + 2 ROW_PAD), C] fp32 rows, weight [1, C, 1], bias [1], mask [B, 1, T], owner) -> [B, 1, T]."""

    @staticmethod
    def forward(ctx, d, w, bias, mask, owner=None):
        from . import _lib
        B, T = int(mask.shape[0]), int(mask.shape[2])
        C = int(d.shape[1])
        d, w, bias, mask = d.contiguous(), w.contiguous(), bias.contiguous(), mask.contiguous()
        out = torch.empty(B, 1, T, device=d.device)
        _lib.check(_dur_lib().glowtts_dur_proj_fwd(_lib.ptr(d), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(mask), _lib.ptr(out), B, T, ROW_PAD, C, _lib.stream()),
                   "glowtts_dur_proj_fwd")
        ctx.save_for_backward(d, w, mask)
        ctx.owner = owner
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        d, w, mask = ctx.saved_tensors
        B, T = int(mask.shape[0]), int(mask.shape[2])
        C = int(d.shape[1])
        dev = d.device
        g = g.contiguous()
        dd, dw, db = torch.empty_like(d), torch.empty_like(w), torch.empty(1, device=dev)
        counter = _lib.counter(dev, "dur_proj", ctx.owner)
        if counter is None:                                   # (first use inside a capture: a counter of this launch's own, zeroed by a fill)
            counter = torch.zeros(1, dtype=torch.int32, device=dev)
        scratch = torch.empty(B * (C + 1), device=dev)
        _lib.check(_dur_lib().glowtts_dur_proj_bwd(_lib.ptr(g), _lib.ptr(mask), _lib.ptr(d), _lib.ptr(w), _lib.ptr(dd), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(scratch),
                                                   _lib.ptr(counter), B, T, ROW_PAD, C, _lib.stream()), "glowtts_dur_proj_bwd")
        return dd, dw, db, None, None


class PriorSplit(torch.autograd.Function):
    """The encoder's projected rows [B (T + 2 ROW_PAD), 2 M] -> mean, log_std [B, M, T] (Modules.py:283-286 `torch.split` of the projection; rows -> channel-first),
    one launch per direction (was two strided copies forward; two zero fills, two slice copies, an add and the transposes backward)."""

    @staticmethod
    def forward(ctx, rows, B, T):
        from . import _lib
        rows = rows.contiguous()
        M = int(rows.shape[1]) // 2
        mean, ls = torch.empty(B, M, T, device=rows.device), torch.empty(B, M, T, device=rows.device)
        _lib.check(_dur_lib().glowtts_prior_split_fwd(_lib.ptr(rows), _lib.ptr(mean), _lib.ptr(ls), B, T, ROW_PAD, M, _lib.stream()), "glowtts_prior_split_fwd")
        ctx.dims = (B, T, M)
        return mean, ls

    @staticmethod
    def backward(ctx, dm, dl):
        from . import _lib
        B, T, M = ctx.dims
        ref = dm if dm is not None else dl
        if ref is None:
            return None, None, None
        dm = None if dm is None else dm.contiguous()
        dl = None if dl is None else dl.contiguous()
        drows = torch.empty(B * (T + 2 * ROW_PAD), 2 * M, device=ref.device)
        _lib.check(_dur_lib().glowtts_prior_split_bwd(_lib.ptr(dm), _lib.ptr(dl), _lib.ptr(drows), B, T, ROW_PAD, M, _lib.stream()), "glowtts_prior_split_bwd")
        return drows, None, None


def from_rows(rows_btc):
    """[B, T+4, C] -> [B,C,T]."""
    return rows_btc[:, ROW_PAD:-ROW_PAD].transpose(1, 2)


class _PackSets:
    """The encoder's weight images as two `ops.PackSet`s (see encoder_forward): `run` launches the first on the current stream, the second on a stream
    of its own, and returns the function that joins it."""

    def __init__(self, sets, sig):
        self.sets, self.sig = sets, sig

    def get(self, key):
        fwd = None
        for st in self.sets:
            if (key, False) in st.packed:
                fwd = st.packed[(key, False)]
                break
        if fwd is None:
            raise KeyError(f"no packed weight image for {key!r}")
        tr = None
        for st in self.sets:
            if (key, True) in st.packed:
                tr = st.packed[(key, True)]
        return fwd, tr

    def run(self, aux):
        """aux: a stream the CALLER has forked from the step's origin stream (a stream forked from the encoder's own - a fork of a fork - made
        hipStreamEndCapture crash on ROCm 7.2), or None: everything on the current stream."""
        self.sets[0].run()
        if len(self.sets) == 1:
            return None
        if aux is None:
            self.sets[1].run()
            return None
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(aux):
            self.sets[1].run()
        return lambda: cur.wait_stream(aux)


def token_masks(token_lengths, T):
    """-> (mask [B, 1, T] float, rowmask [B * (T + 4)]) of a token batch, one launch (glowtts_token_masks)."""
    import ctypes
    from . import _lib
    L = _lib.lib()
    L.glowtts_token_masks.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 2 + [ctypes.c_void_p]
    B = token_lengths.shape[0]
    tl = token_lengths.to(torch.int64).contiguous()
    mask = torch.empty(B, 1, T, device=tl.device)
    rowmask = torch.empty(B * (T + 2 * ROW_PAD), device=tl.device)
    _lib.check(L.glowtts_token_masks(tl.data_ptr(), mask.data_ptr(), rowmask.data_ptr(), B, T, _lib.stream()), "glowtts_token_masks")
    return mask, rowmask


def encoder_forward(P, hp, tokens, mask, speakers=None, prosodies=None, training=False, prefix="layer_Dict.Encoder",
                    precision=1, cache=None, on_prior_ready=None, pack_stream=None, rowmask=None, seed_t=None):
    """Modules.py:262-284 -> mean [B,mel,T], log_std [B,mel,T], log_durations [B,1,T] (channel-first, like the reference).
    pack_stream: a stream forked from the step's origin stream for the second weight-packing launch (see `_PackSets.run`)."""
    e = hp.Encoder
    C = e.Channels
    B, T = tokens.shape
    Tp = T + 2 * ROW_PAD
    # [B*Tp] row mask (0 on pad rows); `rowmask`: already built with the mask (token_masks)
    rmf = rowmask if rowmask is not None else F.pad(mask.squeeze(1), (ROW_PAD, ROW_PAD)).reshape(-1).contiguous()
    dev = tokens.device
    if training and seed_t is None:
        from .decoder import step_seed
        seed_t = step_seed(dev)
    elif not training:
        seed_t = None
    counter = [0]

    def nseed():                                     # a distinct dropout stream per call site (added to the device seed word)
        counter[0] += 7919
        return counter[0]

    # Conv parameters pass through one ParamGate: the convs' weight gradients are then deferred to a few grouped launches at the end
    # of the encoder's backward (conv_fn.WgradTape).  QKV: one fused [3C, C, 1] weight per layer (views of a flat tensor with `cache`).
    Pc = dict(P)
    for i in range(e.Transformer.Stacks):
        a = f"{prefix}.layer_Dict.Transformer.layer_Dict.ANCRDCN_{i}.layer_Dict.Attention"
        names = [a + ".layer_Dict." + n for n in ("Query", "Key", "Value")]
        if cache is not None:
            if a not in cache:
                from .decoder import LeafStack
                cache[a] = (LeafStack([P[n + ".weight"] for n in names], (3,)), LeafStack([P[n + ".bias"] for n in names], (3,)))
            Pc[a + ".QKV.weight"], Pc[a + ".QKV.bias"] = cache[a][0].tensor().view(3 * C, C, 1), cache[a][1].tensor().view(3 * C)
        else:
            Pc[a + ".QKV.weight"] = torch.cat([P[n + ".weight"] for n in names], 0)
            Pc[a + ".QKV.bias"] = torch.cat([P[n + ".bias"] for n in names], 0)
    tape = None
    rel_gated = False
    if torch.is_grad_enabled():
        fused = (".Query.weight", ".Key.weight", ".Value.weight")                # consumed through the fused QKV tensor only
        gated = [k for k, v in Pc.items() if k.startswith(prefix) and k.endswith(".weight") and v.dim() == 3 and v.requires_grad
                 and not k.endswith(fused)]
        gated += [k[:-len(".weight")] + ".bias" for k in gated if (k[:-len(".weight")] + ".bias") in Pc]
        if precision == ops.BF16 and C % 64 == 0:
            # the transformer's LayerNorm parameters too: the block functions (conv_fn.FFNBlock / AttentionBlock) leave their gradients to the
            # tape's one batched reduction, so they must reach autograd's accumulation BEHIND the gate's flush like the conv weights do
            for i in range(e.Transformer.Stacks):
                for j in (0, 1):
                    q = f"{prefix}.layer_Dict.Transformer.layer_Dict.ANCRDCN_{i}.layer_Dict.LayerNorm_{j}"
                    gated += [q + ".weight", q + ".bias"]
                # ... and the attention cores' relative-position embeddings (their gradients are summed by the tape's flush too)
                a_ = f"{prefix}.layer_Dict.Transformer.layer_Dict.ANCRDCN_{i}.layer_Dict.Attention"
                gated += [a_ + ".weight_K", a_ + ".weight_V"]
            rel_gated = True
        if gated:
            tape = WgradTape()
            from .conv_fn import FUSE
            if precision == ops.BF16 and B % max(int(FUSE["wgrad_splits"]), 1) == 0:
                tape.splits, tape.rows_per_utt = max(int(FUSE["wgrad_splits"]), 1), Tp
            for k, v in zip(gated, ParamGate.apply(tape, *[Pc[k] for k in gated])):
                Pc[k] = v

    # every conv weight (and, when training, its transpose for the data gradient) is packed into MFMA tile order: the prenet's forward images - what the
    # first convs wait for - by one small launch on this stream, everything else (the transformer's, the projection's, the duration predictor's images and
    # every transposed image, 95 % of the bytes) by a second launch on a stream of its own that is joined in front of the transformer: the encoder's chain is
    # what the first half of the step waits for, and one launch for all images kept its first conv ~85 us behind the step's start
    packset = None
    if cache is not None:
        items = [(k[:-len(".weight")], v, tr) for k, v in Pc.items()
                 if k.startswith(prefix) and k.endswith(".weight") and v.dim() == 3 and v.shape[0] > 1
                 and not k.endswith((".Query.weight", ".Key.weight", ".Value.weight"))
                 for tr in ((False, True) if torch.is_grad_enabled() else (False,))]
        head = [it for it in items if ".Prenet." in it[0] and not it[2]]
        rest = [it for it in items if not (".Prenet." in it[0] and not it[2])]
        from .decoder import TUNE
        slot = ("packset", precision, torch.is_grad_enabled(), bool(TUNE["enc_pack_split"]))
        packset = cache.get(slot)
        if packset is None or packset.sig != ops.PackSet.signature(items):
            parts = (head, rest) if TUNE["enc_pack_split"] else (items,)
            packset = cache[slot] = _PackSets([ops.PackSet(part, precision) for part in parts if part], ops.PackSet.signature(items))
        pack_join = packset.run(pack_stream)
    else:
        pack_join = None

    # bf16 mode: every LayerNorm also writes its rows as bf16 and the convs that read them (and chains of convs) run on bf16-stored
    # operands - the LDS-DMA kernel instead of the register-staged one (fp32 -> bf16 in the loop); see conv_fn.ConvRows
    bf_rows = precision == ops.BF16 and C % 32 == 0

    def conv(xr, name, relu=False, mask_out=False, residual=None, drop=0.0, out_bf16=False, xb=None):
        p_ = float(drop) if training else 0.0
        xb = (xb if xb is not None else bf16_of(xr)) if bf_rows else None
        out_bf16 = out_bf16 and (xb is not None or xr.dtype == torch.bfloat16)
        return conv_rows(xr, Pc[name + ".weight"], Pc.get(name + ".bias"), rmf, relu=relu, mask_out=mask_out, residual=residual,
                         precision=precision, drop_p=p_, seed=nseed(), seed_t=seed_t, tape=tape,
                         packs=packset.get(name) if packset is not None else None, xb=xb, out_bf16=out_bf16)

    def ln(a, b, name, relu=False, drop=0.0):
        p_ = float(drop) if training else 0.0
        return layernorm_rows(a, b, P[name + ".weight"], P[name + ".bias"], rmf, relu=relu, drop_p=p_, seed=nseed(), seed_t=seed_t,
                              want_bf16=bf_rows)

    from .decoder import stamp
    stamp("enc_fwd_begin")
    x = EmbeddingRows.apply(tokens, P[prefix + ".layer_Dict.Embedding.weight"], rmf, math.sqrt(C))          # :267
    # Prenet :438-489   Conv(x*mask) -> LayerNorm -> ReLU -> Dropout, x3; Conv1x1 + residual; *mask
    res = x
    for i in range(e.Prenet.Stacks):
        q = f"{prefix}.layer_Dict.Prenet.layer_Dict.CLRD_{i}.layer_Dict"
        x = ln(conv(x, q + ".Conv"), None, q + ".LayerNorm", relu=True, drop=e.Prenet.Dropout_Rate)
    x = conv(x, prefix + ".layer_Dict.Prenet.layer_Dict.Conv1x1", mask_out=True, residual=res)
    if pack_join is not None:
        pack_join()                                                    # the second launch's images are needed from here on
    blocks = bf_rows and C % 64 == 0 and tape is not None and packset is not None
    if blocks and bf16_of(x) is None:
        x = _with_bf16(x, x.detach().to(torch.bfloat16))               # (the prenet's last conv writes fp32 rows)
    # Transformer :492-573
    dr = e.Transformer.Dropout_Rate
    H = e.Transformer.Attention.Heads
    win = e.Transformer.Attention.Window_Size
    qkv_pre = None
    from .decoder import TUNE
    for i in range(e.Transformer.Stacks):
        q = f"{prefix}.layer_Dict.Transformer.layer_Dict.ANCRDCN_{i}.layer_Dict"
        a = q + ".Attention"
        if blocks:
            # bf16-stored rows, training: two autograd nodes per layer whose backward chains are hand-ordered (conv_fn.AttentionBlock / FFNBlock)
            pdr = float(dr) if training else 0.0
            x = _with_bf16(*AttentionBlock.apply(x, bf16_of(x), Pc[a + ".QKV.weight"], Pc[a + ".QKV.bias"], Pc[a + ".weight_K"], Pc[a + ".weight_V"],
                                                 Pc[a + ".layer_Dict.Projection.weight"], Pc[a + ".layer_Dict.Projection.bias"],
                                                 Pc[q + ".LayerNorm_0.weight"], Pc[q + ".LayerNorm_0.bias"], rmf, B, Tp, H, win, pdr, (nseed(), nseed()), seed_t,
                                                 tape, packset.get(a + ".QKV"), packset.get(a + ".layer_Dict.Projection"), qkv_pre, rel_gated))
            # the block's closing LayerNorm also computes the NEXT block's fused Q / K / V conv (one launch less on the forward chain per block)
            nxt = None
            if i + 1 < e.Transformer.Stacks and TUNE["enc_ln_qkv"]:
                an = f"{prefix}.layer_Dict.Transformer.layer_Dict.ANCRDCN_{i + 1}.layer_Dict.Attention"
                nxt = (packset.get(an + ".QKV")[0], Pc[an + ".QKV.bias"])
            out = FFNBlock.apply(x, bf16_of(x), Pc[q + ".Conv_0.weight"], Pc[q + ".Conv_0.bias"], Pc[q + ".Conv_1.weight"], Pc[q + ".Conv_1.bias"],
                                 Pc[q + ".LayerNorm_1.weight"], Pc[q + ".LayerNorm_1.bias"], rmf, pdr, (nseed(), nseed()), seed_t, tape,
                                 packset.get(q + ".Conv_0"), packset.get(q + ".Conv_1"), nxt)
            x, qkv_pre = _with_bf16(out[0], out[1]), (out[2] if nxt is not None else None)
            continue
        qkv = conv(x, a + ".QKV")                                                                            # RPR_MHA.py:82-84 (one fused 1x1 conv)
        att = RPRAttention.apply(qkv, P[a + ".weight_K"], P[a + ".weight_V"], rmf, B, Tp, H, win,
                                 float(dr) if training else 0.0, nseed(), seed_t, precision)                 # RPR_MHA.py:95-128
        att = conv(att, a + ".layer_Dict.Projection", drop=dr)                                               # :93 + Dropout :561
        x = ln(att, x, q + ".LayerNorm_0")                                                                   # :562
        h = conv(x, q + ".Conv_0", relu=True, mask_out=True, drop=dr, out_bf16=True)                         # :565-567 (only Conv_1 reads it)
        h = conv(h, q + ".Conv_1", mask_out=True, drop=dr)                                                   # :568-569 (x*mask at :571)
        x = ln(h, x, q + ".LayerNorm_1")                                                                     # :571
    proj = conv(x, prefix + ".layer_Dict.Project", mask_out=True)
    stamp("enc_fwd_project")
    M = hp.Sound.Mel_Dim
    if proj.is_cuda and proj.dtype == torch.float32 and proj.shape[1] == 2 * M and 32 * (2 * M + 1) * 4 <= 64 * 1024:
        mean, log_std = PriorSplit.apply(proj, B, Tp - 2 * ROW_PAD)         # (here, on the encoder stream: the consumers need dense channel-first tensors)
    else:
        proj = from_rows(proj.view(B, Tp, -1))
        mean, log_std = proj[:, :M].contiguous(), proj[:, M:].contiguous()
    if on_prior_ready is not None:
        on_prior_ready(mean, log_std)         # the log-prior / MAS side of the step can start: what follows (duration predictor) is only needed by the losses
    # Duration predictor on detached features (:277-282, 602-618)
    d, db = x.detach(), bf16_of(x)
    cond = None
    if speakers is not None:
        cond = speakers.detach()
    if prosodies is not None:
        cond = prosodies.detach() if cond is None else cond + prosodies.detach()
    if cond is not None:
        d = torch.cat([d.view(B, Tp, C), cond.unsqueeze(1).expand(-1, Tp, -1) * rmf.view(B, Tp, 1)], dim=2).reshape(B * Tp, -1).contiguous()
        db = None
    dp = e.Duration_Predictor
    for i in range(dp.Stacks):
        q = f"{prefix}.layer_Dict.Duration_Predictor.layer_Dict.CRND_{i}.layer_Dict.Conv"
        d = conv(d, q, relu=True, mask_out=True, drop=dp.Dropout_Rate, xb=db if i == 0 else None, out_bf16=i + 1 < dp.Stacks)
    # Projection to one channel: N = 1 is not a GEMM; a masked dot product per frame (multiply + row reduction: rocBLAS' gemv took
    # 82 us for these 3.5k x 256 rows, 5x the two elementwise kernels)
    wq = prefix + ".layer_Dict.Duration_Predictor.layer_Dict.Projection"
    if d.is_cuda and d.dtype == torch.float32 and Pc[wq + ".weight"].dtype == torch.float32 and _dur_lib().glowtts_dur_proj_supported(int(d.shape[1])):
        log_dur = DurProj.apply(d, Pc[wq + ".weight"], Pc[wq + ".bias"], mask, cache)
    else:
        log_dur = ((d * Pc[wq + ".weight"][0, :, 0]).sum(-1) + Pc[wq + ".bias"]).view(B, Tp)[:, ROW_PAD:-ROW_PAD].unsqueeze(1) * mask
    return mean, log_std, log_dur
