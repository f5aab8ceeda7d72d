"""Conditioning encoders of the PE / GR modes - NOT part of the hot path (each produces one small
per-utterance tensor, < 1 % of the FLOPs): the GST `Prosody_Encoder` (Modules.py:312-385), the adversarial `Speaker_Classifier_GR` behind the
gradient-reversal layer (Modules.py:407-435, Gradient_Reversal_Layer.py:6-35) and the `Pitch_Interpolater` (Modules.py:387-405).  Plain
PyTorch-ROCm modules with the reference's parameter names, so PE- and GR-mode checkpoints load strictly; they feed the HIP decoder / encoder
as conditioning vectors.  Parity: tests/test_gpu_modes.py against the golden vectors of the reference (tiny_pe.npz, tiny_gr.npz)."""
import ctypes
import math

import torch


class _GRUFunction(torch.autograd.Function):
    """torch.nn.GRU (one layer, batch_first, h0 = 0) on the GPU: the input projection and the weight-gradient GEMMs are torch matmuls, the
    recurrence is one launch per direction of glowtts_gru_fwd / glowtts_gru_bwd (MIOpen issues ~30 small launches per time step)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        import ctypes
        from . import _lib
        L = _lib.lib()
        L.glowtts_gru_fwd.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
        B, T, _ = x.shape
        H = w_hh.shape[1]
        gi = torch.addmm(b_ih, x.reshape(B * T, -1), w_ih.t()).contiguous()
        hs, keep = torch.empty(B, T, H, device=x.device), torch.empty(B, T, 4 * H, device=x.device)
        w_hh_c, b_hh_c = w_hh.contiguous(), b_hh.contiguous()
        _lib.check(L.glowtts_gru_fwd(_lib.ptr(gi), _lib.ptr(w_hh_c), _lib.ptr(b_hh_c), _lib.ptr(hs), _lib.ptr(keep), B, T, H, _lib.stream()), "glowtts_gru_fwd")
        ctx.save_for_backward(x, w_ih, w_hh_c, hs, keep)
        return hs

    @staticmethod
    def backward(ctx, dhs):
        import ctypes
        from . import _lib
        x, w_ih, w_hh, hs, keep = ctx.saved_tensors
        L = _lib.lib()
        L.glowtts_gru_bwd.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
        B, T, H = hs.shape
        dgi, dgh = torch.empty(B, T, 3 * H, device=x.device), torch.empty(B, T, 3 * H, device=x.device)
        _lib.check(L.glowtts_gru_bwd(_lib.ptr(dhs.contiguous()), _lib.ptr(hs), _lib.ptr(keep), _lib.ptr(w_hh), _lib.ptr(dgi), _lib.ptr(dgh), B, T, H,
                                     _lib.stream()), "glowtts_gru_bwd")
        dgi2, dgh2 = dgi.view(B * T, 3 * H), dgh.view(B * T, 3 * H)
        hprev = torch.cat([hs.new_zeros(B, 1, H), hs[:, :-1]], dim=1).reshape(B * T, H)
        dx = (dgi2 @ w_ih).view_as(x) if ctx.needs_input_grad[0] else None
        return dx, dgi2.t() @ x.reshape(B * T, -1), dgh2.t() @ hprev, dgi2.sum(0), dgh2.sum(0)


class _PackJob(ctypes.Structure):
    """glowtts_c2d_pack_job (include/glowtts_hip.h)"""
    _fields_ = [("w", ctypes.c_void_p), ("img", ctypes.c_void_p)] + [(k, ctypes.c_int) for k in ("Ci", "Co", "cls", "N", "K", "npad", "kchunks", "block0")]


class _ReduceJob(ctypes.Structure):
    """glowtts_c2d_reduce_job"""
    _fields_ = [("partial", ctypes.c_void_p), ("dw", ctypes.c_void_p)] + [(k, ctypes.c_int) for k in ("splits", "Ci", "Co", "block0")]


_C2D_DECLARED = []


def _c2d():
    from . import _lib
    L = _lib.lib()
    if not _C2D_DECLARED:
        vp, ci, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        L.glowtts_conv3x3s2_supported.argtypes = [ci] * 5
        L.glowtts_conv3x3s2_image_bytes.argtypes = [ci, ci, ci, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        L.glowtts_conv3x3s2_pack_job_init.argtypes = [ctypes.POINTER(_PackJob), vp, ci, ci, ci, ci, vp, ci, ctypes.POINTER(ci)]
        L.glowtts_conv3x3s2_pack.argtypes = [vp, ci, ci, ci, vp]
        L.glowtts_conv3x3s2_fwd.argtypes = [vp, vp, vp, vp] + [ci] * 7 + [vp]
        L.glowtts_conv3x3s2_dgrad.argtypes = [vp, ctypes.POINTER(vp), vp, vp] + [ci] * 6 + [vp]
        L.glowtts_conv3x3s2_wgrad_scratch_floats.argtypes = [ci] * 5
        L.glowtts_conv3x3s2_wgrad_scratch_floats.restype = i64
        L.glowtts_conv3x3s2_wgrad.argtypes = [vp, vp, vp] + [ci] * 6 + [ctypes.POINTER(ci), vp]
        L.glowtts_conv3x3s2_wgrad_reduce.argtypes = [ctypes.POINTER(_ReduceJob), ci, vp]
        _C2D_DECLARED.append(True)
    return L


class _ConvImages:
    """Weight images of a conv stack (forward image per layer with Ci > 1, four data-gradient class images when gradients are needed) in one
    buffer, plus the device job table of the ONE launch that rewrites them from the current weights (static while the parameters stay in place)."""

    def __init__(self, weights, precision, need_bwd):
        from . import _lib
        L = _c2d()
        self.key = (tuple(w.data_ptr() for w in weights), int(precision), bool(need_bwd))
        dev = weights[0].device
        sizes = []
        for w in weights:
            Co, Ci = int(w.shape[0]), int(w.shape[1])
            fwd, dg = ctypes.c_int64(0), (ctypes.c_int64 * 4)()
            _lib.check(L.glowtts_conv3x3s2_image_bytes(Ci, Co, precision, ctypes.byref(fwd), dg), "glowtts_conv3x3s2_image_bytes")
            sizes.append((int(fwd.value) if Ci > 1 else 0, [int(v) if (need_bwd and Ci > 1) else 0 for v in dg]))
        total = sum(f + sum(d) for f, d in sizes)
        self.buf = torch.empty(max(total, 16), dtype=torch.uint8, device=dev)
        base, off = self.buf.data_ptr(), 0
        self.fwd, self.dgrad, jobs, block = [], [], [], 0
        for w, (f, d) in zip(weights, sizes):
            Co, Ci = int(w.shape[0]), int(w.shape[1])
            fp, dps = None, None
            todo = []
            if f:
                fp = base + off
                off += f
                todo.append((-1, fp))
            if any(d):
                dps = []
                for cls in range(4):
                    dps.append(base + off)
                    todo.append((cls, base + off))
                    off += d[cls]
            for cls, img in todo:
                job, nb = _PackJob(), ctypes.c_int(0)
                _lib.check(L.glowtts_conv3x3s2_pack_job_init(ctypes.byref(job), w.data_ptr(), Ci, Co, cls, precision, img, block, ctypes.byref(nb)),
                           "glowtts_conv3x3s2_pack_job_init")
                block += nb.value
                jobs.append(job)
            self.fwd.append(fp)
            self.dgrad.append(dps)
        self.njobs, self.blocks = len(jobs), block
        self.table = None
        if jobs:
            raw = b"".join(bytes(j) for j in jobs)
            self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.precision = precision

    def pack(self):
        from . import _lib
        if self.table is not None:
            _lib.check(_c2d().glowtts_conv3x3s2_pack(self.table.data_ptr(), self.njobs, self.blocks, self.precision, _lib.stream()), "glowtts_conv3x3s2_pack")


class _ConvStack(torch.autograd.Function):
    """The reference encoder's Conv2d(3x3, stride 2, padding 1, no bias) + ReLU stack (Modules.py:320-333, 366-368) as direct HIP kernels on
    channels-last activations (csrc/conv2d_ops.hip): 1 weight-image launch + one launch per layer forward; backward one data-gradient launch
    (four parity classes, the producing layer's ReLU in its epilogue) and one weight-gradient launch per layer + ONE deterministic reduction for all
    layers.  apply(mels [B, Mel, T], images, precision, *weights [Co, Ci, 3, 3]) -> [B, Mel', T', C] (channels-last)."""

    @staticmethod
    def forward(ctx, mels, images, precision, *weights):
        from . import _lib
        L = _c2d()
        x = mels.contiguous()                                  # [B][H = Mel][W = T] == channels-last with C = 1
        B, H, W = x.shape
        images.pack()
        acts, shapes = [x], []
        s = _lib.stream()
        for l, w in enumerate(weights):
            Co, Ci = int(w.shape[0]), int(w.shape[1])
            Ho, Wo = (H + 1) // 2, (W + 1) // 2
            y = torch.empty(B, Ho, Wo, Co, device=x.device)
            _lib.check(L.glowtts_conv3x3s2_fwd(acts[-1].data_ptr(), w.data_ptr(), images.fwd[l], y.data_ptr(), B, H, W, Ci, Co, 1, precision, s),
                       "glowtts_conv3x3s2_fwd")
            shapes.append((H, W, Ci, Co))
            acts.append(y)
            H, W = Ho, Wo
        ctx.save_for_backward(*acts, *weights)
        ctx.cfg = (images, precision, shapes, B)
        from . import decoder
        decoder.stamp("pros_convs_fwd_end")
        return acts[-1]

    @staticmethod
    def backward(ctx, dout):
        from . import _lib
        L = _c2d()
        images, precision, shapes, B = ctx.cfg
        n = len(shapes)
        acts, weights = ctx.saved_tensors[:n + 1], ctx.saved_tensors[n + 1:]
        s = _lib.stream()
        from . import decoder
        decoder.stamp("pros_convs_bwd_begin")
        # the last layer's ReLU (tiny: [B, 2, 13, 128] at the default sizes): d pre = d out where the output is not zero - one launch
        dout = dout.contiguous()
        dpre = torch.empty_like(dout)
        L.glowtts_gate_bwd.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        _lib.check(L.glowtts_gate_bwd(_lib.ptr(dout), _lib.ptr(acts[n]), None, _lib.ptr(dpre), dout.numel() // dout.shape[-1], int(dout.shape[-1]), 1.0, s),
                   "glowtts_gate_bwd")
        jobs = (_ReduceJob * n)()
        keep, grads = [], [None] * n
        for l in range(n - 1, -1, -1):
            H, W, Ci, Co = shapes[l]
            x = acts[l]
            partial = torch.empty(int(L.glowtts_conv3x3s2_wgrad_scratch_floats(B, H, W, Ci, Co)), device=x.device)
            splits = ctypes.c_int(0)
            _lib.check(L.glowtts_conv3x3s2_wgrad(x.data_ptr(), dpre.data_ptr(), partial.data_ptr(), B, H, W, Ci, Co, precision, ctypes.byref(splits), s),
                       "glowtts_conv3x3s2_wgrad")
            dw = torch.empty_like(weights[l])
            jobs[l].partial, jobs[l].dw, jobs[l].splits, jobs[l].Ci, jobs[l].Co = partial.data_ptr(), dw.data_ptr(), splits.value, Ci, Co
            keep.append(partial)
            grads[l] = dw
            if l > 0:
                dx = torch.empty_like(x)
                imgs = (ctypes.c_void_p * 4)(*images.dgrad[l])
                _lib.check(L.glowtts_conv3x3s2_dgrad(dpre.data_ptr(), imgs, x.data_ptr(), dx.data_ptr(), B, H, W, Ci, Co, precision, s),
                           "glowtts_conv3x3s2_dgrad")
                dpre = dx
        _lib.check(L.glowtts_conv3x3s2_wgrad_reduce(jobs, n, s), "glowtts_conv3x3s2_wgrad_reduce")
        from . import decoder
        decoder.stamp("pros_convs_bwd_end")
        return (None, None, None) + tuple(grads)


def conv_stack_supported(convs, mels):
    """The HIP conv stack takes: 3x3 / stride 2 / padding 1 / no bias, first layer one input channel, every other layer 32 / 64 / 128 channels in
    and out (the reference's defaults, Hyper_Parameters.yaml Prosody_Encoder.Reference_Encoder); anything else runs torch's Conv2d."""
    if not (mels.is_cuda and mels.dtype == torch.float32 and mels.dim() == 3):
        return False
    if not all(c.kernel_size == (3, 3) and c.stride == (2, 2) and c.padding == (1, 1) and c.bias is None and c.weight.dtype == torch.float32 for c in convs):
        return False
    L = _c2d()
    B, H, W = (int(v) for v in mels.shape)
    for c in convs:
        if not L.glowtts_conv3x3s2_supported(B, H, W, c.in_channels, c.out_channels):
            return False
        H, W = (H + 1) // 2, (W + 1) // 2
    return convs[0].in_channels == 1 and all(c.in_channels > 1 for c in convs[1:])


def conv_stack_hip(convs, mels, precision, cache=None):
    """mels [B, Mel, T] -> [B, T', C * Mel'] (the GRU's input, feature = c * Mel' + h as in the reference's reshape at Modules.py:369)."""
    weights = [c.weight for c in convs]
    need_bwd = torch.is_grad_enabled() and any(w.requires_grad for w in weights)
    key = (tuple(w.data_ptr() for w in weights), int(precision), bool(need_bwd))
    images = cache.get("images") if cache is not None else None
    if images is None or images.key != key:
        images = _ConvImages([w.detach() for w in weights], int(precision), need_bwd)
        if cache is not None:
            cache["images"] = images
    x = _ConvStack.apply(mels, images, int(precision), *weights)
    B, H, W, C = x.shape
    return x.permute(0, 2, 3, 1).reshape(B, W, C * H)


_GST_DECLARED = []


def _gst():
    from . import _lib
    L = _lib.lib()
    if not _GST_DECLARED:
        vp, ci, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        L.glowtts_gst_supported.argtypes = [ci] * 7
        L.glowtts_gst_keep_floats.argtypes = [ci] * 6
        L.glowtts_gst_keep_floats.restype = i64
        L.glowtts_gst_fwd.argtypes = [vp, vp, ci] + [vp] * 13 + [ci] * 7 + [vp]
        L.glowtts_gst_bwd.argtypes = [vp, vp, vp, ci] + [vp] * 18 + [ci] * 7 + [vp]
        _GST_DECLARED.append(True)
    return L


class _GSTTail(torch.autograd.Function):
    """Modules.py:371-385 behind the GRU - the state at each utterance's last valid step attends over tanh(gst_Tokens) - as two launches forward (token keys / values;
    one workgroup per utterance for gather, query, four-head softmax attention and output projection) and three backward (csrc/gst_ops.hip).  PyTorch ran ~35 / ~45
    launches here, on the chain in front of the flow decoder / in front of the conv stack's backward.
    apply(hs [B, T', G], lengths [B], stride_prod, heads, tokens [I, NT], Wq, bq, Wk, bk, Wv, bv, Wp, bp) -> [B, C]."""

    @staticmethod
    def forward(ctx, hs, lengths, stride_prod, heads, tokens, Wq, bq, Wk, bk, Wv, bv, Wp, bp):
        from . import _lib
        L = _gst()
        hs = hs.contiguous()
        B, Tp, G = hs.shape
        I, NT = tokens.shape
        C = Wq.shape[0]
        dev = hs.device
        c = lambda t: None if t is None else t.contiguous()
        tokens, Wq, bq, Wk, bk, Wv, bv, Wp, bp = (c(t) for t in (tokens, Wq, bq, Wk, bk, Wv, bv, Wp, bp))
        lengths = lengths.to(torch.int64).contiguous()
        K, V, out = torch.empty(C, NT, device=dev), torch.empty(C, NT, device=dev), torch.empty(B, C, device=dev)
        keep = torch.empty(int(L.glowtts_gst_keep_floats(B, G, C, heads, NT, I)), device=dev)
        _lib.check(L.glowtts_gst_fwd(_lib.ptr(hs), _lib.ptr(lengths), int(stride_prod), _lib.ptr(tokens), _lib.ptr(Wq), _lib.ptr(bq), _lib.ptr(Wk), _lib.ptr(bk),
                                     _lib.ptr(Wv), _lib.ptr(bv), _lib.ptr(Wp), _lib.ptr(bp), _lib.ptr(K), _lib.ptr(V), _lib.ptr(out), _lib.ptr(keep),
                                     B, Tp, G, C, int(heads), NT, I, _lib.stream()), "glowtts_gst_fwd")
        ctx.save_for_backward(keep, lengths, tokens, Wq, Wk, Wv, Wp, K, V)
        ctx.cfg = (B, Tp, G, C, int(heads), NT, I, int(stride_prod), bq is not None, bk is not None, bv is not None, bp is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import _lib
        L = _gst()
        keep, lengths, tokens, Wq, Wk, Wv, Wp, K, V = ctx.saved_tensors
        B, Tp, G, C, H, NT, I, sp, hbq, hbk, hbv, hbp = ctx.cfg
        dev = dout.device
        dout = dout.contiguous()
        dhs = torch.empty(B, Tp, G, device=dev)
        scratch = torch.empty(B * (2 * C + H * NT) + 2 * C * NT, device=dev)
        dWq, dWk, dWv, dWp, dtok = torch.empty_like(Wq), torch.empty_like(Wk), torch.empty_like(Wv), torch.empty_like(Wp), torch.empty_like(tokens)
        db = [torch.empty(C, device=dev) if has else None for has in (hbq, hbk, hbv, hbp)]
        _lib.check(L.glowtts_gst_bwd(_lib.ptr(dout), _lib.ptr(keep), _lib.ptr(lengths), sp, _lib.ptr(tokens), _lib.ptr(Wq), _lib.ptr(Wk), _lib.ptr(Wv), _lib.ptr(Wp),
                                     _lib.ptr(K), _lib.ptr(V), _lib.ptr(dhs), _lib.ptr(scratch), _lib.ptr(dWq), _lib.ptr(db[0]), _lib.ptr(dWk), _lib.ptr(db[1]),
                                     _lib.ptr(dWv), _lib.ptr(db[2]), _lib.ptr(dWp), _lib.ptr(db[3]), _lib.ptr(dtok), B, Tp, G, C, H, NT, I, _lib.stream()),
                   "glowtts_gst_bwd")
        return dhs, None, None, None, dtok, dWq, db[0], dWk, db[1], dWv, db[2], dWp, db[3]


class _ConvBlock(torch.nn.Sequential):
    def __init__(self, cin, cout, k, stride):
        super().__init__()
        conv = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=(k - 1) // 2, bias=False)
        torch.nn.init.kaiming_uniform_(conv.weight, nonlinearity="relu")           # Modules.py:1010-1014
        self.add_module("Conv", conv)
        self.add_module("ReLU", torch.nn.ReLU(inplace=True))


def _conv1x1(conv, x):
    """torch.nn.Conv1d(kernel_size = 1) on [B, C, T] as a matmul with the module's own parameters."""
    y = torch.matmul(conv.weight.squeeze(-1), x)
    return y if conv.bias is None else y + conv.bias.view(1, -1, 1)


class _Attention(torch.nn.Module):
    """RPR_Multihead_Attention without relative positions (Modules.py:349-355; RPR_MHA.py:69-128 with masks = None)."""

    def __init__(self, qc, kc, calc, out, heads):
        super().__init__()
        self.heads = heads
        self.layer_Dict = torch.nn.ModuleDict({
            "Query": torch.nn.Conv1d(qc, calc, 1), "Key": torch.nn.Conv1d(kc, calc, 1), "Value": torch.nn.Conv1d(kc, calc, 1),
            "Projection": torch.nn.Conv1d(calc, out, 1), "Dropout": torch.nn.Dropout(0.0)})
        for n in ("Query", "Key", "Value"):
            torch.nn.init.xavier_uniform_(self.layer_Dict[n].weight)

    def forward(self, queries, keys):
        B, _, Tq = queries.shape
        Tk = keys.shape[2]
        H = self.heads
        # the 1x1 Conv1d layers as batched matmuls (rocBLAS): MIOpen serves these shapes with its naive direct kernels (~0.5 ms a call, and
        # which solver a captured hipGraph got depended on when it was captured)
        q = _conv1x1(self.layer_Dict["Query"], queries)
        k = _conv1x1(self.layer_Dict["Key"], keys)
        v = _conv1x1(self.layer_Dict["Value"], keys)
        D = q.shape[1] // H
        q, k, v = (t.view(B, H, D, -1).transpose(2, 3) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(2, 3) / math.sqrt(D), dim=-1) @ v
        return _conv1x1(self.layer_Dict["Projection"], a.transpose(2, 3).reshape(B, H * D, Tq))


class Prosody_Encoder(torch.nn.Module):
    def __init__(self, hp):
        super().__init__()
        pe = hp.Prosody_Encoder
        self.strides = list(pe.Reference_Encoder.Conv.Strides)
        self.layer_Dict = torch.nn.ModuleDict()
        cin, height = 1, hp.Sound.Mel_Dim
        for i, (k, c, s) in enumerate(zip(pe.Reference_Encoder.Conv.Kernel_Size, pe.Reference_Encoder.Conv.Channels, self.strides)):
            self.layer_Dict[f"Conv_{i}"] = _ConvBlock(cin, c, k, s)
            cin, height = c, math.ceil(height / s)
        self.layer_Dict["GRU"] = torch.nn.GRU(cin * height, pe.Reference_Encoder.GRU.Size, pe.Reference_Encoder.GRU.Stacks, batch_first=True)
        self.layer_Dict["Attention"] = _Attention(pe.Reference_Encoder.GRU.Size, pe.Style_Token.Size, pe.Size, pe.Size, pe.Style_Token.Attention_Head)
        self.gst_Tokens = torch.nn.Parameter(torch.randn(pe.Style_Token.Size, pe.Style_Token.Num_Tokens) * 0.5)
        self.n_conv = len(self.strides)

    hip_precision = 0                                         # ops.F32 / ops.BF16: arithmetic of the HIP conv stack (set by GlowTTS from HIP_Precision)
    # The six stride-2 Conv2d layers as direct HIP kernels (conv_stack_hip -> csrc/conv2d_ops.hip; round 6).  Until round 5 this switch selected a
    # patch-matrix + GEMM path that was no faster than MIOpen's direct kernels and stayed off; yaml shapes the kernels do not take (other kernel
    # sizes / strides / channel counts) still run torch's Conv2d.
    use_hip_convs = True
    # The style-token tail behind the GRU (gather of the last valid state, four-head attention over tanh(gst_Tokens), projections) as five HIP launches
    # forward + backward (_GSTTail -> csrc/gst_ops.hip) instead of ~80 PyTorch ones
    use_hip_gst = True

    def forward(self, x, lengths):
        convs = [self.layer_Dict[f"Conv_{i}"].Conv for i in range(self.n_conv)]
        hip_convs = self.use_hip_convs and conv_stack_supported(convs, x)
        if hip_convs:
            if not hasattr(self, "_c2d_cache"):
                self._c2d_cache = {}
            xt = conv_stack_hip(convs, x, self.hip_precision, self._c2d_cache)             # [B, T', C * Mel']
        else:                                                 # other kernel sizes / strides of the yaml: torch's Conv2d (on the same device)
            x = x.unsqueeze(1)
            for i in range(self.n_conv):
                x = self.layer_Dict[f"Conv_{i}"](x)
            xt = x.reshape(x.size(0), x.size(1) * x.size(2), x.size(3)).transpose(2, 1)
        gru = self.layer_Dict["GRU"]
        x = xt.transpose(2, 1)
        if x.is_cuda and gru.num_layers == 1 and 3 * gru.hidden_size <= 1024 and x.dtype == torch.float32:
            x = _GRUFunction.apply(xt.contiguous(), gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0)
        elif not self.training and torch.is_grad_enabled() and x.requires_grad:
            # MIOpen's fused RNN has no backward in eval mode ("miopen RNN backward can only be called in training mode"): gradients
            # through an eval()-mode model take torch's native GRU cell instead (same arithmetic, Modules.py:371)
            with torch.backends.cudnn.flags(enabled=False):
                x = self.layer_Dict["GRU"](x.transpose(2, 1))[0]
        else:
            x = self.layer_Dict["GRU"](x.transpose(2, 1))[0]                              # [B, T', G]
        att = self.layer_Dict["Attention"]
        q_, k_, v_, p_ = (att.layer_Dict[n] for n in ("Query", "Key", "Value", "Projection"))
        if (self.use_hip_gst and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and
                _gst().glowtts_gst_supported(x.size(0), x.size(1), x.size(2), q_.weight.shape[0], att.heads, self.gst_Tokens.shape[1], self.gst_Tokens.shape[0])):
            sq = lambda w: w.squeeze(-1)                       # Conv1d(k = 1) weights [O, I, 1]
            return _GSTTail.apply(x, lengths, int(math.prod(self.strides)), att.heads, self.gst_Tokens, sq(q_.weight), q_.bias, sq(k_.weight), k_.bias,
                                  sq(v_.weight), v_.bias, sq(p_.weight), p_.bias)
        idx = (torch.ceil(lengths / float(math.prod(self.strides))).long() - 1).clamp_min(0)   # Modules.py:373
        x = x[torch.arange(x.size(0), device=x.device), idx]                               # [B, G]
        keys = torch.tanh(self.gst_Tokens).unsqueeze(0).expand(x.size(0), -1, -1)
        return self.layer_Dict["Attention"](x.unsqueeze(2), keys).squeeze(2)


class _GRLFunc(torch.autograd.Function):
    """Gradient_Reversal_Layer.py:6-20: identity forward, -weight * grad backward."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.weight = weight
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return -ctx.weight * g, None


class GRL(torch.nn.Module):
    def __init__(self, weight=1.0):
        super().__init__()
        self.weight = weight

    def forward(self, x):
        return _GRLFunc.apply(x, self.weight)


class Speaker_Classifier_GR(torch.nn.Module):
    """Modules.py:407-435: GRL -> [Conv1x1 + ReLU] per entry of Speaker_Classifier_GR.Channels -> Conv1x1 to Num_Speakers.  Keys
    `layer.Hidden_{i}.{weight,bias}`, `layer.Output_{last i}.{weight,bias}` (the reference names the output after the last hidden index)."""

    def __init__(self, hp):
        super().__init__()
        self.layer = torch.nn.Sequential()
        self.layer.add_module("GRL", GRL(float(hp.Train.Adversarial_Speaker_Weight)))
        cin, index = hp.Prosody_Encoder.Size, 0
        for index, ch in enumerate(hp.Speaker_Classifier_GR.Channels):
            conv = torch.nn.Conv1d(cin, ch, 1)
            torch.nn.init.kaiming_uniform_(conv.weight, nonlinearity="relu")            # Modules.py:983-1003, w_init_gain 'relu'
            torch.nn.init.zeros_(conv.bias)
            self.layer.add_module(f"Hidden_{index}", conv)
            self.layer.add_module(f"ReLU_{index}", torch.nn.ReLU())
            cin = ch
        out = torch.nn.Conv1d(cin, hp.Speaker_Embedding.Num_Speakers, 1)
        torch.nn.init.xavier_uniform_(out.weight, gain=torch.nn.init.calculate_gain("linear"))
        torch.nn.init.zeros_(out.bias)
        self.layer.add_module(f"Output_{index}", out)

    def forward(self, x):
        x = x.unsqueeze(2)
        for m in self.layer:
            x = _conv1x1(m, x) if isinstance(m, torch.nn.Conv1d) else m(x)
        return x.squeeze(2)


class Pitch_Interpolater(torch.nn.Module):
    """Modules.py:387-405: per utterance, the first base_length pitch values linearly interpolated (align_corners) to new_length, zero-padded
    to `max_length` (default: the longest, read back from the device like the reference's torch.max)."""

    def forward(self, pitches, base_lengths, new_lengths, max_length=None):
        """One gather for the whole batch - the reference loops over the utterances and calls `interpolate` on each (`.tolist()`: two device syncs and B
        small launches per call).  torch's linear / align_corners arithmetic restated in fp32: source position j (in - 1) / (out - 1), the two
        neighbours weighted (1 - w, w); the result equals the per-utterance `interpolate` to rounding (<= 1e-7, tests/test_conditioning_encoders.py)."""
        T = int(max_length) if max_length is not None else int(torch.max(new_lengths))
        dev = pitches.device
        bl = base_lengths.to(dev).view(-1, 1)
        nl = new_lengths.to(dev).view(-1, 1)
        j = torch.arange(T, device=dev).view(1, -1)
        scale = torch.where(nl > 1, (bl - 1).to(torch.float32) / (nl - 1).clamp_min(1).to(torch.float32), torch.zeros_like(nl, dtype=torch.float32))
        src = scale * j.to(torch.float32)
        i0 = src.floor().long().clamp_(min=0)
        i0 = torch.minimum(i0, (bl - 1).clamp_min(0))
        i1 = torch.minimum(i0 + 1, (bl - 1).clamp_min(0))
        w = src - i0.to(torch.float32)
        p = pitches.to(torch.float32)
        out = p.gather(1, i0.clamp_max(p.shape[1] - 1)) * (1.0 - w) + p.gather(1, i1.clamp_max(p.shape[1] - 1)) * w
        return torch.where(j < nl, out, torch.zeros_like(out)).to(pitches.dtype)
