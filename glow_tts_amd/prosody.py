"""Conditioning encoders of the PE / GR modes - NOT part of the hot path (each produces one small
per-utterance tensor, < 1 % of the FLOPs): the GST `Prosody_Encoder` (Modules.py:312-385), the adversarial `Speaker_Classifier_GR` behind the
gradient-reversal layer (Modules.py:407-435, Gradient_Reversal_Layer.py:6-35) and the `Pitch_Interpolater` (Modules.py:387-405).  Plain
PyTorch-ROCm modules with the reference's parameter names, so PE- and GR-mode checkpoints load strictly; they feed the HIP decoder / encoder
as conditioning vectors.  Parity: tests/test_gpu_modes.py against the golden vectors of the reference (tiny_pe.npz, tiny_gr.npz)."""
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


class _Im2ColS2(torch.autograd.Function):
    """Patch matrix of Conv2d(3x3, stride 2, padding 1) over channels-last activations [B, H, W, C] -> [B*Ho*Wo, ldc] (glowtts_im2col3x3s2);
    the backward is the gather-sum adjoint (glowtts_col2im3x3s2)."""

    @staticmethod
    def forward(ctx, x, ldc):
        import ctypes
        from . import _lib
        L = _lib.lib()
        L.glowtts_im2col3x3s2.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
        x = x.contiguous()
        B, H, W, C = x.shape
        col = torch.empty(B * ((H + 1) // 2) * ((W + 1) // 2), ldc, device=x.device)
        _lib.check(L.glowtts_im2col3x3s2(_lib.ptr(x), _lib.ptr(col), B, H, W, C, ldc, _lib.stream()), "glowtts_im2col3x3s2")
        ctx.cfg = (B, H, W, C, ldc)
        return col

    @staticmethod
    def backward(ctx, dcol):
        import ctypes
        from . import _lib
        L = _lib.lib()
        L.glowtts_col2im3x3s2.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
        B, H, W, C, ldc = ctx.cfg
        dcol = dcol.contiguous()
        dx = torch.empty(B, H, W, C, device=dcol.device)
        _lib.check(L.glowtts_col2im3x3s2(_lib.ptr(dcol), _lib.ptr(dx), B, H, W, C, ldc, _lib.stream()), "glowtts_col2im3x3s2")
        return dx, None


def conv_stack_hip(convs, mels, precision):
    """The reference encoder's Conv2d(3x3, stride 2, padding 1, no bias) + ReLU stack (Modules.py:320-333, 366-368) on the HIP path: per layer a
    patch-matrix gather and ONE MFMA GEMM with the ReLU in its epilogue (conv_fn.conv_rows -> glowtts_conv_cl; its backward: gate, data-gradient
    GEMM, weight-gradient kernel), activations channels-last.  mels [B, Mel, T] -> [B, T', C * Mel'] (the GRU's input, feature = c * Mel' + h as
    in the reference's reshape at :369)."""
    from .conv_fn import conv_rows
    x = mels.unsqueeze(-1)                                    # [B, H = Mel, W = T, C = 1]
    for conv in convs:
        w = conv.weight                                       # [Co, Ci, 3, 3]
        B, H, W, C = x.shape
        Co, K = w.shape[0], 9 * C
        ldc = max(32, -(-K // 32) * 32)
        col = _Im2ColS2.apply(x, ldc)
        w2 = torch.nn.functional.pad(w.permute(0, 2, 3, 1).reshape(Co, K), (0, ldc - K)).unsqueeze(-1)      # [Co, (kh, kw, ci) + zero pad, 1]
        x = conv_rows(col, w2, None, None, relu=True, precision=precision).view(B, (H + 1) // 2, (W + 1) // 2, Co)
    B, H, W, C = x.shape
    return x.permute(0, 2, 3, 1).reshape(B, W, C * H)


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
    # The six stride-2 Conv2d layers through the library's own GEMM path (conv_stack_hip) instead of torch's Conv2d (MIOpen).  Off by
    # default: measured on the MI355X at B = 32 (bench.py --config 5) the patch-matrix form is no faster than MIOpen's direct kernels
    # (7.97 vs 7.82 ms forward + backward per step; 1.15 ms of kernels, a third of it writing and re-reading the patch matrices), see DESIGN.md.
    use_hip_convs = False

    def forward(self, x, lengths):
        convs = [self.layer_Dict[f"Conv_{i}"].Conv for i in range(self.n_conv)]
        hip_convs = self.use_hip_convs and x.is_cuda and x.dtype == torch.float32 and all(
            c.kernel_size == (3, 3) and c.stride == (2, 2) and c.padding == (1, 1) and c.bias is None and c.out_channels % 4 == 0 for c in convs)
        if hip_convs:
            xt = conv_stack_hip(convs, x, self.hip_precision)                              # [B, T', C * Mel']
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
