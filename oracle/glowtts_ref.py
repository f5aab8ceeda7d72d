"""ORACLE - test infrastructure, not product code.

A CPU fp32 restatement, in plain functional PyTorch, of the Glow-TTS hot path of the
reference (CODEJIN/Glow_TTS).  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this file; the product path (glow_tts_amd/) never does.

Parity pin: validated in the build container against the *imported reference itself*
(tests/golden/make_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py replays the
fixtures on every run).  Every function cites the reference file:line it follows.

The functions take the reference's state dict (`sd`: name -> tensor, same keys as
`GlowTTS().state_dict()`), so reference checkpoints feed it unchanged.  Everything is
differentiable torch code, so gradients for parity tests come from torch.autograd.
"""
import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

LOG_2PI = math.log(2.0 * math.pi)


@dataclass
class Cfg:
    """The subset of Hyper_Parameters.yaml the hot path reads (Hyper_Parameters.yaml:1-70)."""
    mode: str = "Vanilla"            # Vanilla | SE | PE | GR      (:17-18)
    mel_dim: int = 80                # Sound.Mel_Dim                (:3)
    max_abs_mel: float = 4.0         # Sound.Max_Abs_Mel            (:9)
    enc_channels: int = 192          # Encoder.Channels             (:21)
    n_tokens: int = 35               # Encoder.Embedding_Tokens     (:22)
    prenet_kernel: int = 5           # Encoder.Prenet.Kernel_Size   (:24)
    prenet_stacks: int = 3           # Encoder.Prenet.Stacks        (:26)
    heads: int = 2                   # Transformer.Attention.Heads  (:29)
    window: int = 4                  # Attention.Window_Size        (:30)
    ffn_kernel: int = 3              # Transformer.Conv.Kernel_Size (:32)
    ffn_channels: int = 768          # Transformer.Conv.Calc_Channels (:33)
    enc_stacks: int = 6              # Transformer.Stacks           (:35)
    dp_kernel: int = 3               # Duration_Predictor.Kernel_Size (:37)
    dp_channels: int = 256           # Duration_Predictor.Channels  (:38)
    dp_stacks: int = 2               # Duration_Predictor.Stacks    (:39)
    n_flows: int = 12                # Decoder.Stack                (:43)
    n_squeeze: int = 2               # Decoder.Num_Squeeze          (:44)
    n_split: int = 4                 # Decoder.Num_Split            (:45)
    wn_channels: int = 192           # Affine_Coupling.Calc_Channels (:47)
    wn_layers: int = 4               # WaveNet.Num_Layers           (:49)
    wn_kernel: int = 5               # WaveNet.Kernel_Size          (:50)
    spk_type: str = "LUT"            # Speaker_Embedding.Type       (:55)
    n_speakers: int = 109            # Speaker_Embedding.Num_Speakers (:56)
    spk_dim: int = 256               # Speaker_Embedding.Embedding_Size (:57)
    pro_dim: int = 256               # Prosody_Encoder.Size         (:70)
    pe_strides: tuple = (2, 2, 2, 2, 2, 2)      # Prosody_Encoder.Reference_Encoder.Conv.Strides (:75)
    pe_kernels: tuple = (3, 3, 3, 3, 3, 3)      # ...Conv.Kernel_Size        (:73)
    pe_heads: int = 4                # Prosody_Encoder.Style_Token.Attention_Head (:82)
    grl_weight: float = 0.0005       # Train.Adversarial_Speaker_Weight (:112)
    gr_hidden: int = 1               # len(Speaker_Classifier_GR.Channels) (:84-85)

    @staticmethod
    def from_yaml_dict(d):
        e, dec = d["Encoder"], d["Decoder"]
        return Cfg(
            mode=d["Mode"], mel_dim=d["Sound"]["Mel_Dim"], max_abs_mel=float(d["Sound"]["Max_Abs_Mel"]),
            enc_channels=e["Channels"], n_tokens=e["Embedding_Tokens"],
            prenet_kernel=e["Prenet"]["Kernel_Size"], prenet_stacks=e["Prenet"]["Stacks"],
            heads=e["Transformer"]["Attention"]["Heads"], window=e["Transformer"]["Attention"]["Window_Size"],
            ffn_kernel=e["Transformer"]["Conv"]["Kernel_Size"], ffn_channels=e["Transformer"]["Conv"]["Calc_Channels"],
            enc_stacks=e["Transformer"]["Stacks"], dp_kernel=e["Duration_Predictor"]["Kernel_Size"],
            dp_channels=e["Duration_Predictor"]["Channels"], dp_stacks=e["Duration_Predictor"]["Stacks"],
            n_flows=dec["Stack"], n_squeeze=dec["Num_Squeeze"], n_split=dec["Num_Split"],
            wn_channels=dec["Affine_Coupling"]["Calc_Channels"],
            wn_layers=dec["Affine_Coupling"]["WaveNet"]["Num_Layers"],
            wn_kernel=dec["Affine_Coupling"]["WaveNet"]["Kernel_Size"],
            spk_type=d["Speaker_Embedding"]["Type"], n_speakers=d["Speaker_Embedding"]["Num_Speakers"],
            spk_dim=d["Speaker_Embedding"]["Embedding_Size"], pro_dim=d["Prosody_Encoder"]["Size"],
            pe_strides=tuple(d["Prosody_Encoder"]["Reference_Encoder"]["Conv"]["Strides"]),
            pe_kernels=tuple(d["Prosody_Encoder"]["Reference_Encoder"]["Conv"]["Kernel_Size"]),
            pe_heads=d["Prosody_Encoder"]["Style_Token"]["Attention_Head"],
            grl_weight=float(d["Train"]["Adversarial_Speaker_Weight"]), gr_hidden=len(d["Speaker_Classifier_GR"]["Channels"]),
        )


# --------------------------------------------------------------------------- helpers
def mask_from_lengths(lengths, max_len=None):
    """Modules.py:206-211 Mask_Generate: float [B,1,T], 1 where t < length."""
    T = int(max_len) if max_len is not None else int(lengths.max())
    return (torch.arange(T)[None, :] < lengths[:, None]).unsqueeze(1).to(torch.float32)


def eff_weight(sd, p):
    """Effective conv weight.  Old-style torch weight_norm (Modules.py:766,818,825,...):
    w = g * v / ||v||_2, norm over dims (1,2) per output channel; plain `weight` otherwise."""
    if p + ".weight_g" in sd:
        v = sd[p + ".weight_v"]
        return sd[p + ".weight_g"] * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)
    return sd[p + ".weight"]


def conv(sd, p, x, pad=0):
    b = sd.get(p + ".bias")
    return F.conv1d(x, eff_weight(sd, p), b, padding=pad)


def ln_ch(sd, p, x):
    """LayerNorm over the channel axis, eps 1e-4 (Modules.py:472-475,485,523-526,562,571)."""
    return F.layer_norm(x.transpose(1, 2), (x.shape[1],), sd[p + ".weight"], sd[p + ".bias"], 1e-4).transpose(1, 2)


# --------------------------------------------------------------------------- encoder
def rpr_attention(sd, p, x, mask, cfg):
    """RPR_MHA.py:69-128 self-attention with Shaw relative positions, restated in banded form:
    only offsets d = j - i in [-w, w] contribute and row d + w of weight_K / weight_V is the
    embedding of offset d (Get_Relative_Embedding :131-140 zero-pads the rest).
    x [B,C,T]; mask [B,1,T] float.  Returns [B,C,T]."""
    B, C, T = x.shape
    H, w = cfg.heads, cfg.window
    D = C // H
    q = conv(sd, p + ".layer_Dict.Query", x).view(B, H, D, T).transpose(2, 3)   # [B,H,T,D]  (:82,99)
    k = conv(sd, p + ".layer_Dict.Key", x).view(B, H, D, T).transpose(2, 3)     # (:83,100)
    v = conv(sd, p + ".layer_Dict.Value", x).view(B, H, D, T).transpose(2, 3)   # (:84,101)
    scores = q @ k.transpose(2, 3)                                               # (:103)
    relk = sd[p + ".weight_K"][0]                                                # [2w+1, D] shared over heads (:59-64)
    relv = sd[p + ".weight_V"][0]
    qr = q @ relk.t()                                                            # [B,H,T,2w+1]: q_i . relK[d+w]
    idx = torch.arange(T)
    dmat = idx[None, :] - idx[:, None]                                           # d = j - i
    band = (dmat.abs() <= w)
    gather = (dmat.clamp(-w, w) + w)                                             # [T,T] -> column of qr
    rel_scores = torch.gather(qr, 3, gather.view(1, 1, T, T).expand(B, H, T, T)) * band
    scores = (scores + rel_scores) / math.sqrt(D)                                # (:103,109)
    amask = (mask.transpose(1, 2) * mask).unsqueeze(1)                           # [B,1,T,T]  Modules.py:558
    scores = scores.masked_fill(amask == 0, -1e4)                                # (:117)
    pr = torch.softmax(scores, dim=-1)                                           # (:119) dropout is identity in eval
    out = pr @ v                                                                 # (:121)
    # relative-V term: sum_d P[i, i+d] * relV[d+w]                               # (:124-126)
    pb = torch.zeros(B, H, T, 2 * w + 1, dtype=x.dtype)
    for d in range(-w, w + 1):
        lo, hi = max(0, -d), min(T, T - d)
        if hi > lo:
            ii = torch.arange(lo, hi)
            pb[:, :, ii, d + w] = pr[:, :, ii, ii + d]
    out = out + pb @ relv
    out = out.transpose(2, 3).reshape(B, C, T)                                   # (:128)
    return conv(sd, p + ".layer_Dict.Projection", out)                           # (:93)


def prenet(sd, p, x, mask, cfg):
    """Modules.py:438-489: stacks x [Conv k5 on x*mask -> LN -> ReLU -> Dropout] -> 1x1 + residual -> *mask."""
    res = x
    for i in range(cfg.prenet_stacks):
        q = f"{p}.layer_Dict.CLRD_{i}.layer_Dict"
        x = conv(sd, q + ".Conv", x * mask, pad=(cfg.prenet_kernel - 1) // 2)
        x = torch.relu(ln_ch(sd, q + ".LayerNorm", x))
    return (conv(sd, p + ".layer_Dict.Conv1x1", x) + res) * mask


def transformer(sd, p, x, mask, cfg):
    """Modules.py:492-573 (ANCRDCN x stacks, final *mask)."""
    pad = (cfg.ffn_kernel - 1) // 2
    for i in range(cfg.enc_stacks):
        q = f"{p}.layer_Dict.ANCRDCN_{i}.layer_Dict"
        x = x * mask                                                             # :554 (in place in the reference)
        res = x
        a = rpr_attention(sd, q + ".Attention", x, mask, cfg)                    # :556-559
        x = ln_ch(sd, q + ".LayerNorm_0", a + res)                               # :562
        res = x
        h = torch.relu(conv(sd, q + ".Conv_0", x * mask, pad))                   # :565-566
        h = conv(sd, q + ".Conv_1", h * mask, pad)                               # :568
        x = ln_ch(sd, q + ".LayerNorm_1", h * mask + res)                        # :571
    return x * mask                                                              # :507


def duration_predictor(sd, p, x, mask, cond, cfg):
    """Modules.py:576-648.  cond = speakers + prosodies ([B,256]) or None."""
    if cond is not None:
        x = torch.cat([x, cond.unsqueeze(2).expand(-1, -1, x.shape[2])], dim=1)  # :606-612
    for i in range(cfg.dp_stacks):
        x = torch.relu(conv(sd, f"{p}.layer_Dict.CRND_{i}.layer_Dict.Conv", x * mask, (cfg.dp_kernel - 1) // 2))
    return conv(sd, p + ".layer_Dict.Projection", x * mask) * mask               # :616-618


def encoder(sd, tokens, mask, cfg, speakers=None, prosodies=None, p="layer_Dict.Encoder"):
    """Modules.py:262-284 -> mean [B,mel,T], log_std [B,mel,T], log_dur [B,1,T]."""
    x = F.embedding(tokens, sd[p + ".layer_Dict.Embedding.weight"]).transpose(1, 2) * math.sqrt(cfg.enc_channels)
    x = prenet(sd, p + ".layer_Dict.Prenet", x, mask, cfg)
    x = transformer(sd, p + ".layer_Dict.Transformer", x, mask, cfg)
    proj = conv(sd, p + ".layer_Dict.Project", x) * mask
    mean, log_std = proj[:, :cfg.mel_dim], proj[:, cfg.mel_dim:]
    cond = None
    if speakers is not None or prosodies is not None:
        cond = 0
        if speakers is not None:
            cond = cond + speakers.detach()                                     # :277-280
        if prosodies is not None:
            cond = cond + prosodies.detach()
    log_dur = duration_predictor(sd, p + ".layer_Dict.Duration_Predictor", x.detach(), mask, cond, cfg)
    return mean, log_std, log_dur


# --------------------------------------------------------------------------- conditioning encoders (PE / GR modes)
def prosody_encoder(sd, mels, lengths, cfg, p="layer_Dict.Prosody_Encoder"):
    """Modules.py:312-385 (GST): 6 x [Conv2d(stride 2, no bias) + ReLU] over (mel, time) -> GRU over the compressed time axis -> the
    state at the last valid compressed step of each utterance (:373-374) -> multi-head attention (RPR_MHA.py:69-128 without relative
    positions or masks) over tanh(gst_Tokens) -> [B, Prosody_Encoder.Size]."""
    x = mels.unsqueeze(1)                                                        # [B, 1, Mel_d, T]
    for i, (k, st) in enumerate(zip(cfg.pe_kernels, cfg.pe_strides)):
        x = torch.relu(F.conv2d(x, sd[f"{p}.layer_Dict.Conv_{i}.Conv.weight"], None, stride=st, padding=(k - 1) // 2))
    x = x.reshape(x.size(0), x.size(1) * x.size(2), x.size(3)).transpose(2, 1)   # [B, T', C*H']
    g = f"{p}.layer_Dict.GRU"
    w_ih, w_hh, b_ih, b_hh = sd[g + ".weight_ih_l0"], sd[g + ".weight_hh_l0"], sd[g + ".bias_ih_l0"], sd[g + ".bias_hh_l0"]
    Hs = w_hh.shape[1]
    h = x.new_zeros(x.size(0), Hs)
    hs = []
    for t in range(x.size(1)):                                                   # torch.nn.GRU, one layer, batch_first (gate order r, z, n)
        gi, gh = x[:, t] @ w_ih.t() + b_ih, h @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, :Hs] + gh[:, :Hs])
        zg = torch.sigmoid(gi[:, Hs:2 * Hs] + gh[:, Hs:2 * Hs])
        n = torch.tanh(gi[:, 2 * Hs:] + r * gh[:, 2 * Hs:])
        h = (1 - zg) * n + zg * h
        hs.append(h)
    hs = torch.stack(hs, 1)                                                      # [B, T', Hs]
    idx = torch.ceil(lengths / float(np.prod(cfg.pe_strides, dtype=float))).to(lengths.dtype) - 1     # :373
    q = hs[torch.arange(hs.size(0)), idx].unsqueeze(2)                           # [B, Hs, 1]
    keys = torch.tanh(sd[p + ".gst_Tokens"]).unsqueeze(0).expand(q.size(0), -1, -1)
    a = p + ".layer_Dict.Attention.layer_Dict"
    Q, K, V = conv(sd, a + ".Query", q), conv(sd, a + ".Key", keys), conv(sd, a + ".Value", keys)
    B, Cc, _ = Q.shape
    Hh = cfg.pe_heads
    D = Cc // Hh
    Q, K, V = (t.view(B, Hh, D, -1).transpose(2, 3) for t in (Q, K, V))
    o = torch.softmax(Q @ K.transpose(3, 2) / math.sqrt(D), dim=-1) @ V          # RPR_MHA.py:103,118,121
    o = o.transpose(3, 2).contiguous().view(B, Cc, 1)
    return conv(sd, a + ".Projection", o).squeeze(2)


class _GRL(torch.autograd.Function):
    """Gradient_Reversal_Layer.py:6-20: identity forward, -weight * grad backward."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.weight = weight
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return -ctx.weight * g, None


def speaker_classifier_gr(sd, prosodies, cfg, p="layer_Dict.Speaker_Classifier_GR.layer"):
    """Modules.py:407-435: GRL -> [Conv1x1 + ReLU] x len(Channels) -> Conv1x1 to Num_Speakers, on [B, Size, 1]."""
    x = _GRL.apply(prosodies, cfg.grl_weight).unsqueeze(2)
    for i in range(cfg.gr_hidden):
        x = torch.relu(conv(sd, f"{p}.Hidden_{i}", x))
    return conv(sd, f"{p}.Output_{cfg.gr_hidden - 1}", x).squeeze(2)             # (the reference names it after the LAST hidden index, :426)


def pitch_interpolate(pitches, base_lengths, new_lengths):
    """Modules.py:387-405: per-utterance linear interpolation (align_corners) of the first base_length values to new_length,
    zero-padded to the longest."""
    out = [F.interpolate(pt[:int(bl)].view(1, 1, -1), size=int(nl), mode="linear", align_corners=True).view(-1)
           for pt, bl, nl in zip(pitches, base_lengths, new_lengths)]
    T = int(max(int(n) for n in new_lengths))
    return torch.stack([F.pad(o, [0, T - o.numel()]) for o in out])


# --------------------------------------------------------------------------- flow decoder
def squeeze(x, mask, n=2):
    """Modules.py:895-907: out[b, s*C + c, t] = in[b, c, n*t + s]; mask' = mask[:, :, n-1::n]."""
    B, C, T = x.shape
    T = (T // n) * n
    x = x[:, :, :T].view(B, C, T // n, n).permute(0, 3, 1, 2).reshape(B, C * n, T // n)
    m = mask[:, :, n - 1::n]
    return x * m, m


def unsqueeze(x, mask, n=2):
    """Modules.py:914-924 (inverse permutation, mask repeated n times)."""
    B, C, T = x.shape
    x = x.view(B, n, C // n, T).permute(0, 2, 3, 1).reshape(B, C // n, T * n)
    m = mask.unsqueeze(-1).repeat(1, 1, 1, n).view(B, 1, T * n)
    return x * m, m


def actnorm_init(x, mask):
    """Modules.py:698-711 data-dependent init -> (logs, bias) as [1,C,1]."""
    den = mask.sum((0, 2))
    mean = (x * mask).sum((0, 2)) / den
    sq = (x * x * mask).sum((0, 2)) / den
    logs = 0.5 * torch.log(torch.clamp_min(sq - mean ** 2, 1e-7))
    return (-logs).view(1, -1, 1), (-mean * torch.exp(-logs)).view(1, -1, 1)


def actnorm(sd, p, x, mask, reverse):
    """Modules.py:689-694."""
    logs, bias = sd[p + ".logs"], sd[p + ".bias"]
    if reverse:
        return (x - bias) * torch.exp(-logs) * mask, None
    return (bias + torch.exp(logs) * x) * mask, logs.sum() * mask.sum((1, 2))


def inv1x1(sd, p, x, mask, cfg, reverse):
    """Modules.py:727-758: 4x4 mix of channel groups {2c, 2c+1, C/2+2c, C/2+2c+1}."""
    B, C, T = x.shape
    s = cfg.n_split
    W = sd[p + ".weight"]
    xg = x.view(B, 2, C // s, s // 2, T).permute(0, 1, 3, 2, 4).reshape(B, s, C // s, T)
    if reverse:
        Wm, logdet = torch.inverse(W), None
    else:
        Wm, logdet = W, torch.logdet(W) * (C / s) * mask.sum((1, 2))
    z = torch.einsum("oi,bigt->bogt", Wm, xg)
    z = z.view(B, 2, s // 2, C // s, T).permute(0, 1, 3, 2, 4).reshape(B, C, T) * mask
    return z, logdet


def wavenet(sd, p, x, mask, cfg, speakers=None, prosodies=None, pitches=None, drop=None):
    """Modules.py:858-887 (k=5 'same' conv, NOT dilated / NOT causal).  drop (test hook; eval mode when None): callable(p, layer, ins) -> the
    In_l output after the WaveNet dropout of Modules.py:861-862 - a parity test injects the keep masks the HIP path drew, since torch's
    generator cannot be matched."""
    out = torch.zeros_like(x)
    H = cfg.wn_channels
    for i in range(cfg.wn_layers):
        ins = conv(sd, f"{p}.layer_Dict.In_{i}", x, (cfg.wn_kernel - 1) // 2)
        if drop is not None:
            ins = drop(p, i, ins)                                                # :862
        if speakers is not None:
            ins = ins + conv(sd, f"{p}.layer_Dict.Speaker_{i}", speakers.unsqueeze(2))
        if prosodies is not None:
            ins = ins + conv(sd, f"{p}.layer_Dict.Prosody_{i}", prosodies.unsqueeze(2))
        if pitches is not None:
            ins = ins + conv(sd, f"{p}.layer_Dict.Pitch_{i}", pitches)
        acts = torch.tanh(ins[:, :H]) * torch.sigmoid(ins[:, H:])                # :885-887
        rs = conv(sd, f"{p}.layer_Dict.Res_Skip_{i}", acts)
        if i < cfg.wn_layers - 1:
            x = (x + rs[:, :H]) * mask                                           # :878
            out = out + rs[:, H:]
        else:
            out = out + rs
    return out * mask


def coupling(sd, p, x, mask, cfg, reverse, speakers=None, prosodies=None, pitches=None, drop=None):
    """Modules.py:780-810."""
    C = x.shape[1]
    xa, xb = x[:, :C // 2], x[:, C // 2:]
    h = conv(sd, p + ".layer_Dict.Start", xa) * mask
    h = wavenet(sd, p + ".layer_Dict.WaveNet", h, mask, cfg, speakers, prosodies, pitches, drop)
    outs = conv(sd, p + ".layer_Dict.End", h)
    m, logs = outs[:, :C // 2], outs[:, C // 2:]
    if reverse:
        xb, logdet = (xb - m) * torch.exp(-logs) * mask, None
    else:
        xb, logdet = (m + torch.exp(logs) * xb) * mask, (logs * mask).sum((1, 2))
    return torch.cat([xa, xb], 1), logdet


def decoder(sd, x, mask, cfg, reverse=False, speakers=None, prosodies=None, pitches=None,
            p="layer_Dict.Decoder", drop=None):
    """Modules.py:298-309.  Returns (z, log_dets [B] or None, mask).  drop: see `wavenet`."""
    x, sm = squeeze(x, mask, cfg.n_squeeze)
    if pitches is not None:
        pitches, _ = squeeze(pitches.unsqueeze(1), mask, cfg.n_squeeze)
    logdets = []
    order = range(cfg.n_flows - 1, -1, -1) if reverse else range(cfg.n_flows)
    for f in order:
        q = f"{p}.layer_Dict.Flows.{f}.layers"
        if reverse:                                                              # Modules.py:664
            x, _ = coupling(sd, q + ".2", x, sm, cfg, True, speakers, prosodies, pitches)
            x, _ = inv1x1(sd, q + ".1", x, sm, cfg, True)
            x, _ = actnorm(sd, q + ".0", x, sm, True)
        else:
            x, l0 = actnorm(sd, q + ".0", x, sm, False)
            x, l1 = inv1x1(sd, q + ".1", x, sm, cfg, False)
            x, l2 = coupling(sd, q + ".2", x, sm, cfg, False, speakers, prosodies, pitches, drop)
            logdets += [l0, l1, l2]
    x, m = unsqueeze(x, sm, cfg.n_squeeze)
    return x, (None if reverse else torch.stack(logdets).sum(0)), m


# --------------------------------------------------------------------------- alignment
def log_prior(mean, log_std, z):
    """Modules.py:108-114: log N(z_y; mean_x, std_x) for every (token x, frame y) -> [B,Tx,Ty]."""
    r = torch.exp(-2 * log_std)
    return (torch.sum(-0.5 * LOG_2PI - log_std, dim=1).unsqueeze(-1)
            + r.transpose(2, 1) @ (-0.5 * (z ** 2))
            + (mean * r).transpose(2, 1) @ z
            + torch.sum(-0.5 * (mean ** 2) * r, dim=1).unsqueeze(-1))


def mas(value, mask, max_neg_val=-1e9):
    """monotonic_align/__init__.py:6-21 around core.pyx:40 maximum_path_c, through the C
    restatement oracle/mas_ref.c.  value, mask [B,Tx,Ty] -> path (same dtype as value)."""
    from . import mas_ref
    v = (value * mask).detach().cpu().numpy().astype(np.float32)
    mk = mask.detach().cpu().numpy()
    t_x = mk.sum(1)[:, 0].astype(np.int32)
    t_y = mk.sum(2)[:, 0].astype(np.int32)
    path = mas_ref.maximum_path_c(v, t_x, t_y, max_neg_val)
    return torch.from_numpy(path).to(dtype=value.dtype)


def forward_train(sd, cfg, tokens, token_lengths, mels, mel_lengths, speakers=None, prosodies=None, pitches=None, attn=None):
    """Modules.py:50-126 GlowTTS.forward, every Mode: Vanilla / SE (LUT ids or pre-computed d-vectors) / PE (GST prosody encoder on the
    target mels, :81-82) / GR (LUT + prosody encoder + adversarial speaker classifier :84-87 + per-frame pitch conditioning :89-90, 300-301).
    `speakers`: int64 ids (LUT) or float [B,256] vectors (GE2E d-vectors are an input); `prosodies`: pre-computed vectors
    override the prosody encoder.  `attn` (test hook, not in the reference): use this alignment instead of searching one - lets a parity test
    compare gradients on EQUAL alignments when the search sits on near-ties."""
    mode = cfg.mode.upper()
    attn_given = attn
    if speakers is not None and speakers.dtype == torch.long:
        speakers = F.embedding(speakers, sd["layer_Dict.LUT.weight"])           # :73-74
    elif speakers is not None:
        speakers = speakers.detach()                                             # :77 (GE2E output is detached)
    if prosodies is None and mode in ("PE", "GR"):
        prosodies = prosody_encoder(sd, mels, mel_lengths, cfg)                  # :81-82
    classified = speaker_classifier_gr(sd, prosodies, cfg) if mode == "GR" else None     # :84-87
    if mode != "GR":
        pitches = None                                                           # :89-90
    tmask = mask_from_lengths(token_lengths, tokens.shape[1])
    mmask = mask_from_lengths(mel_lengths, mels.shape[2])
    mean, log_std, log_dur = encoder(sd, tokens, tmask, cfg, speakers, prosodies)
    z, log_dets, mmask2 = decoder(sd, mels, mmask, cfg, False, speakers, prosodies, pitches)
    amask = (tmask.unsqueeze(-1) * mmask2.unsqueeze(2)).squeeze(1)               # :102-103
    with torch.no_grad():
        logp = log_prior(mean, log_std, z)
        attn = mas(logp, amask) if attn_given is None else attn_given           # :116
    mel_mean = mean @ attn                                                       # :120
    mel_log_std = log_std @ attn                                                 # :121
    log_dur_t = torch.log(attn.unsqueeze(1).sum(-1) + 1e-7) * tmask              # :122
    return dict(z=z, mel_mean=mel_mean, mel_log_std=mel_log_std, log_dets=log_dets, log_dur=log_dur,
                log_dur_target=log_dur_t, attn=attn, mean=mean, log_std=log_std, logp=logp, classified=classified,
                prosodies=prosodies)


def mle_loss(z, mean, log_std, log_dets, mel_lengths, cfg):
    """Modules.py:1020-1029."""
    loss = log_std.sum() + 0.5 * (torch.exp(-2 * log_std) * (z - mean) ** 2).sum() - log_dets.sum()
    loss = loss / ((mel_lengths // cfg.n_squeeze).sum() * cfg.n_squeeze * cfg.mel_dim)
    return loss + 0.5 * LOG_2PI


def train_losses(out, mel_lengths, cfg, speakers=None):
    """Train.py:203-216: MLE + MSE(log_durations, targets) (mean over the padded [B,1,Tt]); GR mode with `speakers` ids: also the
    adversarial speaker cross-entropy (returned as a third value)."""
    mle = mle_loss(out["z"], out["mel_mean"], out["mel_log_std"], out["log_dets"], mel_lengths, cfg)
    length = F.mse_loss(out["log_dur"], out["log_dur_target"])
    if speakers is not None and out.get("classified") is not None:
        return mle, length, F.cross_entropy(out["classified"], speakers)
    return mle, length


def path_from_durations(durations, amask):
    """Modules.py:213-229 Path_Generate: hard alignment from cumulative durations."""
    B, Tx, Ty = amask.shape
    cum = torch.cumsum(durations, dim=1)
    ar = torch.arange(Ty)[None, None, :]
    upto = (ar < cum[:, :, None]).to(amask.dtype)
    prev = F.pad(upto, [0, 0, 1, 0])[:, :-1]
    return (upto - prev) * amask


def inference(sd, cfg, tokens, token_lengths, noise, length_scale, noise_scale=1.0,
              speakers=None, prosodies=None, mels_for_prosody=None, mel_lengths_for_prosody=None, pitches=None, pitch_lengths=None):
    """Modules.py:128-204 GlowTTS.inference with the noise tensor *injected*
    (the reference draws torch.randn_like at :187).  `noise` must be [B, mel, >=max T_mel]."""
    mode = cfg.mode.upper()
    if speakers is not None and speakers.dtype == torch.long:
        speakers = F.embedding(speakers, sd["layer_Dict.LUT.weight"])
    if prosodies is None and mode in ("PE", "GR"):
        prosodies = prosody_encoder(sd, mels_for_prosody, mel_lengths_for_prosody, cfg)      # :159-160
    tmask = mask_from_lengths(token_lengths, tokens.shape[1])
    mean, log_std, log_dur = encoder(sd, tokens, tmask, cfg, speakers, prosodies)
    ls = length_scale.view(-1, 1, 1)
    dur = torch.ceil(torch.exp(log_dur) * tmask * ls).squeeze(1)                 # :173
    mel_lengths = torch.clamp_min(dur.sum(1), 1.0).long()                        # :174
    mmask = mask_from_lengths(mel_lengths)
    amask = (tmask.unsqueeze(-1) * mmask.unsqueeze(2)).squeeze(1)
    attn = path_from_durations(dur, amask)
    mel_mean = mean @ attn
    mel_log_std = log_std @ attn
    Tm = mel_mean.shape[2]
    z = (mel_mean + torch.exp(mel_log_std) * noise[:, :, :Tm] * noise_scale) * mmask   # :187-191
    pit = pitch_interpolate(pitches, pitch_lengths, mel_lengths) if mode == "GR" else None     # :193-196
    mels, _, mmask2 = decoder(sd, z, mmask, cfg, True, speakers, prosodies, pit)
    mels = mels.masked_fill(mmask2 == 0.0, -cfg.max_abs_mel)                     # :202
    return mels, mel_lengths, attn
