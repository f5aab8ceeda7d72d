"""Text encoder (Modules.py:232-284, 438-648; RPR_MHA.py:5-165) - functional form over the reference-named
parameters.  INTERIM (round 1): this file still issues PyTorch-ROCm device ops (rocBLAS / MIOpen GEMMs and
elementwise kernels) for the encoder, which is ~10 % of the FLOPs of a step; the hand-written HIP encoder
kernels replace it next (DESIGN.md "status").  It runs on the GPU only - never on the CPU."""
import math

import torch
import torch.nn.functional as F


def _conv(P, p, x, pad=0):
    return F.conv1d(x, P[p + ".weight"], P.get(p + ".bias"), padding=pad)


def _ln(P, p, x):
    """LayerNorm over channels, eps 1e-4 (Modules.py:472-475, 523-526, 541-544)."""
    return F.layer_norm(x.transpose(1, 2), (x.shape[1],), P[p + ".weight"], P[p + ".bias"], 1e-4).transpose(1, 2)


def _band_index(T, w, device):
    i = torch.arange(T, device=device)
    d = i[None, :] - i[:, None]
    return (d.clamp(-w, w) + w), (d.abs() <= w)


def rpr_attention(P, p, x, mask, heads, window, drop, training):
    """RPR_MHA.py:69-128 in banded form (only relative offsets |j - i| <= window contribute)."""
    B, C, T = x.shape
    D = C // heads
    q = _conv(P, p + ".layer_Dict.Query", x).view(B, heads, D, T).transpose(2, 3)
    k = _conv(P, p + ".layer_Dict.Key", x).view(B, heads, D, T).transpose(2, 3)
    v = _conv(P, p + ".layer_Dict.Value", x).view(B, heads, D, T).transpose(2, 3)
    relk, relv = P[p + ".weight_K"][0], P[p + ".weight_V"][0]               # [2w+1, D]
    gidx, band = _band_index(T, window, x.device)
    qr = q @ relk.t()                                                        # [B,H,T,2w+1]
    rel = torch.gather(qr, 3, gidx.view(1, 1, T, T).expand(B, heads, T, T)) * band
    scores = (q @ k.transpose(2, 3) + rel) / math.sqrt(D)                    # RPR_MHA.py:103,109
    amask = (mask.transpose(1, 2) * mask).unsqueeze(1)                       # Modules.py:558
    scores = scores.masked_fill(amask == 0, -1e4)                            # RPR_MHA.py:117
    pr = F.dropout(torch.softmax(scores, dim=-1), drop, training)            # RPR_MHA.py:119-120
    out = pr @ v
    # relative-V: sum_d P[i, i+d] relV[d+w]                                  RPR_MHA.py:124-126
    pb = torch.zeros(B, heads, T, 2 * window + 1, device=x.device, dtype=x.dtype)
    pb.scatter_add_(3, gidx.view(1, 1, T, T).expand(B, heads, T, T), pr * band)
    out = out + pb @ relv
    return _conv(P, p + ".layer_Dict.Projection", out.transpose(2, 3).reshape(B, C, T))


def encoder_forward(P, hp, tokens, mask, speakers=None, prosodies=None, training=False, prefix="layer_Dict.Encoder"):
    """Modules.py:262-284 -> mean [B,mel,T], log_std [B,mel,T], log_durations [B,1,T]."""
    e = hp.Encoder
    C = e.Channels
    x = F.embedding(tokens, P[prefix + ".layer_Dict.Embedding.weight"]).transpose(1, 2) * math.sqrt(C)   # :267
    # Prenet :438-489
    res = x
    kp = e.Prenet.Kernel_Size
    for i in range(e.Prenet.Stacks):
        q = f"{prefix}.layer_Dict.Prenet.layer_Dict.CLRD_{i}.layer_Dict"
        x = _conv(P, q + ".Conv", x * mask, (kp - 1) // 2)
        x = F.dropout(torch.relu(_ln(P, q + ".LayerNorm", x)), e.Prenet.Dropout_Rate, training)
    x = (_conv(P, prefix + ".layer_Dict.Prenet.layer_Dict.Conv1x1", x) + res) * mask
    # Transformer :492-573
    kf = e.Transformer.Conv.Kernel_Size
    dr = e.Transformer.Dropout_Rate
    for i in range(e.Transformer.Stacks):
        q = f"{prefix}.layer_Dict.Transformer.layer_Dict.ANCRDCN_{i}.layer_Dict"
        x = x * mask
        res = x
        a = rpr_attention(P, q + ".Attention", x, mask, e.Transformer.Attention.Heads, e.Transformer.Attention.Window_Size, dr, training)
        x = _ln(P, q + ".LayerNorm_0", F.dropout(a, dr, training) + res)
        res = x
        h = F.dropout(torch.relu(_conv(P, q + ".Conv_0", x * mask, (kf - 1) // 2)), dr, training)
        h = F.dropout(_conv(P, q + ".Conv_1", h * mask, (kf - 1) // 2), dr, training)
        x = _ln(P, q + ".LayerNorm_1", h * mask + res)
    x = x * mask
    proj = _conv(P, prefix + ".layer_Dict.Project", x) * mask
    M = hp.Sound.Mel_Dim
    mean, log_std = proj[:, :M], proj[:, M:]
    # Duration predictor on detached features (:277-282, 602-618)
    d = x.detach()
    cond = None
    if speakers is not None:
        cond = speakers.detach()
    if prosodies is not None:
        cond = prosodies.detach() if cond is None else cond + prosodies.detach()
    if cond is not None:
        d = torch.cat([d, cond.unsqueeze(2).expand(-1, -1, d.shape[2])], dim=1)
    dp = e.Duration_Predictor
    for i in range(dp.Stacks):
        q = f"{prefix}.layer_Dict.Duration_Predictor.layer_Dict.CRND_{i}.layer_Dict.Conv"
        d = F.dropout(torch.relu(_conv(P, q, d * mask, (dp.Kernel_Size - 1) // 2)), dp.Dropout_Rate, training)
    log_dur = _conv(P, prefix + ".layer_Dict.Duration_Predictor.layer_Dict.Projection", d * mask) * mask
    return mean, log_std, log_dur
