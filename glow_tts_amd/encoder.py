"""Text encoder (Modules.py:232-284, 438-648; RPR_MHA.py:5-165) - functional form over the reference-named parameters,
channels-last "rows" layout ([B, T + 4, C] with two zero pad rows around every utterance).

Every convolution (Prenet k=5, Q/K/V/Projection 1x1, FFN k=3, Project, Duration predictor) runs on the hand-written
MFMA kernels (glow_tts_amd/csrc/gemm_cl.hip, wgrad_cl.hip) through conv_fn.ConvRows.
INTERIM (round 1, DESIGN.md "status"): LayerNorm, the attention contractions / softmax / relative-position gathers,
dropout and the embedding lookup are still PyTorch-ROCm device ops on this layout; they are the next kernels to be
written.  Nothing here runs on the CPU.

Layout invariant: every stored activation is zero on padded frames and pad rows.  The reference multiplies by the
mask only in front of convolutions (Modules.py:484,565,568,616,643) and at block ends; masking earlier only changes
rows that the reference discards as well (all other ops are per-frame)."""
import math

import torch
import torch.nn.functional as F

from .conv_fn import conv_rows

ROW_PAD = 2


def to_rows(x_bct):
    """[B,C,T] -> [B, T+4, C] with zero pad rows."""
    return F.pad(x_bct.transpose(1, 2), (0, 0, ROW_PAD, ROW_PAD)).contiguous()


def from_rows(rows_btc):
    """[B, T+4, C] -> [B,C,T]."""
    return rows_btc[:, ROW_PAD:-ROW_PAD].transpose(1, 2)


def _ln(P, p, x):
    """LayerNorm over channels, eps 1e-4 (Modules.py:472-475, 523-526, 541-544): the last dim in this layout."""
    return F.layer_norm(x, (x.shape[-1],), P[p + ".weight"], P[p + ".bias"], 1e-4)


def _band_index(T, w, device):
    i = torch.arange(T, device=device)
    d = i[None, :] - i[:, None]
    return (d.clamp(-w, w) + w), (d.abs() <= w)


def rpr_attention(P, p, x, rm, heads, window, drop, training, conv):
    """RPR_MHA.py:69-128 in banded form (only relative offsets |j - i| <= window contribute).
    x [B,Tp,C] (masked), rm [B,Tp] row mask."""
    B, Tp, C = x.shape
    D = C // heads
    xr = x.reshape(B * Tp, C)
    split = lambda t: t.view(B, Tp, heads, D).transpose(1, 2)                # [B,H,Tp,D]
    q = split(conv(xr, p + ".layer_Dict.Query"))
    k = split(conv(xr, p + ".layer_Dict.Key"))
    v = split(conv(xr, p + ".layer_Dict.Value"))
    relk, relv = P[p + ".weight_K"][0], P[p + ".weight_V"][0]               # [2w+1, D]
    gidx, band = _band_index(Tp, window, x.device)
    gi = gidx.view(1, 1, Tp, Tp).expand(B, heads, Tp, Tp)
    qr = q @ relk.t()                                                        # [B,H,Tp,2w+1]
    rel = torch.gather(qr, 3, gi) * band
    scores = (q @ k.transpose(2, 3) + rel) / math.sqrt(D)                    # RPR_MHA.py:103,109
    amask = (rm.unsqueeze(2) * rm.unsqueeze(1)).unsqueeze(1)                 # Modules.py:558
    scores = scores.masked_fill(amask == 0, -1e4)                            # RPR_MHA.py:117
    pr = F.dropout(torch.softmax(scores, dim=-1), drop, training)            # RPR_MHA.py:119-120
    out = pr @ v
    pb = torch.zeros(B, heads, Tp, 2 * window + 1, device=x.device, dtype=x.dtype)
    pb.scatter_add_(3, gi, pr * band)                                        # RPR_MHA.py:124-126
    out = out + pb @ relv
    out = out.transpose(1, 2).reshape(B * Tp, C)
    return conv(out, p + ".layer_Dict.Projection").view(B, Tp, C)


def encoder_forward(P, hp, tokens, mask, speakers=None, prosodies=None, training=False, prefix="layer_Dict.Encoder",
                    precision=1):
    """Modules.py:262-284 -> mean [B,mel,T], log_std [B,mel,T], log_durations [B,1,T] (channel-first, like the reference)."""
    e = hp.Encoder
    C = e.Channels
    B, T = tokens.shape
    Tp = T + 2 * ROW_PAD
    rm = F.pad(mask.squeeze(1), (ROW_PAD, ROW_PAD))                          # [B,Tp] row mask (0 on pad rows)
    rmf = rm.reshape(-1).contiguous()
    rm3 = rm.unsqueeze(2)

    def conv(xr, name, relu=False, mask_out=False, residual=None):
        return conv_rows(xr, P[name + ".weight"], P.get(name + ".bias"), rmf, relu=relu, mask_out=mask_out,
                         residual=residual, precision=precision)

    x = F.embedding(F.pad(tokens, (ROW_PAD, ROW_PAD)), P[prefix + ".layer_Dict.Embedding.weight"]) * math.sqrt(C) * rm3   # :267
    # Prenet :438-489
    res = x
    for i in range(e.Prenet.Stacks):
        q = f"{prefix}.layer_Dict.Prenet.layer_Dict.CLRD_{i}.layer_Dict"
        h = conv(x.reshape(B * Tp, C), q + ".Conv").view(B, Tp, C)
        x = F.dropout(torch.relu(_ln(P, q + ".LayerNorm", h)), e.Prenet.Dropout_Rate, training) * rm3
    x = conv(x.reshape(B * Tp, C), prefix + ".layer_Dict.Prenet.layer_Dict.Conv1x1", mask_out=True,
             residual=res.reshape(B * Tp, C)).view(B, Tp, C)
    # Transformer :492-573
    dr = e.Transformer.Dropout_Rate
    Fc = e.Transformer.Conv.Calc_Channels
    for i in range(e.Transformer.Stacks):
        q = f"{prefix}.layer_Dict.Transformer.layer_Dict.ANCRDCN_{i}.layer_Dict"
        res = x
        a = rpr_attention(P, q + ".Attention", x, rm, e.Transformer.Attention.Heads, e.Transformer.Attention.Window_Size, dr, training, conv)
        x = _ln(P, q + ".LayerNorm_0", F.dropout(a, dr, training) + res) * rm3
        res = x
        h = conv(x.reshape(B * Tp, C), q + ".Conv_0", relu=True, mask_out=True)
        h = F.dropout(h, dr, training)
        h = conv(h, q + ".Conv_1", mask_out=False)
        h = F.dropout(h, dr, training).view(B, Tp, C)
        x = _ln(P, q + ".LayerNorm_1", h * rm3 + res) * rm3
    xr = x.reshape(B * Tp, C)
    proj = conv(xr, prefix + ".layer_Dict.Project", mask_out=True).view(B, Tp, -1)
    M = hp.Sound.Mel_Dim
    proj = from_rows(proj)
    mean, log_std = proj[:, :M], proj[:, M:]
    # Duration predictor on detached features (:277-282, 602-618)
    d = x.detach()
    cond = None
    if speakers is not None:
        cond = speakers.detach()
    if prosodies is not None:
        cond = prosodies.detach() if cond is None else cond + prosodies.detach()
    if cond is not None:
        d = torch.cat([d, cond.unsqueeze(1).expand(-1, Tp, -1) * rm3], dim=2)
    dp = e.Duration_Predictor
    dr_ = d.reshape(B * Tp, -1).contiguous()
    for i in range(dp.Stacks):
        q = f"{prefix}.layer_Dict.Duration_Predictor.layer_Dict.CRND_{i}.layer_Dict.Conv"
        dr_ = F.dropout(conv(dr_, q, relu=True, mask_out=True), dp.Dropout_Rate, training)
    # Projection to one channel: N = 1 is not a GEMM; a masked dot product per frame
    wq = prefix + ".layer_Dict.Duration_Predictor.layer_Dict.Projection"
    log_dur = (dr_ @ P[wq + ".weight"][0, :, 0] + P[wq + ".bias"]).view(B, Tp)[:, ROW_PAD:-ROW_PAD].unsqueeze(1) * mask
    return mean, log_std, log_dur
