"""Encoder kernels against plain PyTorch references of the same ops (fp32)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def attention_core_ref(qkv, relk, relv, rowmask, B, Tp, H, win):
    """RPR_MHA.py:95-128 on fused rows [B*Tp, 3*H*D] (Q | K | V), every row a position, rowmask 0 on padding."""
    C3 = qkv.shape[1]
    D = C3 // (3 * H)
    x = qkv.view(B, Tp, 3, H, D)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))                      # [B,H,Tp,D]
    scores = q @ k.transpose(2, 3)
    qr = q @ relk.t()                                                             # [B,H,Tp,2w+1]
    idx = torch.arange(Tp, device=qkv.device)
    dmat = idx[None, :] - idx[:, None]
    band = (dmat.abs() <= win)
    gather = (dmat.clamp(-win, win) + win)
    rel = torch.gather(qr, 3, gather.view(1, 1, Tp, Tp).expand(B, H, Tp, Tp)) * band
    scores = (scores + rel) / math.sqrt(D)
    m = rowmask.view(B, Tp)
    scores = scores.masked_fill((m[:, :, None] * m[:, None, :]).unsqueeze(1) == 0, -1e4)
    pr = torch.softmax(scores, dim=-1)
    out = pr @ v
    pb = torch.zeros(B, H, Tp, 2 * win + 1, device=qkv.device, dtype=qkv.dtype)
    for d in range(-win, win + 1):
        lo, hi = max(0, -d), min(Tp, Tp - d)
        ii = torch.arange(lo, hi, device=qkv.device)
        pb[:, :, ii, d + win] = pr[:, :, ii, ii + d]
    out = out + pb @ relv
    return out.transpose(1, 2).reshape(B * Tp, H * D)


@pytest.mark.parametrize("T,D", [(120, 96), (57, 96), (100, 64), (150, 96), (200, 96), (252, 64), (125, 96), (40, 16), (250, 16)])
def test_rpr_attention_core_forward_backward(T, D):
    """Single-workgroup MFMA path (Tp <= 128, D in {64, 96}), the long MFMA path (Tp <= 256: the reference trains on texts of up to 200
    tokens) and the general path, forward and all four gradients."""
    from glow_tts_amd.conv_fn import RPRAttention
    B, H, win = 3, 2, 4
    Tp = T + 4
    g = torch.Generator().manual_seed(T + D)
    lens = torch.tensor([T, T - 9, max(5, T // 3)])
    rowmask = torch.zeros(B, Tp)
    for b in range(B):
        rowmask[b, 2:2 + lens[b]] = 1.0
    rowmask = rowmask.reshape(-1).cuda()
    qkv = (torch.randn(B * Tp, 3 * H * D, generator=g) * 0.5).cuda().requires_grad_(True)
    relk = (torch.randn(1, 2 * win + 1, D, generator=g) * D ** -0.5).cuda().requires_grad_(True)
    relv = (torch.randn(1, 2 * win + 1, D, generator=g) * D ** -0.5).cuda().requires_grad_(True)
    dout = torch.randn(B * Tp, H * D, generator=g).cuda() * rowmask[:, None]
    out = RPRAttention.apply(qkv, relk, relv, rowmask, B, Tp, H, win, 0.0, 0, None)
    out.backward(dout)
    got = [out.detach(), qkv.grad.clone(), relk.grad.clone(), relv.grad.clone()]
    qkv.grad = relk.grad = relv.grad = None
    ref = attention_core_ref(qkv.double(), relk[0].double(), relv[0].double(), rowmask.double(), B, Tp, H, win)
    ref.backward(dout.double())
    want = [ref.detach(), qkv.grad, relk.grad, relv.grad]
    valid = rowmask[:, None] > 0
    for name, a, b_ in zip(("out", "dqkv", "drelK", "drelV"), got, want):
        a, b_ = a.double(), b_.double()
        if name in ("out", "dqkv"):
            a, b_ = a * valid, b_ * valid               # padding rows carry no signal in the model
        err = (a - b_).abs().max().item() / max(1.0, b_.abs().max().item())
        assert err < 2e-5, (name, err)


@pytest.mark.parametrize("T,D", [(120, 96), (57, 96), (200, 96), (40, 16)])
def test_rpr_attention_core_against_the_oracle(T, D):
    """The attention kernels against `oracle.glowtts_ref.rpr_attention` ITSELF (the restatement that is pinned against the imported reference's RPR_MHA), not only
    against this file's local copy of the banded form: Q / K / V convs from random weights, an identity Projection, ragged lengths - a wrong band or offset
    convention (row d + w of weight_K / weight_V = offset d = j - i) shows here, one level below the whole-encoder parity tests."""
    from oracle import glowtts_ref as O
    from glow_tts_amd.conv_fn import RPRAttention
    B, H, win = 3, 2, 4
    C, Tp = H * D, T + 4
    g = torch.Generator().manual_seed(1000 + T + D)
    lens = torch.tensor([T, T - 9, max(5, T // 3)])
    mask = (torch.arange(T)[None] < lens[:, None]).float().unsqueeze(1)                 # [B, 1, T]
    x = torch.randn(B, C, T, generator=g) * 0.5 * mask
    p = "att"
    sd = {}
    for name in ("Query", "Key", "Value"):
        sd[f"{p}.layer_Dict.{name}.weight"] = torch.randn(C, C, 1, generator=g) * C ** -0.5
        sd[f"{p}.layer_Dict.{name}.bias"] = torch.randn(C, generator=g) * 0.1
    sd[f"{p}.layer_Dict.Projection.weight"] = torch.eye(C).unsqueeze(-1)
    sd[f"{p}.layer_Dict.Projection.bias"] = torch.zeros(C)
    sd[f"{p}.weight_K"] = torch.randn(1, 2 * win + 1, D, generator=g) * D ** -0.5
    sd[f"{p}.weight_V"] = torch.randn(1, 2 * win + 1, D, generator=g) * D ** -0.5
    cfg = O.Cfg(enc_channels=C, heads=H, window=win)
    want = O.rpr_attention({k: v.double() for k, v in sd.items()}, p, x.double(), mask.double(), cfg)      # [B, C, T]
    # the kernel's operands: fused Q | K | V rows with two pad rows around every utterance
    q, k, v = (O.conv(sd, f"{p}.layer_Dict.{n}", x) for n in ("Query", "Key", "Value"))  # [B, C, T] each
    rows = torch.zeros(B, Tp, 3 * C)
    rows[:, 2:2 + T] = torch.cat([q, k, v], dim=1).transpose(1, 2)
    rowmask = torch.zeros(B, Tp)
    rowmask[:, 2:2 + T] = mask[:, 0]
    out = RPRAttention.apply(rows.reshape(B * Tp, 3 * C).cuda(), sd[f"{p}.weight_K"].cuda(), sd[f"{p}.weight_V"].cuda(), rowmask.reshape(-1).cuda(),
                             B, Tp, H, win, 0.0, 0, None)
    got = out.view(B, Tp, C)[:, 2:2 + T].transpose(1, 2).cpu().double()
    err = ((got - want) * mask.double()).abs().max().item() / max(1.0, want.abs().max().item())
    assert err < 2e-5, err


@pytest.mark.parametrize("T,D", [(120, 96), (57, 96), (100, 64), (124, 96)])
def test_rpr_attention_core_bf16_mode(T, D):
    """bf16 arithmetic mode of the single-workgroup attention core (the benchmarked configuration: 120 tokens, D = 96): the five contractions
    run on bf16 MFMAs (launch class asserted), softmax in fp32.  Against the fp64 reference: bf16-level agreement of the output and of all
    four gradients (relative to each tensor's largest entry) and a cosine >= 0.999."""
    from glow_tts_amd.conv_fn import RPRAttention
    from glow_tts_amd import ops
    from helpers import launch_counts, launch_reset
    B, H, win = 3, 2, 4
    Tp = T + 4
    g = torch.Generator().manual_seed(T + D)
    lens = torch.tensor([T, T - 9, max(5, T // 3)])
    rowmask = torch.zeros(B, Tp)
    for b in range(B):
        rowmask[b, 2:2 + lens[b]] = 1.0
    rowmask = rowmask.reshape(-1).cuda()
    qkv = (torch.randn(B * Tp, 3 * H * D, generator=g) * 0.5).cuda().requires_grad_(True)
    relk = (torch.randn(1, 2 * win + 1, D, generator=g) * D ** -0.5).cuda().requires_grad_(True)
    relv = (torch.randn(1, 2 * win + 1, D, generator=g) * D ** -0.5).cuda().requires_grad_(True)
    dout = torch.randn(B * Tp, H * D, generator=g).cuda() * rowmask[:, None]
    launch_reset()
    out = RPRAttention.apply(qkv, relk, relv, rowmask, B, Tp, H, win, 0.0, 0, None, ops.BF16)
    out.backward(dout)
    counts = launch_counts()
    assert counts.get(f"attn_fwd_mfma<{D},bf16>", 0) == 1 and counts.get(f"attn_bwd_mfma<{D},bf16>", 0) == 1, counts
    got = [out.detach(), qkv.grad.clone(), relk.grad.clone(), relv.grad.clone()]
    qkv.grad = relk.grad = relv.grad = None
    ref = attention_core_ref(qkv.double(), relk[0].double(), relv[0].double(), rowmask.double(), B, Tp, H, win)
    ref.backward(dout.double())
    want = [ref.detach(), qkv.grad, relk.grad, relv.grad]
    valid = rowmask[:, None] > 0
    for name, a, b_ in zip(("out", "dqkv", "drelK", "drelV"), got, want):
        a, b_ = a.double(), b_.double()
        if name in ("out", "dqkv"):
            a, b_ = a * valid, b_ * valid
        err = (a - b_).abs().max().item() / max(1.0, b_.abs().max().item())
        cos = (a.flatten() @ b_.flatten() / (a.norm() * b_.norm())).item()
        assert err < 2e-2 and cos > 0.999, (name, err, cos)


@pytest.mark.parametrize("T", [120, 200])
def test_rpr_attention_dropout_mask_consistency(T):
    """p > 0: output changes, same seed reproduces it, and the backward uses the forward's keep mask (linear in V: exact check)."""
    from glow_tts_amd.conv_fn import RPRAttention
    B, H, win, D = 2, 2, 4, 96
    Tp = T + 4
    g = torch.Generator().manual_seed(1)
    rowmask = torch.zeros(B, Tp); rowmask[:, 2:T + 2] = 1.0
    rowmask = rowmask.reshape(-1).cuda()
    qkv = (torch.randn(B * Tp, 3 * H * D, generator=g) * 0.5).cuda().requires_grad_(True)
    relk = (torch.randn(1, 2 * win + 1, D, generator=g) * 0.1).cuda()
    relv = (torch.randn(1, 2 * win + 1, D, generator=g) * 0.1).cuda()
    run = lambda x, p: RPRAttention.apply(x, relk, relv, rowmask, B, Tp, H, win, p, 1234, None)
    o0, o1, o2 = run(qkv, 0.0), run(qkv, 0.3), run(qkv, 0.3)
    assert torch.equal(o1, o2) and (o1 - o0).abs().max() > 1e-3 and torch.isfinite(o1).all()
    # the output is linear in V for a fixed mask: <dout, out(V + dV) - out(V)> == <dV-gradient, dV> exactly (up to rounding)
    dout = torch.randn(B * Tp, H * D, generator=g).cuda()
    (o1 * dout).sum().backward()
    dv = torch.zeros_like(qkv); dv[:, 2 * H * D:] = torch.randn(B * Tp, H * D, generator=g).cuda()
    with torch.no_grad():
        delta = ((run(qkv + dv, 0.3) - o1) * dout).sum().item()
    pred = (qkv.grad * dv).sum().item()
    assert abs(delta - pred) <= 1e-3 * max(1.0, abs(pred)), (delta, pred)


def test_expand_prior_forward_backward():
    """ExpandPrior (Modules.py:120-121: mean @ attn with a one-hot monotonic attn) and its gradient (segment sums)."""
    from glow_tts_amd.alignment import ExpandPrior
    g = torch.Generator().manual_seed(4)
    B, C, Tx, Ty = 3, 80, 37, 211
    idx = torch.full((B, Ty), -1, dtype=torch.int32)
    for b, (tx, ty) in enumerate([(37, 211), (20, 150), (5, 5)]):
        cuts = torch.sort(torch.randperm(ty - 1, generator=g)[:tx - 1] + 1).values if tx > 1 else torch.zeros(0, dtype=torch.long)
        bounds = torch.cat([torch.zeros(1, dtype=torch.long), cuts, torch.tensor([ty])])
        for x in range(tx):
            idx[b, bounds[x]:bounds[x + 1]] = x
    src = torch.randn(B, C, Tx, generator=g).cuda().requires_grad_(True)
    dout = torch.randn(B, C, Ty, generator=g).cuda()
    out = ExpandPrior.apply(src, idx.cuda())
    out.backward(dout)
    onehot = (idx.unsqueeze(1) == torch.arange(Tx).view(1, Tx, 1)).float().cuda()           # [B, Tx, Ty]
    want = src.detach() @ onehot
    assert torch.equal(out.detach(), want)
    want_g = dout.double() @ onehot.double().transpose(1, 2)
    assert (src.grad.double() - want_g).abs().max() <= 1e-5 * want_g.abs().max()


@pytest.mark.parametrize("Cm,Tx,Ty,ns", [(80, 120, 800, 2), (80, 37, 96, 2), (12, 9, 30, 2), (160, 70, 64, 1)])
def test_log_prior_operands_and_values(Cm, Tx, Ty, ns):
    """glowtts_logprior_prep (tiled kernel; Cm = 160 takes the plain one) + the SQNEG GEMM reproduce Modules.py:108-114 on ragged lengths:
    log N(z_y; mu_x, sigma_x) for valid (token, frame) pairs, exactly 0 elsewhere; mel lengths rounded down to a multiple of `ns`."""
    from glow_tts_amd import alignment
    torch.manual_seed(11)
    B = 5
    mean, log_std = torch.randn(B, Cm, Tx, device="cuda"), 0.3 * torch.randn(B, Cm, Tx, device="cuda")
    z = torch.randn(B, Cm, Ty, device="cuda")
    tl = torch.tensor([Tx, max(1, Tx // 2), Tx - 1, 3, Tx], device="cuda")
    ml = torch.tensor([Ty, Ty - 1, max(ns, Ty // 3), Ty - 3, ns], device="cuda")
    got, tx32, ty32 = alignment.log_prior_t(mean, log_std, z, tl, ml, ns, return_lengths=True)
    zl = (ml // ns) * ns
    assert tx32.dtype == torch.int32 and torch.equal(tx32.long(), tl) and torch.equal(ty32.long(), zl)
    m64, s64, z64 = mean.double(), log_std.double(), z.double()
    r = torch.exp(-2 * s64)
    want = (-0.5 * math.log(2 * math.pi) - s64).sum(1)[:, :, None] + torch.einsum("bcx,bcy->bxy", r, -0.5 * z64 ** 2) + \
        torch.einsum("bcx,bcy->bxy", m64 * r, z64) + (-0.5 * m64 ** 2 * r).sum(1)[:, :, None]                   # [B, Tx, Ty]
    mask = (torch.arange(Tx, device="cuda")[None, :, None] < tl[:, None, None]) & (torch.arange(Ty, device="cuda")[None, None, :] < zl[:, None, None])
    want = (want * mask).transpose(1, 2)
    assert got.shape == (B, Ty, Tx)
    assert (got[~mask.transpose(1, 2)] == 0).all()
    assert (got.double() - want).abs().max().item() <= 2e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("B,T,D,H", [(5, 13, 256, 128), (3, 5, 8, 8), (2, 1, 12, 20), (4, 40, 64, 341)])
def test_gru_recurrence_matches_torch_gru(B, T, D, H):
    """glowtts_gru_fwd / _bwd (the prosody encoder's GRU, Modules.py:338-343, 371) against torch.nn.GRU's own arithmetic in fp64 on the
    CPU: every step's state and the gradients of the input and of all four parameter tensors, for a loss that touches every step."""
    from glow_tts_amd.prosody import _GRUFunction
    g = torch.Generator().manual_seed(B * 100 + H)
    ref = torch.nn.GRU(D, H, 1, batch_first=True).double()
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn(p.shape, generator=g, dtype=torch.float64) * min(0.3, 1.5 / H ** 0.5))    # (large H: keep the gates out of saturation, where fp32 / fp64 trajectories part)
    x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    w = torch.randn(B, T, H, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    (ref(xr)[0] * w).sum().backward()
    want = ref(x)[0].detach()
    params = [getattr(ref, n).detach().float().cuda().requires_grad_(True) for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
    xg = x.float().cuda().requires_grad_(True)
    hs = _GRUFunction.apply(xg, *params)
    (hs * w.float().cuda()).sum().backward()
    torch.cuda.synchronize()
    assert (hs.detach().cpu().double() - want).abs().max() <= 2e-5
    close = lambda got, ref_: (got.cpu().double() - ref_).abs().max().item() <= 2e-4 * max(1.0, ref_.abs().max().item())
    assert close(xg.grad, xr.grad)
    for p, n in zip(params, ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")):
        assert close(p.grad, getattr(ref, n).grad), n


@pytest.mark.parametrize("B,M,T,chans,precision", [(3, 80, 201, [32, 32, 64, 64, 128, 128], 0), (2, 13, 41, [32, 64, 128], 0), (1, 80, 7, [32, 32], 0),
                                                   (5, 1, 1, [32, 128], 0), (32, 80, 64, [32, 32, 64, 64, 128, 128], 0),
                                                   (3, 80, 201, [32, 32, 64, 64, 128, 128], 1), (2, 13, 41, [64, 32, 128], 1)])
def test_prosody_conv_stack_matches_conv2d(B, M, T, chans, precision):
    """Reference encoder of the GST prosody encoder (Modules.py:320-333, 366-369): six Conv2d(3x3, stride 2, padding 1, no bias) + ReLU.
    HIP path = the direct kernels of csrc/conv2d_ops.hip (round 6: gather implicit GEMM forward / data gradient, exact-fp32 MFMA weight gradient,
    VALU first layer; prosody.conv_stack_hip) against torch's Conv2d in fp64: the GRU input [B, T', C * Mel'] and the gradient of every conv weight
    (odd sizes: the last window hangs over the edge; 1 x 1 images; B = 32: several row tiles and weight-gradient splits).  The mel input is data and
    gets no gradient - as in the reference's use (Modules.py:81-82)."""
    from glow_tts_amd.prosody import conv_stack_hip, conv_stack_supported
    from helpers import launch_counts, launch_reset
    g = torch.Generator().manual_seed(B * 1000 + T)
    convs, cin = [], 1
    for c in chans:
        conv = torch.nn.Conv2d(cin, c, 3, stride=2, padding=1, bias=False)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (9 * cin)) ** 0.5)
        convs.append(conv.cuda())
        cin = c
    mels = torch.randn(B, M, T, generator=g).cuda()
    assert conv_stack_supported(convs, mels)
    launch_reset()
    out = conv_stack_hip(convs, mels, precision, {})
    dout = torch.randn(out.shape, generator=g).cuda()
    out.backward(dout)
    torch.cuda.synchronize()
    counts = launch_counts()
    n = len(chans)
    assert counts.get("conv3x3s2_pack", 0) == 1 and counts.get("conv3x3s2_first_fwd", 0) == 1 and counts.get("conv3x3s2_first_wgrad", 0) == 1, counts
    assert sum(v for k, v in counts.items() if k.startswith("conv3x3s2_fwd<")) == n - 1 and sum(v for k, v in counts.items() if k.startswith("conv3x3s2_wgrad<")) == n - 1, counts
    assert sum(v for k, v in counts.items() if k.startswith("conv3x3s2_dgrad<")) == n - 1 and counts.get("conv3x3s2_wgrad_reduce", 0) == 1, counts
    got = [out.detach().double().cpu()] + [c.weight.grad.double().cpu() for c in convs]
    x = mels.detach().double().cpu()
    ws = [c.weight.detach().double().cpu().requires_grad_(True) for c in convs]
    y = x.unsqueeze(1)
    for w in ws:
        y = torch.relu(torch.nn.functional.conv2d(y, w, None, stride=2, padding=1))
    ref = y.reshape(y.size(0), y.size(1) * y.size(2), y.size(3)).transpose(2, 1)             # Modules.py:369-370
    assert ref.shape == out.shape
    ref.backward(dout.double().cpu())
    want = [ref.detach()] + [w.grad for w in ws]
    for name, a, b_ in zip(["out"] + [f"dW{i}" for i in range(len(ws))], got, want):
        err = (a - b_).abs().max().item() / max(1e-6, b_.abs().max().item())
        cos = (a.flatten() @ b_.flatten() / (a.norm() * b_.norm() + 1e-30)).item()
        if precision == 0:
            assert err < 2e-5, (name, err)
        else:     # bf16 operands through six layers: direction and scale of every gradient, not its worst element (a flipped ReLU moves one)
            assert cos > 0.99 and 0.95 < (a.norm() / b_.norm()).item() < 1.05, (name, err, cos)


def test_prosody_encoder_unsupported_conv_shapes_take_conv2d():
    """Channel counts the direct kernels do not take (yaml values other than 32 / 64 / 128) run torch's Conv2d: same module, same numbers as the reference's
    Conv2d stack in fp64, no HIP conv launch."""
    from glow_tts_amd.prosody import conv_stack_supported
    convs = [torch.nn.Conv2d(1, 4, 3, stride=2, padding=1, bias=False).cuda(), torch.nn.Conv2d(4, 8, 3, stride=2, padding=1, bias=False).cuda()]
    assert not conv_stack_supported(convs, torch.randn(2, 12, 40).cuda())
    assert not conv_stack_supported([torch.nn.Conv2d(1, 32, 5, stride=2, padding=2, bias=False).cuda()], torch.randn(2, 12, 40).cuda())


@pytest.mark.gpu
@pytest.mark.parametrize("drop_p", [0.0, 0.1])
def test_projection_and_layernorm_in_one_launch(drop_p):
    """Round 4: `glowtts_proj_layernorm` (csrc/gemm_cl.hip proj_ln_kernel; Modules.py:560-562) against the two launches it replaces - the
    register-staged 1x1 conv with bias + dropout on the fp32 attention rows, then `glowtts_layernorm_fwd_io` with the residual: the same dropout
    decisions (kept projection: zeros in the same places), s / statistics / y to fp32 rounding of another summation order, the bf16 copy of y within
    one bf16 step; rows that do not fill the last 32-row fragment; and the block function end to end (outputs and every gradient) with the fusion
    on and off."""
    import torch
    from glow_tts_amd import conv_fn as CF, ops, _lib
    L = CF._L()
    torch.manual_seed(5)
    R, C = 32 * 11 + 7, 192
    att, x = torch.randn(R, C, device="cuda"), torch.randn(R, C, device="cuda")
    w, b = torch.randn(C, C, 1, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    rowmask = (torch.rand(R, device="cuda") > 0.2).float()
    pw = ops.pack_weight(w, precision=ops.BF16)
    seed_t = torch.tensor([12345], dtype=torch.int32, device="cuda")
    # the two launches
    proj = torch.empty(R, C, device="cuda")
    CF._conv_launch(att, pw, C, R, 1, ops.F_BIAS | (ops.F_DROPOUT if drop_p > 0 else 0), C, b, rowmask, proj, drop_p=drop_p, seed=77, seed_t=seed_t, a_bf=False)
    y0, yb0, s0, st0 = torch.empty(R, C, device="cuda"), torch.empty(R, C, device="cuda", dtype=torch.bfloat16), torch.empty(R, C, device="cuda"), torch.empty(R, 2, device="cuda")
    _lib.check(L.glowtts_layernorm_fwd_io(proj.data_ptr(), x.data_ptr(), s0.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rowmask.data_ptr(), y0.data_ptr(),
                                          st0.data_ptr(), R, C, 1e-4, 0, 0.0, 0, None, yb0.data_ptr(), _lib.stream()), "ln")
    # one launch
    pk = torch.full((R, C), 3.0, device="cuda")
    y1, yb1, s1, st1 = torch.empty(R, C, device="cuda"), torch.empty(R, C, device="cuda", dtype=torch.bfloat16), torch.empty(R, C, device="cuda"), torch.empty(R, 2, device="cuda")
    _lib.check(L.glowtts_proj_layernorm(att.data_ptr(), C, pw.data.data_ptr(), pw.npad, b.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                        rowmask.data_ptr(), pk.data_ptr() if drop_p > 0 else None, s1.data_ptr(), st1.data_ptr(), y1.data_ptr(), yb1.data_ptr(),
                                        R, C, 1e-4, drop_p, 77, seed_t.data_ptr(), _lib.stream()), "proj_ln")
    torch.cuda.synchronize()
    if drop_p > 0:
        assert torch.equal(pk == 0, proj == 0) and 0.05 < float((proj == 0).float().mean()) < 0.15
        assert (pk - proj).abs().max().item() <= 1e-5 * proj.abs().max().item()
    assert (s1 - s0).abs().max().item() <= 1e-5 * s0.abs().max().item()
    assert (st1 - st0).abs().max().item() <= 1e-4 * st0.abs().max().item()
    assert (y1 - y0).abs().max().item() <= 2e-5 * y0.abs().max().item()
    assert (yb1.float() - yb0.float()).abs().max().item() <= 2 ** -7 * y0.abs().max().item()
    assert bool((y1[rowmask == 0] == 0).all())


@pytest.mark.gpu
def test_attention_block_with_and_without_the_fused_projection():
    import torch
    from glow_tts_amd import conv_fn as CF, ops
    torch.manual_seed(9)
    B, T, C, H, win = 3, 60, 192, 2, 4
    Tp = T + 4
    R = B * Tp
    rowmask = torch.ones(B, Tp, device="cuda")
    rowmask[:, :2] = 0; rowmask[:, -2:] = 0; rowmask[1, 40:] = 0
    rowmask = rowmask.reshape(-1).contiguous()
    res = []
    for fuse in (False, True):
        CF.FUSE["proj_ln"] = fuse
        try:
            torch.manual_seed(3)
            x = (torch.randn(R, C, device="cuda") * rowmask[:, None]).requires_grad_(True)
            wqkv, bqkv = (torch.randn(3 * C, C, 1, device="cuda") * 0.05).requires_grad_(True), torch.zeros(3 * C, device="cuda", requires_grad=True)
            relk, relv = (torch.randn(1, 2 * win + 1, C // H, device="cuda") * 0.1).requires_grad_(True), (torch.randn(1, 2 * win + 1, C // H, device="cuda") * 0.1).requires_grad_(True)
            wp, bp = (torch.randn(C, C, 1, device="cuda") * 0.05).requires_grad_(True), torch.zeros(C, device="cuda", requires_grad=True)
            gamma, beta = torch.ones(C, device="cuda", requires_grad=True), torch.zeros(C, device="cuda", requires_grad=True)
            tape = CF.WgradTape()
            packs = [(ops.pack_weight(w.detach(), precision=ops.BF16), ops.pack_weight(w.detach(), transpose=True, precision=ops.BF16)) for w in (wqkv, wp)]
            seed_t = torch.tensor([7], dtype=torch.int32, device="cuda")
            y, yb = CF.AttentionBlock.apply(x, x.detach().to(torch.bfloat16), wqkv, bqkv, relk, relv, wp, bp, gamma, beta, rowmask, B, Tp, H, win, 0.1, (11, 13),
                                            seed_t, tape, packs[0], packs[1])
            (y * torch.linspace(-1, 1, C, device="cuda")).sum().backward()
            tape.flush()
            torch.cuda.synchronize()
            # (the weight / LayerNorm-parameter gradients are the tape's deferred launches: same operands either way; compared here: what the chain itself produces)
            res.append([y.detach().clone(), yb.float()] + [t.grad.clone() for t in (x, relk, relv)])
        finally:
            CF.FUSE["proj_ln"] = True
    for a, b in zip(res[0], res[1]):
        assert (a - b).abs().max().item() <= 2e-3 * max(1e-6, a.abs().max().item())


@pytest.mark.gpu
def test_layernorm_and_next_qkv_in_one_launch():
    """Round 4: `glowtts_layernorm_qkv` (csrc/gemm_cl.hip ln_qkv_kernel; Modules.py:571 -> RPR_MHA.py:82-84) against `glowtts_layernorm_fwd_io`
    followed by the fused Q / K / V 1x1 conv on its bf16 rows: s / statistics / y to fp32 rounding of another summation order, y_bf16 within one bf16
    step, qkv within the bf16 operand rounding of the few rows where y_bf16 differs by that step; rows that do not fill the last fragment."""
    import torch
    from glow_tts_amd import conv_fn as CF, ops, _lib
    L = CF._L()
    torch.manual_seed(21)
    R, C = 32 * 9 + 5, 192
    a, b = torch.randn(R, C, device="cuda"), torch.randn(R, C, device="cuda")
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    rowmask = (torch.rand(R, device="cuda") > 0.2).float()
    wq, bq = torch.randn(3 * C, C, 1, device="cuda") * 0.1, torch.randn(3 * C, device="cuda") * 0.1
    pw = ops.pack_weight(wq, precision=ops.BF16)
    y0, yb0, s0, st0 = torch.empty(R, C, device="cuda"), torch.empty(R, C, device="cuda", dtype=torch.bfloat16), torch.empty(R, C, device="cuda"), torch.empty(R, 2, device="cuda")
    _lib.check(L.glowtts_layernorm_fwd_io(a.data_ptr(), b.data_ptr(), s0.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rowmask.data_ptr(), y0.data_ptr(),
                                          st0.data_ptr(), R, C, 1e-4, 0, 0.0, 0, None, yb0.data_ptr(), _lib.stream()), "ln")
    q0 = torch.empty(R, 3 * C, device="cuda")
    CF._conv_launch(yb0, pw, C, R, 1, ops.F_BIAS, 3 * C, bq, rowmask, q0)
    y1, yb1, s1, st1, q1 = (torch.empty(R, C, device="cuda"), torch.empty(R, C, device="cuda", dtype=torch.bfloat16), torch.empty(R, C, device="cuda"),
                            torch.empty(R, 2, device="cuda"), torch.empty(R, 3 * C, device="cuda"))
    _lib.check(L.glowtts_layernorm_qkv(a.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rowmask.data_ptr(), s1.data_ptr(), st1.data_ptr(),
                                       y1.data_ptr(), yb1.data_ptr(), pw.data.data_ptr(), pw.npad, bq.data_ptr(), q1.data_ptr(), R, C, 1e-4, _lib.stream()), "ln_qkv")
    torch.cuda.synchronize()
    assert torch.equal(s1, s0)
    assert (st1 - st0).abs().max().item() <= 1e-5 * st0.abs().max().item()
    assert (y1 - y0).abs().max().item() <= 2e-5 * y0.abs().max().item()
    assert (yb1.float() - yb0.float()).abs().max().item() <= 2 ** -7 * y0.abs().max().item()
    # qkv of the fused launch against an fp32 product of ITS OWN bf16 rows (bf16-rounded weights, fp32 accumulation)
    ref = yb1.float() @ wq[:, :, 0].to(torch.bfloat16).float().t() + bq
    assert (q1 - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    assert (q1 - q0).abs().max().item() <= 2e-2 * q0.abs().max().item()
