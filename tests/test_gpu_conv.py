"""GPU parity: the channels-last implicit-GEMM conv (glowtts_conv_cl, through the C ABI) vs torch
fp32 conv1d on the CPU (= what the oracle calls).  f32 mode: 2e-5 relative; bf16 mode: 2e-2 of the
output scale (operands rounded to 8 mantissa bits)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {0: 2e-5, 1: 2e-2}


def rows_layout(x):
    """[B,C,T] -> rows [B*(T+4), Cp] with 2 zero rows before/after every utterance.  Cp = C rounded up to 8: the kernel
    reads whole 16-byte K slots, so a row must be at least round_up(C, 8) floats wide (include/glowtts_hip.h)."""
    B, C, T = x.shape
    Cp = (C + 7) // 8 * 8
    r = torch.zeros(B, T + 4, Cp)
    r[:, 2:T + 2, :C] = x.transpose(1, 2)
    return r.reshape(B * (T + 4), Cp)


def from_rows(r, B, T):
    return r.reshape(B, T + 4, -1)[:, 2:T + 2].transpose(1, 2)


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("B,T,C,O,K", [(2, 50, 32, 64, 5), (3, 37, 12, 32, 1), (2, 400, 192, 384, 5), (4, 130, 192, 192, 1),
                                       (2, 64, 80, 192, 1), (1, 33, 48, 32, 3), (2, 120, 768, 192, 3)])
def test_linear_conv(prec, B, T, C, O, K):
    from glow_tts_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + T + C + O + K)
    x = torch.randn(B, C, T, generator=g)
    w = torch.randn(O, C, K, generator=g) / (C * K) ** 0.5
    b = torch.randn(O, generator=g)
    want = F.conv1d(x, w, b, padding=(K - 1) // 2)
    a = rows_layout(x).cuda()
    pw = ops.pack_weight(w.cuda(), precision=prec)
    R = a.shape[0]
    out = torch.full((R, O), float("nan"), device="cuda")
    ops.conv_cl(a, pw, C, R, pad=(K - 1) // 2, epi=ops.EPI_LINEAR, flags=ops.F_BIAS, bias=b.cuda(), out0=out, ld0=O)
    torch.cuda.synchronize()
    got = from_rows(out.cpu(), B, T)
    err = (got - want).abs().max().item()
    assert err <= TOL[prec] * want.abs().max().item(), err


@pytest.mark.parametrize("prec", [0, 1])
def test_dgrad_is_transposed_conv(prec):
    """Data gradient of conv1d = the same kernel with weights packed transposed + taps flipped."""
    from glow_tts_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, C, O, K = 2, 70, 64, 96, 5
    x = torch.randn(B, C, T, generator=g, requires_grad=True)
    w = torch.randn(O, C, K, generator=g) / (C * K) ** 0.5
    dy = torch.randn(B, O, T, generator=g)
    F.conv1d(x, w, None, padding=2).backward(dy)
    pw = ops.pack_weight(w.cuda(), transpose=True, precision=prec)
    a = rows_layout(dy).cuda()
    R = a.shape[0]
    out = torch.zeros((R, C), device="cuda")
    ops.conv_cl(a, pw, O, R, pad=2, out0=out, ld0=C)
    torch.cuda.synchronize()
    got = from_rows(out.cpu(), B, T)
    assert (got - x.grad).abs().max().item() <= TOL[prec] * x.grad.abs().max().item()


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("H,C", [(192, 192), (32, 32), (80, 48)])
def test_gate_epilogue_and_pairmul(prec, H, C):
    """In_i conv + conditioning + tanh*sigmoid (Modules.py:861-870), then Res_Skip on acts (:871-881)."""
    from glow_tts_amd import ops
    g = torch.Generator().manual_seed(H + C)
    B, T, K = 2, 45, 5
    x = torch.randn(B, C, T, generator=g)
    w = torch.randn(2 * H, C, K, generator=g) / (C * K) ** 0.5
    b = torch.randn(2 * H, generator=g) * 0.1
    cond = torch.randn(B, 2 * H, generator=g) * 0.3
    ins = F.conv1d(x, w, b, padding=2) + cond.unsqueeze(2)
    ta, sg = torch.tanh(ins[:, :H]), torch.sigmoid(ins[:, H:])
    a = rows_layout(x).cuda()
    R = a.shape[0]
    pw = ops.pack_weight(w.cuda(), perm=ops.PERM_PAIR, perm_h=H, precision=prec)
    G = torch.full((R, 2 * H), float("nan"), device="cuda")
    ops.conv_cl(a, pw, C, R, pad=2, epi=ops.EPI_GATE, h=H, n=2 * H, rows_per_utt=T + 4, bias=b.cuda(), cond=cond.cuda(),
                ldcond=2 * H, out0=G, ld0=2 * H)
    torch.cuda.synchronize()
    Gc = from_rows(G.cpu(), B, T)                                  # [B, 2H, T] interleaved (tanh, sigmoid)
    tol = 1e-5 if prec == 0 else 2e-2
    assert (Gc[:, 0::2] - ta).abs().max() <= tol and (Gc[:, 1::2] - sg).abs().max() <= tol

    # Res_Skip consuming the gates through the PAIRMUL prologue
    w2 = torch.randn(2 * H, H, 1, generator=g) / H ** 0.5
    b2 = torch.randn(2 * H, generator=g) * 0.1
    hid = torch.randn(B, H, T, generator=g)
    mask = (torch.arange(T)[None, :] < torch.tensor([T, T - 7])[:, None]).float()
    rs = F.conv1d(ta * sg, w2, b2)
    want_x = (hid + rs[:, :H]) * mask.unsqueeze(1)
    want_skip = rs[:, H:]
    pw2 = ops.pack_weight(w2.cuda(), precision=prec)
    hrows = rows_layout(hid).cuda()
    rowmask = torch.zeros(B, T + 4); rowmask[:, 2:T + 2] = mask
    xo = torch.full((R, H), float("nan"), device="cuda")
    skip = torch.full((R, H), float("nan"), device="cuda")
    Gexact = rows_layout(torch.stack([ta, sg], 2).reshape(B, 2 * H, T)).cuda()
    ops.conv_cl(Gexact, pw2, H, R, lda=2 * H, apro=ops.APRO_PAIRMUL, epi=ops.EPI_RESSKIP, flags=ops.F_FIRST, h=H, n=2 * H,
                bias=b2.cuda(), rowmask=rowmask.reshape(-1).cuda(), out0=xo, ld0=H, out1=skip, ld1=H, in0=hrows, ldi0=H)
    torch.cuda.synchronize()
    s = TOL[prec] * rs.abs().max().item()
    assert (from_rows(xo.cpu(), B, T) - want_x).abs().max() <= s
    assert (from_rows(skip.cpu(), B, T) - want_skip).abs().max() <= s
