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


@pytest.mark.gpu
@pytest.mark.parametrize("H,K", [(192, 5), (64, 3), (40, 1)])
def test_bf16_activation_storage_matches_fp32_storage(H, K):
    """glowtts_conv_args.io_flags / glowtts_wgrad_args.io_flags: the same bf16 MFMA contraction with the activations stored
    as bf16 tensors.  Inputs are pre-rounded to bf16, so only the stored OUTPUT may differ - by one bf16 rounding."""
    from glow_tts_amd import ops
    from glow_tts_amd.conv_fn import wgrad
    g = torch.Generator().manual_seed(H * 7 + K)
    B, T = 3, 61
    R = B * (T + 4)
    bf = lambda t: t.to(torch.bfloat16)
    rowmask = torch.zeros(B, T + 4); rowmask[:, 2:T + 2] = 1; rowmask[1, T - 9:] = 0
    rm = rowmask.reshape(-1).cuda()
    hs = (torch.randn(R, H, generator=g).cuda() * rm[:, None]).to(torch.bfloat16)
    w_in = (torch.randn(2 * H, H, K, generator=g) / (H * K) ** 0.5).cuda()
    b_in = (torch.randn(2 * H, generator=g) * 0.1).cuda()
    pw = ops.pack_weight(w_in, perm=ops.PERM_PAIR, perm_h=H, precision=ops.BF16)
    # GATE: A bf16 -> gates bf16
    G32 = torch.zeros(R, 2 * H, device="cuda")
    G16 = torch.zeros(R, 2 * H, device="cuda", dtype=torch.bfloat16)
    kw = dict(pad=(K - 1) // 2, epi=ops.EPI_GATE, h=H, n=2 * H, rows_per_utt=T + 4, bias=b_in, ld0=2 * H)
    ops.conv_cl(hs.float(), pw, H, R, out0=G32, **kw)
    ops.conv_cl(hs, pw, H, R, out0=G16, io_flags=ops.IO_A_BF16 | ops.IO_OUT0_BF16, **kw)
    assert (bf(G32).float() - G16.float()).abs().max() <= 2 ** -7      # one bf16 ulp of a value in (-1, 1)
    # RESSKIP: gates bf16 (PAIRMUL), residual in / out bf16, skip fp32
    w_rs = (torch.randn(2 * H, H, 1, generator=g) / H ** 0.5).cuda()
    b_rs = (torch.randn(2 * H, generator=g) * 0.1).cuda()
    pw2 = ops.pack_weight(w_rs, precision=ops.BF16)
    x32, s32 = torch.zeros(R, H, device="cuda"), torch.zeros(R, H, device="cuda")
    x16, s16 = torch.zeros(R, H, device="cuda", dtype=torch.bfloat16), torch.zeros(R, H, device="cuda")
    kw = dict(lda=2 * H, apro=ops.APRO_PAIRMUL, epi=ops.EPI_RESSKIP, flags=ops.F_FIRST, h=H, n=2 * H, bias=b_rs, rowmask=rm, ld0=H, ld1=H, ldi0=H)
    ops.conv_cl(G16.float(), pw2, H, R, out0=x32, out1=s32, in0=hs.float(), **kw)
    ops.conv_cl(G16, pw2, H, R, out0=x16, out1=s16, in0=hs, io_flags=ops.IO_A_BF16 | ops.IO_IN0_BF16 | ops.IO_OUT0_BF16, **kw)
    assert torch.equal(bf(x32), x16) and torch.equal(s32, s16)
    # DGATE: gates read as bf16, gate gradients written as bf16
    pw3 = ops.pack_weight(w_rs, transpose=True, precision=ops.BF16)
    dres, dskip = torch.randn(R, H, generator=g).cuda(), torch.randn(R, H, generator=g).cuda()
    d32 = torch.zeros(R, pw.npad, device="cuda")
    d16 = torch.zeros(R, pw.npad, device="cuda", dtype=torch.bfloat16)
    kw = dict(a2=dskip, lda2=H, ca1=H, epi=ops.EPI_DGATE, n=H, h=H, ldi0=2 * H, ld0=pw.npad)
    ops.conv_cl(dres, pw3, 2 * H, R, out0=d32, in0=G16.float(), **kw)
    ops.conv_cl(dres, pw3, 2 * H, R, out0=d16, in0=G16, io_flags=ops.IO_IN0_BF16 | ops.IO_OUT0_BF16, **kw)
    assert torch.equal(bf(d32), d16)
    # In data gradient: A = gate gradients bf16
    pw4 = ops.pack_weight(w_in, transpose=True, perm=ops.PERM_PAIR, perm_h=H, precision=ops.BF16)
    o32, o16 = torch.zeros(R, H, device="cuda"), torch.zeros(R, H, device="cuda")
    kw = dict(pad=(K - 1) // 2, epi=ops.EPI_LINEAR, flags=ops.F_MASK, n=H, rowmask=rm, ld0=H)
    ops.conv_cl(d16.float(), pw4, pw.npad, R, out0=o32, **kw)
    ops.conv_cl(d16, pw4, pw.npad, R, out0=o16, io_flags=ops.IO_A_BF16, **kw)
    assert torch.allclose(o32, o16, rtol=1e-5, atol=1e-5)        # (the two storage types may run different kernels: K-chunk order differs)
    # weight gradients: (DY, X) = (gate gradients, state) bf16; X = gates bf16 with PAIRMUL
    a32 = wgrad(d16.float(), hs.float(), pw.npad, H, K, ops.BF16, splits=1)
    a16 = wgrad(d16, hs, pw.npad, H, K, ops.BF16, splits=1, io_flags=ops.WIO_DY_BF16 | ops.WIO_X_BF16)
    assert torch.equal(a32[0], a16[0]) and torch.allclose(a32[1], a16[1], rtol=1e-6, atol=1e-5)
    b32 = wgrad(dskip, G16.float(), H, H, 1, ops.BF16, splits=1, xpro=ops.APRO_PAIRMUL)
    b16 = wgrad(dskip, G16, H, H, 1, ops.BF16, splits=1, xpro=ops.APRO_PAIRMUL, io_flags=ops.WIO_X_BF16)
    assert torch.equal(b32[0], b16[0]) and torch.equal(b32[1], b16[1])
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [0, 1])
def test_epilogue_stores_stay_inside_rows(prec):
    """The epilogues rely on the buffer-descriptor bounds check for the rows of the last tile that lie past `rows`:
    nothing beyond rows * ld may be written (LINEAR, GATE, RESSKIP, DGATE)."""
    from glow_tts_amd import ops
    g = torch.Generator().manual_seed(5)
    H, K, Rbuf, R = 64, 3, 400, 301                      # 301 rows: the last 128-row tile is partial
    S = 7.0
    a = torch.randn(Rbuf, H, generator=g).cuda()
    w = (torch.randn(2 * H, H, K, generator=g) / (H * K) ** 0.5).cuda()
    b = torch.randn(2 * H, generator=g).cuda()
    rm = torch.ones(Rbuf, device="cuda")
    pw = ops.pack_weight(w, perm=ops.PERM_PAIR, perm_h=H, precision=prec)
    G = torch.full((Rbuf, 2 * H), S, device="cuda")
    ops.conv_cl(a, pw, H, R, pad=1, epi=ops.EPI_GATE, h=H, n=2 * H, rows_per_utt=R, bias=b, out0=G, ld0=2 * H)
    assert torch.all(G[R:] == S) and not torch.any(G[:R] == S)
    pwl = ops.pack_weight(w, precision=prec)
    Y = torch.full((Rbuf, 2 * H), S, device="cuda")
    ops.conv_cl(a, pwl, H, R, pad=1, flags=ops.F_BIAS | ops.F_MASK, bias=b, rowmask=rm, out0=Y, ld0=2 * H)
    assert torch.all(Y[R:] == S) and not torch.any(Y[:R] == S)
    w2 = (torch.randn(2 * H, H, 1, generator=g) / H ** 0.5).cuda()
    pw2 = ops.pack_weight(w2, precision=prec)
    xo, sk = torch.full((Rbuf, H), S, device="cuda"), torch.full((Rbuf, H), S, device="cuda")
    ops.conv_cl(G, pw2, H, R, lda=2 * H, apro=ops.APRO_PAIRMUL, epi=ops.EPI_RESSKIP, flags=ops.F_FIRST, h=H, n=2 * H, bias=b, rowmask=rm,
                out0=xo, ld0=H, out1=sk, ld1=H, in0=a, ldi0=H)
    assert torch.all(xo[R:] == S) and torch.all(sk[R:] == S) and not torch.any(xo[:R] == S) and not torch.any(sk[:R] == S)
    pw3 = ops.pack_weight(w2, transpose=True, precision=prec)
    d = torch.full((Rbuf, pw.npad), S, device="cuda")
    ops.conv_cl(a, pw3, 2 * H, R, a2=a, lda2=H, ca1=H, epi=ops.EPI_DGATE, n=H, h=H, in0=G, ldi0=2 * H, out0=d, ld0=pw.npad)
    assert torch.all(d[R:] == S) and not torch.any(d[:R, :2 * H] == S)
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [0, 1])            # ops.F32, ops.BF16
def test_pack_set_equals_single_packs(precision):
    """glowtts_pack_weight_multi (one launch, heterogeneous shapes, forward + transposed) writes the same bytes as glowtts_pack_weight."""
    from glow_tts_amd import ops
    torch.manual_seed(3)
    ws = {"a": torch.randn(768, 192, 3, device="cuda"), "b": torch.randn(192, 768, 3, device="cuda"), "c": torch.randn(576, 192, 1, device="cuda"),
          "d": torch.randn(40, 20, 5, device="cuda")}
    items = [(k, w, tr) for k, w in ws.items() for tr in (False, True)]
    ps = ops.PackSet(items, precision)
    ps.run()
    for k, w in ws.items():
        fwd, tr = ps.get(k)
        for got, t in ((fwd, False), (tr, True)):
            ref = ops.pack_weight(w, transpose=t, precision=precision)
            assert (got.npad, got.kchunks, got.taps, got.n) == (ref.npad, ref.kchunks, ref.taps, ref.n)
            assert torch.equal(got.data, ref.data)


@pytest.mark.gpu
@pytest.mark.parametrize("bf_io,taps", [(True, 5), (True, 1), (False, 1), (False, 3)])
def test_wgrad_wide_staging_is_bit_identical(bf_io, taps):
    """GLOWTTS_WIO_WIDE (8 channels per staged item) changes how operands reach LDS, not the arithmetic: grouped weight gradients and
    equal the 4-channel staging bit for bit (bias sums to rounding: other partial-sum grouping), for bf16- and fp32-stored operands,
    ragged tile edges included."""
    from glow_tts_amd import decoder as D, ops
    D._L()
    torch.manual_seed(9)
    R = 1000
    shapes = [(192, 192), (384, 192), (192, 80), (200, 72)]            # (m, ca): multiples of 8, not of the 128 x 64 tile
    dt = torch.bfloat16 if bf_io else torch.float32
    io = (ops.WIO_DY_BF16 | ops.WIO_X_BF16) if bf_io else 0
    dys = [torch.randn(R, m, device="cuda").to(dt) for m, _ in shapes]
    xs = [torch.randn(R, ca, device="cuda").to(dt) for _, ca in shapes]
    outs = []
    for wide in (False, True):
        g = D.WgradGroup(R, taps, ops.BF16, io_flags=io, tag=f"t{int(wide)}")
        dws = [torch.full((m, ca, taps), 7.0, device="cuda") for m, ca in shapes]
        dbs = [torch.full((m,), 7.0, device="cuda") for m, _ in shapes]
        for dy, x, dw, db, (m, ca) in zip(dys, xs, dws, dbs, shapes):
            g.add(dy.data_ptr(), m, m, x.data_ptr(), ca, ca, dw.data_ptr(), db.data_ptr())
        assert g._wide
        g._wide, g._dma = wide, False          # (the staged kernel's two staging widths; the DMA kernel has its own test below)
        g.end_segment(); g.upload(torch.device("cuda")); g.launch_segment(0)
        torch.cuda.synchronize()
        outs.append((dws, dbs))
    for a, b in zip(outs[0][0], outs[1][0]):                         # MFMA operands and K order are the same: identical bits
        assert torch.equal(a, b)
    for a, b in zip(outs[0][1], outs[1][1]):                         # bias sums: the per-thread partial sums group the rows differently
        assert (a - b).abs().max().item() <= 1e-4 * max(1.0, a.abs().max().item())
    ref = torch.einsum("ro,rc->oc", dys[3].float(), xs[3].float()) if taps == 1 else None
    if ref is not None:
        assert (outs[1][0][3][:, :, 0] - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("taps,R", [(5, 1212), (1, 1216), (3, 640), (5, 64)])
def test_wgrad_dma_kernel_matches_the_staged_kernel_and_float64(taps, R):
    """Round 4: `wgrad_dma_kernel` (LDS-DMA staging into fragment-ordered / padded-row tiles, 16x16x32 MFMAs on 96 x 32 x taps wave tiles; GLOWTTS_WIO_DMA)
    against the register-staged kernel on the same grouped problems (same bf16 operands, fp32 accumulation in another order: 1e-5 of the tensor's
    largest entry) and against a float64 contraction.  Rows not a multiple of the 64-row step (the last step and the tap halo read zeros through the
    buffer descriptor), PAIR-permuted output channels, bias sums beside the MFMAs, several jobs per launch, shapes that do not fill a tile (the End conv's 2 x 80 PAIR-packed output channels,
    the Start conv's 80 input channels: the tile reads on into the next row and the store masks it)."""
    from glow_tts_amd import decoder as D, ops
    D._L()
    torch.manual_seed(17)
    pad = (taps - 1) // 2
    shapes = [(384, 192, ops.PERM_PAIR, 192), (192, 64, ops.PERM_NONE, 0), (576, 128, ops.PERM_PAIR, 288), (192, 192, ops.PERM_PAIR, 80), (192, 80, ops.PERM_NONE, 0), (200, 72, ops.PERM_NONE, 0)]
    io = ops.WIO_DY_BF16 | ops.WIO_X_BF16
    dys = [(torch.randn(R, m, device="cuda") * 0.5).to(torch.bfloat16) for m, _, _, _ in shapes]
    xs = [torch.randn(R, ca, device="cuda").to(torch.bfloat16) for _, ca, _, _ in shapes]
    outs = []
    for dma in (False, True):
        g = D.WgradGroup(R, taps, ops.BF16, io_flags=io, tag=f"d{int(dma)}")
        dws = [torch.full((m, ca, taps), 7.0, device="cuda") for m, ca, _, _ in shapes]
        dbs = [torch.full((m,), 7.0, device="cuda") for m, _, _, _ in shapes]
        for dy, x, dw, db, (m, ca, perm, ph) in zip(dys, xs, dws, dbs, shapes):
            g.add(dy.data_ptr(), m, m, x.data_ptr(), ca, ca, dw.data_ptr(), db.data_ptr(), perm=perm, perm_h=ph)
        assert g._wide
        g._dma = dma                          # (decoder.TUNE decides per tap count in the product; forced here)
        from helpers import launch_counts, launch_reset
        launch_reset()
        g.end_segment(); g.upload(torch.device("cuda")); g.launch_segment(0)
        torch.cuda.synchronize()
        lc = launch_counts()
        assert (sum(n for k, n in lc.items() if k.startswith("wgrad_dma<")) == 1) == dma, lc
        outs.append((dws, dbs))
    for i, (m, ca, perm, ph) in enumerate(shapes):
        a, b = outs[0][0][i], outs[1][0][i]
        assert (a - b).abs().max().item() <= 1e-5 * a.abs().max().item(), (i, (a - b).abs().max().item())
        assert (outs[0][1][i] - outs[1][1][i]).abs().max().item() <= 1e-4 * max(1.0, outs[0][1][i].abs().max().item())
        # float64: dW[o][c][t] = sum_r DY[r][col(o)] X[r + t - pad][c]; PAIR: output channel h * H + 32 p + j sits in DY column 64 p + 32 h + j
        dy64, x64 = dys[i].double(), xs[i].double()
        nout = m
        if perm == ops.PERM_PAIR:
            nout = 2 * ph                                              # (2 x 80 output channels sit in 192 PAIR-packed columns)
            o = torch.arange(nout, device="cuda")
            h, jj = o // ph, o % ph
            col = (jj // 32) * 64 + h * 32 + jj % 32
            dy64 = dy64[:, col]
        xp = torch.nn.functional.pad(x64, (0, 0, pad, pad))
        ref = torch.stack([dy64.t() @ xp[t:t + R] for t in range(taps)], dim=2)
        assert (b[:nout].double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
        assert (outs[1][1][i][:nout].double() - dy64.sum(0)).abs().max().item() <= 1e-4 * max(1.0, dy64.sum(0).abs().max().item())
        assert bool((b[nout:] == 7.0).all()) and bool((outs[1][1][i][nout:] == 7.0).all())          # nothing written past the real output channels
