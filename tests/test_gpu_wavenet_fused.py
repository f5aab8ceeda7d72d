"""GPU parity of the FUSED coupling-network kernel (csrc/wavenet_fused.hip: Start conv .. End conv + affine coupling of a flow in one
launch, Modules.py:785-806 / 858-887) against the per-conv launches it replaces, at the benchmarked width (C = 160, 192 channels, 4 layers,
k = 5).  Both paths round the same tensors to bf16 (WaveNet state, gates, tanh * sigmoid) and differ only in fp32 accumulation order, so a
kept bf16 tensor may differ by an occasional one-ulp flip and what follows from it; a halo / tile-seam bug would show as O(1) errors on the
rows next to a multiple of the 52-row tile.  The oracle comparisons of the fused path itself are tests/test_gpu_decoder_fullwidth.py and
tests/test_gpu_fullsize.py (which run it by default)."""
import pytest
import torch

from helpers import full_width_state, launch_counts, launch_reset

pytestmark = pytest.mark.gpu


def _setup(n_flows, lengths, tm, spk_dim=0, seed=0):
    from glow_tts_amd import decoder as D
    g = torch.Generator().manual_seed(seed)
    cfg, sd = full_width_state(n_flows, g, spk_dim=spk_dim)
    dc = D.DecoderConfig(cfg.mel_dim, cfg.n_flows, cfg.n_squeeze, cfg.n_split, cfg.wn_channels, cfg.wn_layers, cfg.wn_kernel, 1)
    P = {k: v.cuda() for k, v in sd.items()}
    B = len(lengths)
    mels = (torch.randn(B, 80, tm, generator=g) * 1.5).clamp(-4, 4).cuda()
    ml = torch.tensor(lengths).cuda()
    spk = None
    if spk_dim:
        spk = torch.randn(B, spk_dim, generator=g)
        spk = (spk / spk.norm(dim=1, keepdim=True)).cuda()
    W = dict(zip(D.WEIGHT_KEYS, [w.contiguous() for w in D.stack_decoder_weights(P, dc)]))
    cond = D.conditioning(P, dc, speakers=spk) if spk is not None else None
    return D, dc, W, mels, ml, cond


def _forward(D, dc, W, mels, ml, cond, fused, drop_p=0.0, seed=None):
    D.TUNE["fused_wn"] = fused
    try:
        with torch.no_grad():
            prep = D._Prepared(dc, W, need_bwd=False, cond=cond)
            launch_reset()
            z, logdet, buf, rowmask, T, _ = D._run_forward(dc, prep, mels, ml, drop_p, seed)
            torch.cuda.synchronize()
            return z, logdet, buf, rowmask, launch_counts()
    finally:
        D.TUNE["fused_wn"] = True


def _outs_columns(c2=80):
    """Columns of the kept (m, logs) buffer: packed column p * 64 + h * 32 + j holds half h (0 = m, 1 = logs) of channel p * 32 + j < c2."""
    m = [p * 64 + j for p in range(3) for j in range(32) if p * 32 + j < c2]
    return torch.tensor(m, device="cuda"), torch.tensor([c + 32 for c in m], device="cuda")


def _close(a, b, what, rowmask=None, atol=0.0, frac=0.0, big=None):
    """|a - b| <= atol everywhere except on at most `frac` of the elements (bf16 one-ulp flips), where it stays <= big."""
    a, b = a.float(), b.float()
    if rowmask is not None:
        m = rowmask.view(-1, *([1] * (a.dim() - 1)))
        a, b = a * m, b * m
    d = (a - b).abs()
    bad = (d > atol).float().mean().item()
    assert bad <= frac, (what, "fraction above atol", bad, "max", d.max().item())
    if big is not None:
        assert d.max().item() <= big, (what, d.max().item())


@pytest.mark.parametrize("lengths,tm", [([640, 522, 240, 2], 640), ([800] * 3, 800), ([104], 104), ([2], 2), ([422, 36, 36, 36, 36, 800], 800)])
def test_fused_forward_matches_per_conv_launches(lengths, tm):
    D, dc, W, mels, ml, _ = _setup(3, lengths, tm, seed=11)
    # the fused launch keeps the fp32 skip rows only when asked to (round 5: the training step's backward reads the bf16 copy; a->skip = NULL drops
    # the stores): the comparison below wants them, and the default - not kept - must change nothing else, bit for bit
    zd, ldd, bd, _, _ = _forward(D, dc, W, mels, ml, None, True)
    assert not any(bd.keep_skip32)
    D.TUNE["drop_skip32"] = False
    try:
        zf, ldf, bf, rm, cf = _forward(D, dc, W, mels, ml, None, True)
    finally:
        D.TUNE["drop_skip32"] = True
    cm_, cl_ = _outs_columns()                # (the pad columns of the kept (m, logs) rows are never written)
    assert all(bf.keep_skip32) and torch.equal(zd, zf) and torch.equal(ldd, ldf) and torch.equal(bd.skipb, bf.skipb)
    assert torch.equal(bd.outs[:, :, cm_], bf.outs[:, :, cm_]) and torch.equal(bd.outs[:, :, cl_], bf.outs[:, :, cl_])
    zu, ldu, bu, _, cu = _forward(D, dc, W, mels, ml, None, False)
    assert cf.get("wn_fwd<nodrop>", 0) == 3 and not any(k.startswith("conv_dma") or k.startswith("conv_chain") for k in cf), cf
    assert any(k.startswith("conv_chain<RESSKIP,COUPLE>") for k in cu) and not any(k.startswith("wn_fwd") for k in cu), cu
    # kept activations of flow 0 see identical inputs: only accumulation-order noise and its one-ulp consequences
    _close(bf.hs[0, 0], bu.hs[0, 0], "h0", rm, atol=0.0, frac=2e-3, big=0.05)
    for l in range(4):
        _close(bf.gates[0, l], bu.gates[0, l], f"gates{l}", rm, atol=1e-6, frac=2e-2 * (l + 1), big=0.05)
        _close(bf.actp[0, l], bu.actp[0, l], f"acts{l}", rm, atol=1e-6, frac=2e-2 * (l + 1), big=0.05)
        _close(bf.hs[0, l], bu.hs[0, l], f"hs{l}", rm, atol=1e-6, frac=2e-2 * (l + 1), big=0.1)
    _close(bf.skip[0], bu.skip[0], "skip", rm, atol=2e-2, frac=1e-3, big=0.2)
    cm, cl = _outs_columns()                  # PAIR-packed (m | logs) per 32 channels; the pad columns are never written
    _close(bf.outs[0][:, cm], bu.outs[0][:, cm], "m", rm, atol=5e-3, frac=2e-3, big=0.05)
    _close(bf.outs[0][:, cl], bu.outs[0][:, cl], "logs", rm, atol=5e-3, frac=2e-3, big=0.05)
    # pad rows / masked frames are exact zeros in both
    assert (bf.x[1] * (1 - rm).unsqueeze(1)).abs().max() == 0
    # end to end
    valid = (torch.arange(tm, device="cuda")[None, :] < (ml // 2 * 2)[:, None]).unsqueeze(1)
    # (after three flows the one-ulp flips of 36 bf16-stored tensors have spread: both paths sit equally far from the fp32 oracle, whose bar for
    #  either is 0.1 in tests/test_gpu_decoder_fullwidth.py)
    assert ((zf - zu) * valid).abs().max() <= 0.2, ((zf - zu) * valid).abs().max()
    assert (((zf - zu) * valid) ** 2).mean().sqrt() <= 5e-3 * ((zu * valid) ** 2).mean().sqrt()
    assert ((ldf - ldu).abs() <= 2e-3 * ldu.abs() + 0.05).all(), (ldf, ldu)
    # tile seams: the error next to the 52-row tile boundaries is of the same size as elsewhere
    R = bf.x.shape[1]
    err = ((bf.x[1] - bu.x[1]).abs().max(dim=1).values * rm)
    seam = torch.zeros(R, dtype=torch.bool, device="cuda")
    for k in range(0, R, 52):
        seam[max(k - 2, 0):k + 2] = True
    if seam.any() and (~seam).any() and err[~seam].max() > 0:
        assert err[seam].max() <= 4 * err[~seam].max() + 1e-3, (err[seam].max().item(), err[~seam].max().item())


def test_fused_forward_with_speaker_conditioning_and_dropout():
    D, dc, W, mels, ml, cond = _setup(2, [640, 300, 64, 10], 640, spk_dim=256, seed=5)
    seed = torch.tensor([12345], device="cuda", dtype=torch.int32)
    for drop in (0.0, 0.3):
        zf, ldf, bf, rm, cf = _forward(D, dc, W, mels, ml, cond, True, drop, seed if drop else None)
        zu, ldu, bu, _, _ = _forward(D, dc, W, mels, ml, cond, False, drop, seed if drop else None)
        assert sum(n for k, n in cf.items() if k.startswith("wn_fwd<")) == 2, cf
        for l in range(4):      # same dropout masks: a mismatch would zero ~30 % of the gates in one path only
            _close(bf.gates[0, l], bu.gates[0, l], f"gates{l} drop={drop}", rm, atol=1e-6, frac=2e-2 * (l + 1), big=0.05)
        valid = (torch.arange(640, device="cuda")[None, :] < (ml // 2 * 2)[:, None]).unsqueeze(1)
        assert ((zf - zu) * valid).abs().max() <= 0.2


@pytest.mark.parametrize("lengths,tm", [([400, 222, 8], 400), ([2074] * 2, 2074)])
def test_fused_inverse_matches_per_conv_launches(lengths, tm):
    D, dc, W, mels, ml, _ = _setup(3, lengths, tm, seed=3)
    z = torch.randn(len(lengths), 80, tm, device="cuda") * 0.6
    outs = []
    for fused in (True, False):
        D.TUNE["fused_wn"] = fused
        try:
            with torch.no_grad():
                launch_reset()
                outs.append(D.decoder_inverse(dc, W, z, ml))
                torch.cuda.synchronize()
                c = launch_counts()
                assert (sum(n for k, n in c.items() if k.startswith("wn_fwd<")) == 3) == fused, c
        finally:
            D.TUNE["fused_wn"] = True
    valid = (torch.arange(outs[0].shape[2], device="cuda")[None, :] < (ml // 2 * 2)[:, None]).unsqueeze(1)
    d = ((outs[0] - outs[1]) * valid).abs()
    assert d.max() <= 0.2 and (d ** 2).mean().sqrt() <= 5e-3 * ((outs[1] * valid) ** 2).mean().sqrt(), d.max()
    # and the round trip through the fused forward returns the input
    with torch.no_grad():
        prep = D._Prepared(dc, W, need_bwd=False)
        back = D._run_forward(dc, prep, outs[0], ml)[0]
    assert ((back - z) * valid[:, :, :back.shape[2]]).abs().max() <= 0.1


def test_fused_inverse_with_conditioning_at_long_form_size():
    """Serving shape of the speaker-conditioned model: 12 flows, 32 utterances of 2074 frames.  rows x (F L 2H) x 4 bytes exceeds 2^31 - the
    per-UTTERANCE conditioning table is indexed by utterance, so the 32-bit buffer offsets only have to span B x (F L 2H) (a bound on the row
    count rejected this call with GLOWTTS_E_ARG until round 3).  Fused against per-conv launches on the same conditioning."""
    Bn, tm = 32, 2074
    D, dc, W, mels, ml, cond = _setup(12, [tm] * (Bn - 1) + [1500], tm, spk_dim=256, seed=11)
    assert Bn * (tm // 2 + 4) * cond[0].numel() * 4 >= 2 ** 31
    z = torch.randn(Bn, 80, tm, device="cuda") * 0.6
    outs = []
    for fused in (True, False):
        D.TUNE["fused_wn"] = fused
        try:
            with torch.no_grad():
                launch_reset()
                outs.append(D.decoder_inverse(dc, W, z, ml, cond=cond))
                torch.cuda.synchronize()
                c = launch_counts()
                assert (sum(n for k, n in c.items() if k.startswith("wn_fwd<nodrop,cond")) == 12) == fused, c
        finally:
            D.TUNE["fused_wn"] = True
    valid = (torch.arange(outs[0].shape[2], device="cuda")[None, :] < (ml // 2 * 2)[:, None]).unsqueeze(1)
    d = ((outs[0] - outs[1]) * valid).abs()
    assert torch.isfinite(outs[0]).all()
    assert (d ** 2).mean().sqrt() <= 2e-2 * ((outs[1] * valid) ** 2).mean().sqrt(), (d.max(), (d ** 2).mean().sqrt())


def test_exact_wait_counts_equal_conservative_waits():
    """The fused kernel waits for its weight slabs with exact vmcnt counts that step over its own outstanding stores (wavenet_fused.hip
    begin_step).  A count that is too large would read a slab before it has landed - a race that shows as a wrong value now and then.  With
    conservative waits the kernel is slower but cannot race: both must agree BIT FOR BIT, on several shapes and repeated launches."""
    from glow_tts_amd import _lib
    L = _lib.lib()
    for lengths, tm, drop in (([800] * 32, 800, 0.05), ([640, 522, 240, 2], 640, 0.0), ([800] * 8, 800, 0.3)):
        D, dc, W, mels, ml, _ = _setup(2, lengths, tm, seed=21)
        seed = torch.tensor([99], device="cuda", dtype=torch.int32)
        try:
            L.glowtts_wavenet_debug_safe_waits(1)
            ref = _forward(D, dc, W, mels, ml, None, True, drop, seed if drop else None)
        finally:
            L.glowtts_wavenet_debug_safe_waits(0)
        for _ in range(5):
            got = _forward(D, dc, W, mels, ml, None, True, drop, seed if drop else None)
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
            for name in ("hs", "gates", "actp", "skipb", "x", "xmid"):      # (skip: its fp32 rows are not kept by default - the bf16 copy is)
                assert torch.equal(getattr(got[2], name), getattr(ref[2], name)), name
            cm, cl = _outs_columns()
            assert torch.equal(got[2].outs[:, :, cm], ref[2].outs[:, :, cm]) and torch.equal(got[2].outs[:, :, cl], ref[2].outs[:, :, cl])


def _grads(D, dc, P, mels, ml, wz, wl, fused_bwd, drop_p=0.0, cond=None):
    default = D.TUNE["fused_wn_bwd"]
    D.TUNE["fused_wn_bwd"] = fused_bwd
    try:
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        x = mels.clone().requires_grad_(True)
        W = D.stack_decoder_weights(Pg, dc)
        c = cond.clone().requires_grad_(True) if cond is not None else None
        launch_reset()
        z, logdet, _ = D.DecoderFunction.apply(dc, x, ml, c, drop_p, None, None, None, *W)
        ((z * wz).sum() + (logdet * wl).sum()).backward()
        torch.cuda.synchronize()
        g = {k: p.grad for k, p in Pg.items() if p.grad is not None}
        if c is not None:
            g["<conditioning>"] = c.grad
        return g, x.grad, launch_counts()
    finally:
        D.TUNE["fused_wn_bwd"] = default


@pytest.mark.parametrize("lengths,tm,drop", [([640, 522, 240, 2], 640, 0.0), ([800] * 3, 800, 0.3), ([104], 104, 0.0), ([422, 36, 36, 800], 800, 0.05)])
def test_fused_backward_matches_per_conv_backward(lengths, tm, drop):
    """The fused data-gradient kernel (csrc/wavenet_fused_bwd.hip) against the per-conv backward on the SAME fused forward: every parameter
    gradient and d(mel).  Both round d skip, the gate gradients and d x_l to bf16 at the same places; a seam / halo bug, a wrong dropout mask or
    a wrong partner in the partial-sum exchange would show as O(1) errors."""
    from glow_tts_amd import decoder as D
    g = torch.Generator().manual_seed(7)
    cfg, sd = full_width_state(2, g)
    dc = D.DecoderConfig(cfg.mel_dim, cfg.n_flows, cfg.n_squeeze, cfg.n_split, cfg.wn_channels, cfg.wn_layers, cfg.wn_kernel, 1)
    P = {k: v.cuda() for k, v in sd.items()}
    B = len(lengths)
    mels = (torch.randn(B, 80, tm, generator=g) * 1.5).clamp(-4, 4).cuda()
    ml = torch.tensor(lengths).cuda()
    wz, wl = torch.randn(B, 80, tm, generator=g).cuda(), (torch.randn(B, generator=g) * 0.05).cuda()
    torch.manual_seed(3)
    gf, dxf, cf = _grads(D, dc, P, mels, ml, wz, wl, True, drop)
    torch.manual_seed(3)
    gu, dxu, cu = _grads(D, dc, P, mels, ml, wz, wl, False, drop)
    assert sum(n for k, n in cf.items() if k.startswith("wn_bwd<")) == 2 and not any(k.startswith("conv_dma<LINEAR,5") for k in cf), cf
    assert any("<LINEAR,5" in k for k in cu) and not any(k.startswith("wn_bwd") for k in cu), cu
    worst = (2.0, "")
    for k, want in gu.items():
        a, b = gf[k].flatten().double(), want.flatten().double()
        cos = (a @ b / (a.norm() * b.norm() + 1e-30)).item()
        ratio = (a.norm() / (b.norm() + 1e-30)).item()
        worst = min(worst, (cos, k))
        assert cos >= 0.9995 and 0.99 <= ratio <= 1.01, (k, cos, ratio)
    valid = (torch.arange(tm, device="cuda")[None, :] < (ml // 2 * 2)[:, None]).unsqueeze(1)
    a, b = (dxf * valid).flatten().double(), (dxu * valid).flatten().double()
    assert (a @ b / (a.norm() * b.norm())).item() >= 0.9995 and 0.99 <= (a.norm() / b.norm()).item() <= 1.01
    print("fused vs per-conv backward: worst gradient cosine", worst)


def test_backward_exact_wait_counts_equal_conservative_waits():
    """The fused BACKWARD kernel's hand-counted `vmcnt(2 + X)` waits (wavenet_fused_bwd.hip begin_step: they step over the wave's own prefetched gate loads and
    copy-out stores) against the same kernel compiled with the conservative `vmcnt(2)` everywhere (`glowtts_wavenet_debug_safe_waits`): every gradient bit for
    bit, several shapes, repeated launches - a count that is one too large would read a weight slab before it has landed."""
    from glow_tts_amd import _lib, decoder as D
    L = _lib.lib()
    g = torch.Generator().manual_seed(11)
    cfg, sd = full_width_state(2, g)
    dc = D.DecoderConfig(cfg.mel_dim, cfg.n_flows, cfg.n_squeeze, cfg.n_split, cfg.wn_channels, cfg.wn_layers, cfg.wn_kernel, 1)
    P = {k: v.cuda() for k, v in sd.items()}
    for lengths, tm, drop in (([800] * 32, 800, 0.05), ([640, 522, 240, 2], 640, 0.3), ([104], 104, 0.05)):
        B = len(lengths)
        mels = (torch.randn(B, 80, tm, generator=g) * 1.5).clamp(-4, 4).cuda()
        ml = torch.tensor(lengths).cuda()
        wz, wl = torch.randn(B, 80, tm, generator=g).cuda(), (torch.randn(B, generator=g) * 0.05).cuda()
        try:
            L.glowtts_wavenet_debug_safe_waits(1)
            torch.manual_seed(3)
            ref, dxr, cr = _grads(D, dc, P, mels, ml, wz, wl, True, drop)
        finally:
            L.glowtts_wavenet_debug_safe_waits(0)
        assert sum(n for k, n in cr.items() if k.startswith("wn_bwd<")) == 2, cr
        for _ in range(3):
            torch.manual_seed(3)
            got, dxg, _ = _grads(D, dc, P, mels, ml, wz, wl, True, drop)
            assert torch.equal(dxg, dxr)
            for k in ref:
                assert torch.equal(got[k], ref[k]), k


@pytest.mark.parametrize("lengths,tm,drop", [([640, 522, 240, 2], 640, 0.3), ([40, 36, 20, 8, 40, 40, 12], 40, 0.3), ([800] * 3, 800, 0.0)])
def test_fused_backward_conditioning_gradient(lengths, tm, drop):
    """Speaker / prosody conditioning (Modules.py:863-866) joins the gate pre-activation AFTER the dropout: its gradient is the per-utterance sum
    of the gate gradients BEFORE the keep mask.  The fused backward accumulates it per workgroup over the rows it OWNS (halo rows are another
    workgroup's), one run per utterance: windows inside one utterance (the fast path), straddling two (640 frames: 324 rows per utterance
    against 52-row windows), and holding several whole utterances (40 frames: 24 rows each) must all agree with the per-conv backward - in
    the conditioning gradient and, with it present, in every other gradient."""
    from glow_tts_amd import decoder as D
    g = torch.Generator().manual_seed(17)
    cfg, sd = full_width_state(2, g, spk_dim=256)
    dc = D.DecoderConfig(cfg.mel_dim, cfg.n_flows, cfg.n_squeeze, cfg.n_split, cfg.wn_channels, cfg.wn_layers, cfg.wn_kernel, 1)
    P = {k: v.cuda() for k, v in sd.items()}
    B = len(lengths)
    mels = (torch.randn(B, 80, tm, generator=g) * 1.5).clamp(-4, 4).cuda()
    ml = torch.tensor(lengths).cuda()
    spk = torch.randn(B, 256, generator=g)
    cond = D.conditioning(P, dc, speakers=(spk / spk.norm(dim=1, keepdim=True)).cuda()).detach()
    wz, wl = torch.randn(B, 80, tm, generator=g).cuda(), (torch.randn(B, generator=g) * 0.05).cuda()
    torch.manual_seed(3)
    gf, dxf, cf = _grads(D, dc, P, mels, ml, wz, wl, True, drop, cond)
    torch.manual_seed(3)
    gu, dxu, cu = _grads(D, dc, P, mels, ml, wz, wl, False, drop, cond)
    tag = "wn_bwd<drop,cond>" if drop else "wn_bwd<nodrop,cond>"
    assert cf.get(tag, 0) == 2 and not any(k.startswith("conv_dma<LINEAR,5") for k in cf), cf
    assert not any(k.startswith("wn_bwd") for k in cu), cu
    assert gf["<conditioning>"].shape == cond.shape and gu["<conditioning>"].abs().max() > 0
    for k, want in gu.items():
        a, b = gf[k].flatten().double(), want.flatten().double()
        cos = (a @ b / (a.norm() * b.norm() + 1e-30)).item()
        ratio = (a.norm() / (b.norm() + 1e-30)).item()
        assert cos >= 0.9995 and 0.99 <= ratio <= 1.01, (k, cos, ratio)
    # per utterance as well (a run credited to the wrong utterance keeps the total)
    for b_ in range(B):
        a, b = gf["<conditioning>"][b_].flatten().double(), gu["<conditioning>"][b_].flatten().double()
        if b.norm() > 0:
            assert (a @ b / (a.norm() * b.norm() + 1e-30)).item() >= 0.999, (b_, (a @ b / (a.norm() * b.norm() + 1e-30)).item())


def test_balanced_weight_gradient_launch_equals_the_single_round_form():
    """decoder.TUNE["wgrad_balance"] (round 6; off by default - it shortens the decoder's tail, not the step): at B = 32 x 800 frames the In_l weight-gradient group is 48 problems x 6 one-per-CU tiles = 288 tiles on 256 CUs -
    two rounds, the second almost empty.  The balanced form keeps 42 problems whole and cuts 6 into 8 row splits over 4 utterances each (short tiles beside and
    behind the whole ones, partial images summed in a fixed order).  Same gradients up to fp32 summation order, for every class; repeated runs bit-identical."""
    from glow_tts_amd import decoder as D
    g = torch.Generator().manual_seed(29)
    cfg, sd = full_width_state(12, g)
    dc = D.DecoderConfig(cfg.mel_dim, cfg.n_flows, cfg.n_squeeze, cfg.n_split, cfg.wn_channels, cfg.wn_layers, cfg.wn_kernel, 1)
    P = {k: v.cuda() for k, v in sd.items()}
    B, tm = 32, 800
    mels = (torch.randn(B, 80, tm, generator=g) * 1.5).clamp(-4, 4).cuda()
    ml = torch.tensor([800] * 20 + [640, 522, 240, 2, 798, 400, 96, 12, 800, 700, 600, 500]).cuda()
    wz, wl = torch.randn(B, 80, tm, generator=g).cuda(), (torch.randn(B, generator=g) * 0.05).cuda()
    res = []
    for bal in (True, False, True):
        D.TUNE["wgrad_balance"] = bal
        try:
            torch.manual_seed(3)
            grads, dx, counts = _grads(D, dc, P, mels, ml, wz, wl, -1, 0.05)
        finally:
            D.TUNE["wgrad_balance"] = False
        assert counts.get("wgrad_dma<5>/grouped") == 1, counts
        res.append((grads, dx))
    (ga, dxa), (gb, dxb), (gc, dxc) = res
    assert torch.equal(dxa, dxb) and torch.equal(dxa, dxc)
    n_diff = 0
    for k in gb:
        assert torch.equal(ga[k], gc[k]), k                                     # deterministic
        if not torch.equal(ga[k], gb[k]):
            n_diff += 1
            assert "In_" in k, k                                                # only the In_l classes are cut
            scale = gb[k].abs().max()
            assert (ga[k] - gb[k]).abs().max() <= 2e-5 * scale, (k, ((ga[k] - gb[k]).abs().max() / scale).item())
    assert n_diff > 0, "the balanced launch did not engage at B = 32 x 800"


@pytest.mark.parametrize("fused", [True, False])
def test_conditioning_gradient_propagates_non_finite_values(fused):
    """The conditioning gradient is summed in 64-bit fixed point (integer atomics: reproducible).  A NaN / Inf gate gradient must not vanish in the float ->
    integer conversion: it poisons its accumulator and `glowtts_fx_to_float` reads NaN - for the utterance it belongs to, and only that one."""
    from glow_tts_amd import decoder as D
    g = torch.Generator().manual_seed(19)
    cfg, sd = full_width_state(2, g, spk_dim=256)
    dc = D.DecoderConfig(cfg.mel_dim, cfg.n_flows, cfg.n_squeeze, cfg.n_split, cfg.wn_channels, cfg.wn_layers, cfg.wn_kernel, 1)
    P = {k: v.cuda() for k, v in sd.items()}
    lengths, tm = [640, 522, 240], 640
    B = len(lengths)
    mels = (torch.randn(B, 80, tm, generator=g) * 1.5).clamp(-4, 4).cuda()
    ml = torch.tensor(lengths).cuda()
    spk = torch.randn(B, 256, generator=g)
    cond = D.conditioning(P, dc, speakers=(spk / spk.norm(dim=1, keepdim=True)).cuda()).detach()
    wz, wl = torch.randn(B, 80, tm, generator=g).cuda(), (torch.randn(B, generator=g) * 0.05).cuda()
    wz[1, 7, 100] = float("nan")                               # d loss / d z of one frame of utterance 1
    torch.manual_seed(3)
    grads, _, _ = _grads(D, dc, P, mels, ml, wz, wl, fused, 0.05, cond)
    dc_ = grads["<conditioning>"]
    assert torch.isnan(dc_[1]).any(), "the non-finite gate gradient vanished from the conditioning gradient"
    assert torch.isfinite(dc_[0]).all() and torch.isfinite(dc_[2]).all()


def test_mixed_forward_first_flow_per_conv():
    """decoder.TUNE["fused_wn_fwd_skip"] = 1: flow 0 of the training forward on the per-conv launches, reading the Start / In_l / last
    Res_Skip / End slabs of the FUSED weight image as they are (only the PAIR-packed Res_Skip_l is packed once more), flow 1 fused.  Same
    gradients as the all-fused forward."""
    from glow_tts_amd import decoder as D
    g = torch.Generator().manual_seed(23)
    cfg, sd = full_width_state(2, g)
    dc = D.DecoderConfig(cfg.mel_dim, cfg.n_flows, cfg.n_squeeze, cfg.n_split, cfg.wn_channels, cfg.wn_layers, cfg.wn_kernel, 1)
    P = {k: v.cuda() for k, v in sd.items()}
    lengths, tm = [640, 418, 96], 640
    B = len(lengths)
    mels = (torch.randn(B, 80, tm, generator=g) * 1.5).clamp(-4, 4).cuda()
    ml = torch.tensor(lengths).cuda()
    wz, wl = torch.randn(B, 80, tm, generator=g).cuda(), (torch.randn(B, generator=g) * 0.05).cuda()
    res = []
    default = D.TUNE["fused_wn_fwd_skip"]
    for skip in (0, 1):
        D.TUNE["fused_wn_fwd_skip"] = skip
        try:
            torch.manual_seed(3)
            res.append(_grads(D, dc, P, mels, ml, wz, wl, 0, 0.05))
        finally:
            D.TUNE["fused_wn_fwd_skip"] = default
    (g0, dx0, c0), (g1, dx1, c1) = res
    assert sum(n for k, n in c0.items() if k.startswith("wn_fwd<")) == 2 and sum(n for k, n in c1.items() if k.startswith("wn_fwd<")) == 1, (c0, c1)
    assert c1.get("conv_dma<GATE,5>", 0) == 4 and c0.get("conv_dma<GATE,5>", 0) == 0, c1
    for k, want in g0.items():
        a, b = g1[k].flatten().double(), want.flatten().double()
        cos = (a @ b / (a.norm() * b.norm() + 1e-30)).item()
        assert cos >= 0.9995 and 0.99 <= (a.norm() / (b.norm() + 1e-30)).item() <= 1.01, (k, cos)
