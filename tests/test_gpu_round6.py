"""GPU tests of round 6's additions (each against a plain fp64 / fp32 torch restatement of the reference op it replaces)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cond_ref(kinds):
    """sum over kinds of vec @ (g v / ||v||)^T + bias  (Modules.py:832-845 weight_norm'ed Conv1d(D -> 2H, k = 1), :863-866), fp64."""
    out = None
    for g, v, b, vec in kinds:
        g, v, b, vec = g.double(), v.double(), b.double(), vec.double()
        w = g.view(-1, 1) * v.view(g.numel(), -1) / v.view(g.numel(), -1).norm(dim=1, keepdim=True)
        c = vec @ w.t() + b.view(1, -1)
        out = c if out is None else out + c
    return out


@pytest.mark.parametrize("D,B,supported", [(256, 32, True), (512, 40, True), (512, 41, False), (512, 64, False), (384, 64, True)])
def test_cond_linear_supported_matches_the_launch(D, B, supported):
    """glowtts_cond_linear_supported (ABI 6) is the kernels' own acceptance test: where it says yes, CondLinear runs and matches the fp64 product
    (forward and all four gradients); where it says no, glowtts_cond_linear_fwd / _bwd really reject the shape (decoder.py's guard was
    wider than the kernels' LDS budget, D = 512 with B > 40 raised in the forward)."""
    from glow_tts_amd import _lib, decoder
    L = decoder._L()
    N = 2 * 2 * 384
    assert bool(L.glowtts_cond_linear_supported(N, D, B)) == supported
    g_ = torch.Generator().manual_seed(D + B)
    g = (torch.rand(N, 1, 1, generator=g_) + 0.5).cuda().requires_grad_()
    v = torch.randn(N, D, 1, generator=g_).cuda().requires_grad_()
    b = torch.randn(N, generator=g_).cuda().requires_grad_()
    vec = torch.randn(B, D, generator=g_).cuda().requires_grad_()
    if not supported:
        with pytest.raises(_lib.GlowTTSHipError):
            decoder.CondLinear.apply(g, v, b, vec)
        return
    out = decoder.CondLinear.apply(g, v, b, vec)
    dout = torch.randn(B, N, generator=g_).cuda()
    out.backward(dout)
    ref_in = [t.detach().double().requires_grad_() for t in (g, v, b, vec)]
    ref = _cond_ref([ref_in])
    ref.backward(dout.double())
    assert (out.double() - ref).abs().max() <= 1e-4 * ref.abs().max()
    for got, want in zip((g, v, b, vec), ref_in):
        assert (got.grad.double() - want.grad).abs().max() <= 2e-4 * want.grad.abs().max(), got.shape


def test_wide_conditioning_vectors_take_the_matmul_path():
    """DecoderStacks.conditioning with 512-wide speaker vectors at B = 64 - a shape the conditioning kernels do not hold in LDS - must fall back to weight
    norm + matmul (and give the same numbers as the kernel gives at B = 32), not raise."""
    import copy
    from glow_tts_amd import hparams
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS
    d = copy.deepcopy(hparams.load_yaml(hparams.DEFAULT_YAML))
    d["Mode"] = "SE"
    d["Speaker_Embedding"]["Type"] = "GE2E"
    d["Speaker_Embedding"]["Embedding_Size"] = 512
    d["Decoder"]["Stack"] = 2
    d["Encoder"]["Transformer"]["Stacks"] = 1
    torch.manual_seed(3)
    model = GlowTTS(Recursive_Parse(d)).cuda()
    stacks = model._stacks(model._params())
    vec = torch.randn(64, 512, device="cuda")
    wide = stacks.conditioning(vec, None)                    # B = 64: matmul path
    half = stacks.conditioning(vec[:32].contiguous(), None)  # B = 32: the HIP kernel
    assert wide.shape == (64, 2, 4, 384)
    assert (wide[:32] - half).abs().max() <= 1e-4 * half.abs().max()
    wide.sum().backward()
    assert all(p.grad is not None for k, p in model.named_parameters() if "Speaker_" in k)


def _prior_case(B=3, C=80, Tx=37, Ty=96, seed=0):
    g = torch.Generator().manual_seed(seed)
    tl = torch.tensor([Tx, 20, 5][:B])
    ml = torch.tensor([Ty, 64, 80][:B])
    idx = torch.full((B, Ty), -1, dtype=torch.int32)
    for b in range(B):                                    # a monotonic alignment with skewed runs (one token owns half the frames)
        cuts = torch.sort(torch.randint(1, int(ml[b]), (int(tl[b]) - 1,), generator=g)).values if tl[b] > 1 else torch.zeros(0, dtype=torch.long)
        x = torch.zeros(int(ml[b]), dtype=torch.int32)
        for c in cuts.tolist():
            x[c:] += 1
        idx[b, :int(ml[b])] = x
    mean, ls = torch.randn(B, C, Tx, generator=g), torch.randn(B, C, Tx, generator=g) * 0.3
    mmask = (torch.arange(Ty)[None] < ml[:, None]).float()
    z = torch.randn(B, C, Ty, generator=g) * mmask[:, None]
    ld = torch.randn(B, generator=g)
    return tl, ml, idx, mean, ls, z, ld


def test_expand_pair_and_prior_loss_equal_the_separate_launches():
    """Round 6 fusions of Modules.py:120-122 and of the MLE backward through them: `ExpandPair` (one launch) == two `ExpandPrior` + `duration_targets`, and
    `PriorLoss` (one backward launch: d z + token-space segment sums) == `MLELoss` behind `ExpandPrior` - bit for bit, values and all four gradients; and
    `MLE_Loss` takes the fused path exactly when it is handed the tagged tensors."""
    from glow_tts_amd import alignment as A
    from helpers import launch_counts, launch_reset
    tl, ml, idx, mean, ls, z, ld = _prior_case()
    dev = lambda t: t.cuda()
    res = []
    for fused in (False, True):
        m, l, zz, dd = (dev(t).requires_grad_(True) for t in (mean, ls, z, ld))
        if fused:
            mm, ms, tg, path = A.ExpandPair.apply(m, l, dev(idx), dev(tl), None, True)
            from glow_tts_amd.monotonic_align import path_from_idx
            assert torch.equal(path, path_from_idx(dev(idx), mean.shape[2], torch.float32)) and not path.requires_grad       # Modules.py:116 in the same launch
            A.tag_prior(mm, ms, m, l, dev(idx))
            loss = A.mle_loss(zz, mm, ms, dd, dev(ml), 2, 80)
            assert type(loss.grad_fn).__name__.startswith("PriorLoss")
        else:
            mm, ms = A.ExpandPrior.apply(m, dev(idx)), A.ExpandPrior.apply(l, dev(idx))
            tg = A.duration_targets(dev(idx), dev(tl), mean.shape[2]).squeeze(1)
            loss = A.mle_loss(zz, mm, ms, dd, dev(ml), 2, 80)
            assert type(loss.grad_fn).__name__.startswith("MLELoss")
        (loss * 1.7).backward()
        res.append([t.detach().cpu() for t in (mm, ms, tg, loss, m.grad, l.grad, zz.grad, dd.grad)])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # the tagged outputs used anywhere else still get the general backward
    m, l = dev(mean).requires_grad_(True), dev(ls).requires_grad_(True)
    mm, ms, _, none = A.ExpandPair.apply(m, l, dev(idx), dev(tl), None)
    assert none is None
    (mm.sum() + 2 * ms.sum()).backward()
    cnt = (idx[:, None, :] == torch.arange(mean.shape[2])[None, :, None]).sum(-1).float()
    assert torch.allclose(m.grad.cpu(), cnt[:, None, :].expand_as(mean)) and torch.allclose(l.grad.cpu(), 2 * cnt[:, None, :].expand_as(mean))


def test_seeded_losses_write_their_gradients_in_the_forward_launch():
    """`LossTerms.backward` seeds the loss terms with `_lib.one` (no sum node, no ones_like fill); `PriorLoss` and `DurationMSE`, told that seed, write their
    gradients in the forward launch (glowtts_prior_loss: reduction + gradients + last-workgroup finalisation in one launch) and launch NOTHING in the backward -
    same bits as `(mle + length).backward()` on the unseeded nodes; any other seed takes the backward launches; a second module's counter is its own."""
    from glow_tts_amd import _lib, alignment as A
    from helpers import launch_counts, launch_reset
    tl, ml, idx, mean, ls, z, ld = _prior_case()
    dev = lambda t: t.cuda()
    g = torch.Generator().manual_seed(9)
    a0, t0 = torch.randn(mean.shape[0], 1, mean.shape[2], generator=g), torch.randn(mean.shape[0], 1, mean.shape[2], generator=g)

    def run(mode):
        m, l, zz, dd, a = (dev(t).requires_grad_(True) for t in (mean, ls, z, ld, a0))
        A.FUSED["seeded"] = mode != "plain"
        try:
            mm, ms, tg, _ = A.ExpandPair.apply(m, l, dev(idx), dev(tl), None)
            A.tag_prior(mm, ms, m, l, dev(idx))
            owner = {}
            mle = A.mle_loss(zz, mm, ms, dd, dev(ml), 2, 80, owner=owner)
            length = A.duration_mse(a, dev(t0))
        finally:
            A.FUSED["seeded"] = True
        launch_reset()
        if mode == "plain":
            (mle + length).backward()
        elif mode == "seeded":
            A.LossTerms([mle, length]).backward()
        else:                                                  # seeded nodes, foreign seed: the general backward launches
            (mle + length).backward()
        counts = launch_counts()
        if mode == "foreign":
            assert counts.get("prior_loss_bwd") == 1 and counts.get("mse_loss_bwd") == 1, counts
        if mode == "seeded":
            assert not any(n and k.startswith(("prior", "mse", "mle")) for k, n in counts.items()), counts
            assert int(owner[next(iter(owner))].item()) == 0          # the completion counter is left zero
        return [t.detach().cpu() for t in (mle, length, m.grad, l.grad, zz.grad, dd.grad, a.grad)]
    plain, seeded, foreign = run("plain"), run("seeded"), run("foreign")
    for x, y, w in zip(plain, seeded, foreign):
        assert torch.equal(x, y) and torch.equal(x, w)
    # weighted terms (data parallel): seeds are device scalars the forward launches read
    wm, wr = torch.tensor(0.37).cuda(), torch.tensor(0.5).cuda()
    m, l, zz, dd, a = (dev(t).requires_grad_(True) for t in (mean, ls, z, ld, a0))
    A.SEEDS["mle"], A.SEEDS["rest"] = wm, wr
    try:
        mm, ms, tg, _ = A.ExpandPair.apply(m, l, dev(idx), dev(tl), None)
        A.tag_prior(mm, ms, m, l, dev(idx))
        mle, length = A.mle_loss(zz, mm, ms, dd, dev(ml), 2, 80), A.duration_mse(a, dev(t0))
    finally:
        A.SEEDS["mle"] = A.SEEDS["rest"] = None
    terms = A.LossTerms([mle, length], [wm, wr])
    terms.backward()
    got = [t.grad.detach().cpu() for t in (m, l, zz, dd, a)]
    m2, l2, zz2, dd2, a2 = (dev(t).requires_grad_(True) for t in (mean, ls, z, ld, a0))
    mm2, ms2 = A.ExpandPrior.apply(m2, dev(idx)), A.ExpandPrior.apply(l2, dev(idx))
    tot = A.mle_loss(zz2, mm2, ms2, dd2, dev(ml), 2, 80) * wm + A.duration_mse(a2, dev(t0)) * wr
    tot.backward()
    for x, y in zip(got, [t.grad.detach().cpu() for t in (m2, l2, zz2, dd2, a2)]):
        assert torch.allclose(x, y, rtol=1e-6, atol=1e-12)
    assert abs(terms.item() - tot.item()) <= 1e-6 * abs(tot.item())


@pytest.mark.parametrize("B,T,C", [(32, 120, 256), (3, 7, 448), (1, 1, 4), (5, 33, 1024)])
def test_duration_projection_and_prior_split_match_torch(B, T, C):
    """csrc/dur_ops.hip: Duration_Predictor's Projection (Modules.py:596-618) and the split of the encoder's projected rows into mean / log_std
    (Modules.py:283-286), one launch per direction each, against the torch expressions they replace (fp64)."""
    from glow_tts_amd.encoder import ROW_PAD, DurProj, PriorSplit, from_rows
    g = torch.Generator().manual_seed(B * 1000 + T)
    Tp = T + 2 * ROW_PAD
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = T
    mask = (torch.arange(T)[None] < lens[:, None]).float().unsqueeze(1)
    d = torch.randn(B, Tp, C, generator=g)
    d[:, :ROW_PAD] = 0
    d[:, -ROW_PAD:] = 0
    w, b = torch.randn(1, C, 1, generator=g) * 0.1, torch.randn(1, generator=g)
    gout = torch.randn(B, 1, T, generator=g)
    dc, wc, bc = (t.cuda().requires_grad_(True) for t in (d.reshape(B * Tp, C), w, b))
    owner = {}
    out = DurProj.apply(dc, wc, bc, mask.cuda(), owner)
    out.backward(gout.cuda())
    d64, w64, b64 = (t.double().requires_grad_(True) for t in (d.reshape(B * Tp, C), w, b))
    want = ((d64 * w64[0, :, 0]).sum(-1) + b64).view(B, Tp)[:, ROW_PAD:-ROW_PAD].unsqueeze(1) * mask.double()
    want.backward(gout.double())
    assert torch.allclose(out.detach().cpu().double(), want.detach(), rtol=1e-5, atol=1e-5)
    for got, ref in ((dc.grad, d64.grad), (wc.grad, w64.grad), (bc.grad, b64.grad)):
        assert torch.allclose(got.cpu().double(), ref, rtol=1e-5, atol=1e-5 * max(1.0, float(ref.abs().max()))), (got.cpu().double() - ref).abs().max()
    assert int(owner[next(iter(owner))].item()) == 0
    # a second backward through the same counter (left zero by the first)
    dc.grad = wc.grad = bc.grad = None
    DurProj.apply(dc, wc, bc, mask.cuda(), owner).backward(gout.cuda())
    assert torch.allclose(wc.grad.cpu().double(), w64.grad, rtol=1e-5, atol=1e-5 * max(1.0, float(w64.grad.abs().max())))
    # the split
    M = 8 if C % 16 else min(C // 2, 80)
    rows = torch.randn(B * Tp, 2 * M, generator=g)
    rc = rows.cuda().requires_grad_(True)
    mean, ls = PriorSplit.apply(rc, B, T)
    ref = from_rows(rows.view(B, Tp, -1))
    assert torch.equal(mean.cpu(), ref[:, :M].contiguous()) and torch.equal(ls.cpu(), ref[:, M:].contiguous())
    gm, gl = torch.randn(B, M, T, generator=g), torch.randn(B, M, T, generator=g)
    torch.autograd.backward([mean, ls], [gm.cuda(), gl.cuda()])
    r2 = rows.clone().requires_grad_(True)
    p2 = from_rows(r2.view(B, Tp, -1))
    torch.autograd.backward([p2[:, :M].contiguous(), p2[:, M:].contiguous()], [gm, gl])
    assert torch.equal(rc.grad.cpu(), r2.grad)
    rc.grad = None
    mean, ls = PriorSplit.apply(rc, B, T)
    ls.sum().backward()                                        # one of the two unused: zeros in its half
    assert torch.equal(rc.grad.cpu().view(B, Tp, -1)[:, ROW_PAD:-ROW_PAD, :M], torch.zeros(B, T, M)) and float(rc.grad.sum()) == B * T * M


def test_last_workgroup_hand_over_under_load():
    """The one-launch loss and the duration projection's backward hand their partial sums to the last workgroup through returning device-scope atomics (no release
    fence: that writes the XCD's L2 back).  `tools/stress_handover.py`: 200 launches beside a stream that keeps the CUs and the L2s busy, every result bit-identical
    to the first and to the unfused launches, the completion counters left at zero."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_handover.py"), "200"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_duration_mse_matches_torch():
    """`alignment.duration_mse` (one launch per direction) against torch's MSELoss (Train.py:203-211) and against the trainer's extent-normalised form."""
    from glow_tts_amd.alignment import duration_mse
    from glow_tts_amd.trainer import duration_loss
    g = torch.Generator().manual_seed(4)
    a = torch.randn(5, 1, 64, generator=g).cuda().requires_grad_(True)
    t = torch.randn(5, 1, 64, generator=g).cuda()
    tl = torch.tensor([64, 31, 9, 50, 12]).cuda()
    ref_a = a.detach().clone().requires_grad_(True)
    want = torch.nn.functional.mse_loss(ref_a, t)
    (want * 3).backward()
    got = duration_mse(a, t)
    (got * 3).backward()
    assert abs(got.item() - want.item()) <= 1e-6 * abs(want.item()) and torch.allclose(a.grad, ref_a.grad, rtol=1e-6, atol=1e-9)
    for ext in (None, torch.tensor(70.0).cuda()):
        a.grad = None
        got = duration_loss(a, t, tl, ext)
        got.backward()
        den = 5 * (64.0 if ext is None else 70.0)
        want = ((a.detach() - t) ** 2).sum() / den
        assert abs(got.item() - want.item()) <= 1e-6 * abs(want.item())
        assert torch.allclose(a.grad, 2 * (a.detach() - t) / den, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("B,Tp,lens", [(5, 13, [800, 64, 1, 513, 130]), (32, 13, None), (2, 4, [256, 200])])
def test_gst_tail_matches_the_torch_modules(B, Tp, lens):
    """The style-token tail (Modules.py:371-385: last valid GRU state -> 4-head attention over tanh(gst_Tokens) -> projection) as HIP launches (prosody._GSTTail,
    csrc/gst_ops.hip) against the same module run through torch ops in fp64: output, d(GRU states) and all nine parameter gradients."""
    import copy
    from glow_tts_amd import hparams
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.prosody import Prosody_Encoder
    from helpers import launch_counts, launch_reset
    torch.manual_seed(B + Tp)
    hp = Recursive_Parse(copy.deepcopy(hparams.load_yaml(hparams.DEFAULT_YAML)))
    pe = Prosody_Encoder(hp).cuda()
    att = pe.layer_Dict["Attention"]
    g = torch.Generator().manual_seed(7)
    lengths = torch.tensor(lens if lens is not None else torch.randint(1, Tp * 64 + 1, (B,), generator=g).tolist())
    hs = torch.randn(B, Tp, 128, generator=g).cuda().requires_grad_(True)
    dout = torch.randn(B, 256, generator=g).cuda()

    def tail(x, lengths_, mod, tokens):                     # prosody.Prosody_Encoder.forward's torch branch
        import math
        idx = (torch.ceil(lengths_ / 64.0).long() - 1).clamp_min(0)
        h = x[torch.arange(x.size(0), device=x.device), idx]
        keys = torch.tanh(tokens).unsqueeze(0).expand(x.size(0), -1, -1)
        return mod(h.unsqueeze(2), keys).squeeze(2)

    from glow_tts_amd.prosody import _GSTTail
    launch_reset()
    sq = lambda w: w.squeeze(-1)
    q_, k_, v_, p_ = (att.layer_Dict[n] for n in ("Query", "Key", "Value", "Projection"))
    out = _GSTTail.apply(hs, lengths.cuda(), 64, att.heads, pe.gst_Tokens, sq(q_.weight), q_.bias, sq(k_.weight), k_.bias, sq(v_.weight), v_.bias, sq(p_.weight), p_.bias)
    out.backward(dout)
    torch.cuda.synchronize()
    counts = launch_counts()
    assert counts.get("gst_fwd", 0) == 1 and counts.get("gst_bwd", 0) == 1, counts
    params = [pe.gst_Tokens] + [t for m in (q_, k_, v_, p_) for t in (m.weight, m.bias)]
    got = [out.detach().double().cpu(), hs.grad.double().cpu()] + [t.grad.double().cpu() for t in params]
    ref_att = copy.deepcopy(att).double().cpu()
    tok64 = pe.gst_Tokens.detach().double().cpu().requires_grad_(True)
    hs64 = hs.detach().double().cpu().requires_grad_(True)
    ref = tail(hs64, lengths.double(), ref_att, tok64)
    ref.backward(dout.double().cpu())
    rq, rk, rv, rp = (ref_att.layer_Dict[n] for n in ("Query", "Key", "Value", "Projection"))
    want = [ref.detach(), hs64.grad, tok64.grad] + [t.grad for m in (rq, rk, rv, rp) for t in (m.weight, m.bias)]
    names = ["out", "dhs", "dtokens"] + [f"d{n}.{w}" for n in ("Query", "Key", "Value", "Projection") for w in ("weight", "bias")]
    for name, a, b_ in zip(names, got, want):
        assert a.shape == b_.shape, (name, a.shape, b_.shape)
        if name == "dKey.bias":                             # softmax ignores a key bias: the true gradient is 0, both sides hold rounding noise
            assert a.abs().max() <= 1e-4 * max(1.0, want[0].abs().max().item())
            continue
        err = (a - b_).abs().max().item() / max(1e-6, b_.abs().max().item())
        assert err < 2e-5, (name, err)
