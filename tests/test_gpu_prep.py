"""Round 4: the one-launch weight preparation (csrc/prep_ops.hip: weight norm + every weight image of the decoder from the (weight_g, weight_v)
pairs) against the two-step path it replaces (glowtts_weightnorm_fwd, then glowtts_wavenet_pack_images / glowtts_pack_weight_batched): the images
must be BYTE-identical and the training step's outputs and gradients bit-identical (Modules.py:766, 818, 825 weight_norm; reference autograd)."""
import pytest
import torch

from helpers import full_width_state, launch_counts, launch_reset

pytestmark = pytest.mark.gpu


def _stacks(n_flows, seed):
    from glow_tts_amd import decoder as D
    g = torch.Generator().manual_seed(seed)
    cfg, sd = full_width_state(n_flows, g)
    dc = D.DecoderConfig(cfg.mel_dim, n_flows, cfg.n_squeeze, cfg.n_split, cfg.wn_channels, cfg.wn_layers, cfg.wn_kernel, 1)
    P = {k: v.cuda().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    return D, dc, D.DecoderStacks(P, dc), P, g


@pytest.mark.parametrize("nskip,nfb,late", [(1, 2, True), (0, 3, True), (2, 0, True), (1, 1, True), (1, 2, False), (0, 3, False)])
def test_prep_images_byte_identical(nskip, nfb, late):
    """late: the backward-only images as a second launch (TUNE["prep_bwd_late"], round 5's default) or every image in the one launch at the head (round 6's)."""
    D, dc, st, P, _ = _stacks(3, 5)
    old = dict(D.TUNE)
    D.TUNE.update(fused_wn_fwd_skip=nskip, fused_wn_bwd=nfb, prep_bwd_late=late)
    try:
        with torch.no_grad():
            W = dict(zip(D.WEIGHT_KEYS, [w.contiguous() for w in st.weights()]))
            ref = D._Prepared(dc, W, need_bwd=True, rows=4 * 404)
            A = dict(zip(D.WEIGHT_KEYS_GV, [w.contiguous() for w in st.weights(gv=True)]))
            GV = {k: (A.pop("g" + k[1:]), A.pop("v" + k[1:])) for k in D.WN_KEYS}
            launch_reset()
            got = D._Prepared(dc, A, need_bwd=True, rows=4 * 404, GV=GV)
            torch.cuda.synchronize()
            assert launch_counts().get("prep_weights", 0) == 1          # what the forward reads ...
            got.launch_bwd_images()                                     # ... and (round 5: a second launch, issued off the decoder's chain) what only the backward reads
            got.launch_bwd_images()                                     # (idempotent)
            torch.cuda.synchronize()
            lc = launch_counts()
        assert lc.get("prep_weights", 0) == (2 if late else 1) and not any(k.startswith("pack") or k.startswith("weightnorm") for k in lc), lc
        # (the Start conv's three K chunks fill 1.5 of its two slabs; the other half slab is never written nor read)
        hole = slice(36864, 49152)
        a, b = got.wn_img.clone(), ref.wn_img.clone()
        a[:, hole] = 0
        b[:, hole] = 0
        assert torch.equal(a, b)
        assert (got.wn_img_t is None) == (ref.wn_img_t is None)
        if ref.wn_img_t is not None:
            assert torch.equal(got.wn_img_t, ref.wn_img_t)
        for k, pb in ref.pk.items():
            if isinstance(pb, D.PackedBatch):
                assert torch.equal(got.pk[k].data, pb.data), k
        if ref.pk_conv is not None:
            assert torch.equal(got.pk_conv["rs"].data, ref.pk_conv["rs"].data)
        # 1 / ||v|| as the weight-norm forward returns it
        for k, (gk, vk) in GV.items():
            rows, cols = vk.numel() // (vk.shape[-1] * vk.shape[-2]), vk.shape[-1] * vk.shape[-2]
            w, inv = torch.empty_like(vk), torch.empty(rows, device="cuda")
            D._L().glowtts_weightnorm_fwd(vk.data_ptr(), gk.data_ptr(), w.data_ptr(), inv.data_ptr(), rows, cols, None)
            torch.cuda.synchronize()
            assert torch.equal(got.inv[k].reshape(-1), inv), k
    finally:
        D.TUNE.update(old)


@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_decoder_function_gv_form_is_bit_identical(drop):
    """DecoderFunction fed (g, v) stacks == DecoderFunction fed WeightNorm.apply(g, v): z, log-determinants and every leaf gradient."""
    D, dc, st, P, g = _stacks(3, 11)
    mels = torch.randn(4, 80, 200, generator=g).cuda()
    ml = torch.tensor([200, 164, 96, 2]).cuda()
    wz = torch.randn(4, 80, 200, generator=g).cuda()
    res = []
    for gv in (False, True):
        for p in P.values():
            p.grad = None
        torch.manual_seed(3)                                   # (the dropout seed word is drawn from torch's generator)
        W = st.weights(gv=gv)
        z, ld, _ = D.DecoderFunction.apply(dc, mels, ml, None, drop, None, None, None, *W)
        ((z * wz).sum() + ld.sum()).backward()
        torch.cuda.synchronize()
        res.append((z.detach().clone(), ld.detach().clone(), {k: p.grad.clone() for k, p in P.items() if p.grad is not None}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert res[0][2].keys() == res[1][2].keys() and len(res[0][2]) == 96
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k


@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_next_flows_actnorm_in_the_coupling_epilogue_is_bit_identical(drop):
    """Round 4 (ABI 4, glowtts_flow_acts.next_*): a flow on the fused coupling launch applies the NEXT flow's ActNorm + invertible 1x1 conv in that
    launch's epilogue (Modules.py:693-694, 738-756) instead of a launch of its own: z, log-determinants and every gradient - i.e. every kept
    activation the backward reads (xmid, the x_a half of each flow's output, the bf16 x_a rows) - equal the separate pass bit for bit, ragged lengths
    and a two-frame utterance included."""
    D, dc, st, P, g = _stacks(4, 23)
    mels = torch.randn(4, 80, 200, generator=g).cuda()
    ml = torch.tensor([200, 164, 96, 2]).cuda()
    wz = torch.randn(4, 80, 200, generator=g).cuda()
    res, counts = [], []
    old = D.TUNE["chain_actnorm"]
    try:
        for chain in (False, True):
            D.TUNE["chain_actnorm"] = chain
            for p in P.values():
                p.grad = None
            torch.manual_seed(3)
            launch_reset()
            z, ld, _ = D.DecoderFunction.apply(dc, mels, ml, None, drop, None, None, None, *st.weights(gv=True))
            ((z * wz).sum() + ld.sum()).backward()
            torch.cuda.synchronize()
            counts.append(launch_counts())
            res.append((z.detach().clone(), ld.detach().clone(), {k: p.grad.clone() for k, p in P.items() if p.grad is not None}))
    finally:
        D.TUNE["chain_actnorm"] = old
    n_an = lambda lc: sum(n for k, n in lc.items() if k.startswith("actnorm_inv1x1") and "bwd" not in k)
    assert n_an(counts[0]) > n_an(counts[1]) >= 1, (counts[0], counts[1])
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert res[0][2].keys() == res[1][2].keys()
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k
