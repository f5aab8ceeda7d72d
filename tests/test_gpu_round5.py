"""GPU: the small kernels and switches round 5 added, each against a plain PyTorch reference or against the path it replaces.
  glowtts_token_masks           == Mask_Generate (Modules.py:206-211) + the rows layout's zero pad rows
  glowtts_sum_slices / _seg     == torch.sum over the slices (fixed order: bit-identical to a left-to-right sum)
  GRU recurrence, rows in regs  == torch.nn.GRU (Modules.py:338-343, 371), forward and every gradient
  GLOWTTS_F_GATE_IN0            == data-gradient conv followed by the separate relu / dropout gate pass, to bf16 rounding of the intermediate
  row splits of weight gradients (encoder tape, decoder one-tap group) == the unsplit launches, to fp32 summation order"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_token_masks_equal_mask_generate_and_pad():
    from glow_tts_amd import encoder
    tl = torch.tensor([120, 1, 57, 119, 64], device="cuda")
    T = 120
    mask, rowmask = encoder.token_masks(tl, T)
    want = (torch.arange(T, device="cuda")[None, :] < tl[:, None]).unsqueeze(1).float()
    assert torch.equal(mask, want)
    assert torch.equal(rowmask, torch.nn.functional.pad(want.squeeze(1), (encoder.ROW_PAD, encoder.ROW_PAD)).reshape(-1))


@pytest.mark.parametrize("S", [1, 2, 4])
def test_sum_slices_and_segments(S):
    from glow_tts_amd import _lib, decoder as D
    L = D._L()
    L.glowtts_sum_slices.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p]
    torch.manual_seed(S)
    n = 4 * 12345
    part = torch.randn(S, n, device="cuda")
    want = part[0].clone()
    for s in range(1, S):
        want = want + part[s]
    out = torch.empty(n, device="cuda")
    _lib.check(L.glowtts_sum_slices(part.data_ptr(), out.data_ptr(), S, n, _lib.stream()), "glowtts_sum_slices")
    assert torch.equal(out, want)
    sizes = [4 * 100, 4 * 7, 4 * 12000, n - 4 * 12107]
    dsts = [torch.full((k + 8,), 7.0, device="cuda") for k in sizes]             # (8 guard elements behind each destination)
    segs, off = (D.SumSeg * len(sizes))(), 0
    for i, (k, d) in enumerate(zip(sizes, dsts)):
        segs[i] = D.SumSeg(d.data_ptr(), off, k)
        off += k
    _lib.check(L.glowtts_sum_slices_seg(part.data_ptr(), S, n, segs, len(sizes), _lib.stream()), "glowtts_sum_slices_seg")
    off = 0
    for k, d in zip(sizes, dsts):
        assert torch.equal(d[:k], want[off:off + k]) and bool((d[k:] == 7.0).all())
        off += k
    segs[1] = D.SumSeg(dsts[1].data_ptr(), 4 * 100 + 4, 4 * 7)                   # segments must tile the slice: a gap is refused
    assert L.glowtts_sum_slices_seg(part.data_ptr(), S, n, segs, len(sizes), _lib.stream()) != 0


def test_gru_register_kernels_match_torch_gru():
    """H = 128 (the GST encoder's size) takes the kernels that keep W_hh in registers; torch's GRU cell arithmetic in float64 is the reference."""
    from glow_tts_amd.prosody import _GRUFunction
    torch.manual_seed(0)
    B, T, I, H = 5, 13, 256, 128
    gru = torch.nn.GRU(I, H, 1, batch_first=True).cuda()
    x = torch.randn(B, T, I, device="cuda", requires_grad=True)
    hs = _GRUFunction.apply(x, gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0)
    dh = torch.randn_like(hs)
    got = torch.autograd.grad(hs, (x, gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0), dh)
    ref = torch.nn.GRU(I, H, 1, batch_first=True).double()
    ref.load_state_dict({k: v.detach().cpu().double() for k, v in gru.state_dict().items()})
    xr = x.detach().cpu().double().requires_grad_(True)
    hr = ref(xr)[0]
    want = torch.autograd.grad(hr, (xr, ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0), dh.cpu().double())
    assert (hs.detach().cpu().double() - hr.detach()).abs().max() <= 2e-6
    for a, b in zip(got, want):
        assert (a.cpu().double() - b).abs().max() <= 2e-5 * b.abs().max().clamp_min(1e-3), (tuple(a.shape), float((a.cpu().double() - b).abs().max()))


def _encoder_grads(seed=3):
    from helpers import tiny_hp_dict
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS
    hp = tiny_hp_dict("Vanilla")
    hp["HIP_Precision"] = "bf16"
    hp["Encoder"]["Channels"] = 192                                              # (the block functions / LDS-DMA convs serve 64-channel multiples)
    hp["Encoder"]["Transformer"]["Conv"]["Calc_Channels"] = 768
    for k in ("Prenet", "Transformer", "Duration_Predictor"):
        hp["Encoder"][k]["Dropout_Rate"] = 0.1
    torch.manual_seed(seed)
    model = GlowTTS(Recursive_Parse(hp)).cuda().train()
    return model, hp


@pytest.mark.parametrize("switch", ["gate_in_dgrad", "wgrad_splits", "defer_rel_sum"])
def test_encoder_backward_switches_change_no_gradient(switch):
    """The encoder's backward with a round-5 switch off against on, same weights, same dropout seed (the seed word is drawn by torch's generator: re-seeded):
    the gate in the data-gradient conv's epilogue skips one bf16 rounding of an intermediate (<= 2e-2 of a tensor's largest entry), the row splits and the
    deferred relative-position sums only change fp32 summation order (<= 1e-5)."""
    from glow_tts_amd import conv_fn, encoder
    model, hp = _encoder_grads()
    P = dict(model.named_parameters())
    B, T = 4, 60
    g = torch.Generator().manual_seed(1)
    tokens = torch.randint(0, 30, (B, T), generator=g).cuda()
    tl = torch.tensor([60, 41, 17, 60]).cuda()

    def run(value):
        old = conv_fn.FUSE[switch]
        conv_fn.FUSE[switch] = value
        try:
            model.zero_grad(set_to_none=True)
            torch.manual_seed(11)
            mask, rowmask = encoder.token_masks(tl, T)
            mean, log_std, log_dur = encoder.encoder_forward(P, model.hp, tokens, mask, None, None, True, precision=1, cache=model._enc_cache, rowmask=rowmask)
            (mean.square().sum() + (log_std * 0.3).sum() + log_dur.sum()).backward()
            torch.cuda.synchronize()
            return {k: p.grad.detach().clone() for k, p in P.items() if p.grad is not None}
        finally:
            conv_fn.FUSE[switch] = old
    on = run(conv_fn.FUSE[switch])
    off = run(type(conv_fn.FUSE[switch])(0 if switch != "wgrad_splits" else 1))
    assert set(on) == set(off) and len(on) > 50
    tol = 2e-2 if switch == "gate_in_dgrad" else 1e-5
    for k in on:
        if k.endswith("Key.bias"):                                               # (softmax ignores a key bias: its true gradient is 0, what is computed is rounding noise)
            continue
        err = (on[k] - off[k]).abs().max().item() / (off[k].abs().max().item() + 1e-12)
        assert err <= tol, (k, err)


def test_decoder_one_tap_row_splits_change_no_gradient():
    from glow_tts_amd import decoder as D
    from helpers import full_width_state
    g = torch.Generator().manual_seed(2)
    cfg, sd = full_width_state(2, g)
    dc = D.DecoderConfig(cfg.mel_dim, 2, cfg.n_squeeze, cfg.n_split, cfg.wn_channels, cfg.wn_layers, cfg.wn_kernel, 1)
    mels = torch.randn(4, 80, 208, generator=g).cuda()
    ml = torch.tensor([208, 150, 96, 208]).cuda()

    def run(splits):
        old = D.TUNE["wgrad_tail_splits"]
        D.TUNE["wgrad_tail_splits"] = splits
        try:
            P = {k: v.cuda().requires_grad_(v.is_floating_point()) for k, v in sd.items() if "Decoder" in k}
            W = D.stack_decoder_weights(P, dc)
            z, logdet, _ = D.DecoderFunction.apply(dc, mels, ml, None, 0.0, None, None, None, *W)
            (z.square().sum() + logdet.sum()).backward()
            torch.cuda.synchronize()
            return {k: p.grad.detach().clone() for k, p in P.items() if p.grad is not None}
        finally:
            D.TUNE["wgrad_tail_splits"] = old
    a, b = run(2), run(1)
    assert set(a) == set(b) and len(a) > 40
    for k in a:
        err = (a[k] - b[k]).abs().max().item() / (b[k].abs().max().item() + 1e-12)
        assert err <= 2e-5, (k, err)
