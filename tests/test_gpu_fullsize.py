"""GPU parity at REAL size: the default Hyper_Parameters.yaml model (12 flows x 4 WaveNet layers x 192 channels, 6 encoder layers of
192 channels / 2 heads of 96 / 768-channel FFN), B = 4, 800-frame / 120-token padded batch with ragged lengths, through
`GlowTTS.forward` + `MLE_Loss` (Modules.py:50-126, 1020-1029) against `oracle.forward_train` on the same state dict.

f32 mode (the reference's arithmetic; north_star bar "mel / NLL within 1e-3 fp32", alignments bit-exact):
   encoder outputs (mean, log_std, log_dur: the D = 96 MFMA attention core, k = 3 768-channel FFN convs, LayerNorm at C = 192)
   <= 1e-4, z <= 1e-3, alignment exact, per-utterance log-determinant <= 1e-3 relative, |NLL - oracle| <= 1e-3, every parameter
   gradient of (MLE + duration loss) within 5e-3 of the oracle's (relative to the tensor's largest entry).
bf16 mode (the benchmarked dtype: bf16 MFMA operands and bf16-stored WaveNet state / gates, fp32 everything else):
   |NLL - oracle| <= 1e-3, per-utterance log-determinant <= 1e-3 relative, fraction of frames whose alignment differs reported and
   bounded, decoder gradients cosine >= 0.995, norm ratio 0.97..1.03.  A miss here is a finding to fix, not a tolerance to loosen."""
import copy

import numpy as np
import pytest
import torch

from oracle import glowtts_ref as O

pytestmark = pytest.mark.gpu

B, TT, TM = 4, 120, 800
TOK_LEN = [120, 104, 75, 31]
MEL_LEN = [800, 702, 500, 210]


def _hp(precision):
    from glow_tts_amd import hparams
    d = copy.deepcopy(hparams.load_yaml(hparams.DEFAULT_YAML))
    d["Mode"] = "Vanilla"
    d["HIP_Precision"] = precision
    return d


@pytest.fixture(scope="module")
def case():
    """A seeded full-size model whose coupling layers are NOT the identity (the reference zero-initialises End, Modules.py:773-778:
    that would hide the WaveNet from z), ActNorm initialised from the batch by the f32 HIP path (checked against the reference's
    formula in test_gpu_decoder.py); the oracle and both precisions then share one state dict."""
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS
    torch.manual_seed(2024)
    d = _hp("f32")
    model = GlowTTS(Recursive_Parse(d))
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        for f in model.layer_Dict["Decoder"].layer_Dict["Flows"]:
            end = f.layers[2].layer_Dict["End"]
            end.weight.copy_(torch.randn(end.weight.shape, generator=g) * 0.02)
            end.bias.copy_(torch.randn(end.bias.shape, generator=g) * 0.02)
            f.layers[1].weight.add_(0.05 * torch.randn(4, 4, generator=g))
    tokens = torch.randint(0, 35, (B, TT), generator=g)
    mels = (torch.randn(B, 80, TM, generator=g) * 1.5).clamp(-4, 4)
    tl, ml = torch.tensor(TOK_LEN), torch.tensor(MEL_LEN)
    for b in range(B):                                   # the reference's padding (Datasets.py:225-250): <E> = 1, mel pad = -Max_Abs_Mel
        tokens[b, TOK_LEN[b]:] = 1
        mels[b, :, MEL_LEN[b]:] = -4.0
    model = model.cuda().eval()
    with torch.no_grad():
        model(tokens.cuda(), tl.cuda(), mels.cuda(), ml.cuda(), None, None, None)        # first call: ActNorm data init (Modules.py:685-687)
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cfg = O.Cfg.from_yaml_dict(d)
    sdg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    out = O.forward_train(sdg, cfg, tokens, tl, mels, ml)
    mle, length = O.train_losses(out, ml, cfg)
    (mle + length).backward()
    grads = {k: v.grad for k, v in sdg.items() if v.grad is not None}
    return dict(sd=sd, cfg=cfg, tokens=tokens, tl=tl, mels=mels, ml=ml, out={k: v.detach() for k, v in out.items() if v is not None}, mle=mle.item(),
                length=length.item(), grads=grads)


def _run(case, precision):
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS, MLE_Loss
    model = GlowTTS(Recursive_Parse(_hp(precision)))
    model.load_state_dict(case["sd"], strict=True)
    for f in model.layer_Dict["Decoder"].layer_Dict["Flows"]:
        f.layers[0].initialized = True
    model = model.cuda().eval()
    c = lambda k: case[k].cuda()
    from helpers import launch_counts, launch_reset
    launch_reset()
    z, mel_mean, mel_log_std, log_dets, log_dur, log_dur_t, attn, _ = model(c("tokens"), c("tl"), c("mels"), c("ml"), None, None, None)
    mle = MLE_Loss(model.hp)(z=z, mean=mel_mean, std=mel_log_std, log_dets=log_dets, lengths=c("ml"))
    length = torch.nn.functional.mse_loss(log_dur, log_dur_t)
    (mle + length).backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}
    cpu = lambda t: t.detach().cpu()
    return dict(z=cpu(z), mel_mean=cpu(mel_mean), mel_log_std=cpu(mel_log_std), log_dets=cpu(log_dets), log_dur=cpu(log_dur),
                log_dur_target=cpu(log_dur_t), attn=cpu(attn), mle=mle.item(), length=length.item(), grads=grads, launches=launch_counts())


def _masks(case):
    tmask = O.mask_from_lengths(case["tl"], TT)
    mmask = O.mask_from_lengths(case["ml"], TM)
    return tmask, mmask


def test_full_size_f32_matches_oracle(case):
    r, o = _run(case, "f32"), case["out"]
    tmask, mmask = _masks(case)
    # alignment: bit-exact (MAS on fp32 log-priors, Modules.py:107-116)
    assert torch.equal(r["attn"], o["attn"]), f"{(r['attn'] != o['attn']).any(1).sum().item()} frames aligned differently"
    # full-width encoder, through its expansion by the (identical) alignment: mean / log_std per frame, log_dur per token
    assert ((r["mel_mean"] - o["mel_mean"]) * mmask).abs().max() <= 1e-4
    assert ((r["mel_log_std"] - o["mel_log_std"]) * mmask).abs().max() <= 1e-4
    assert ((r["log_dur"] - o["log_dur"]) * tmask).abs().max() <= 1e-4
    assert (r["log_dur_target"] - o["log_dur_target"]).abs().max() <= 1e-5
    assert ((r["z"] - o["z"]) * mmask).abs().max() <= 1e-3
    assert (r["z"] * (1 - mmask)).abs().max() == 0
    assert ((r["log_dets"] - o["log_dets"]).abs() <= 1e-3 * o["log_dets"].abs().clamp_min(1.0)).all()
    assert abs(r["mle"] - case["mle"]) <= 1e-3 and abs(r["length"] - case["length"]) <= 1e-3
    worst = ("", 0.0)
    for k, want in case["grads"].items():
        got = r["grads"].get(k)
        got = got if got is not None else torch.zeros_like(want)
        err = (got - want).abs().max().item() / (want.abs().max().item() + 1e-6)
        worst = max(worst, (k, err), key=lambda t: t[1])
        assert err <= 5e-3, (k, err)
    print(f"f32 full size: NLL {r['mle']:.6f} vs oracle {case['mle']:.6f}; worst gradient {worst}")


def test_full_size_bf16_nll_within_1e3(case):
    r, o = _run(case, "bf16"), case["out"]
    tmask, mmask = _masks(case)
    differ = ((r["attn"] != o["attn"]).any(1).float() * mmask[:, 0]).sum().item() / mmask.sum().item()
    dz = ((r["z"] - o["z"]) * mmask).abs().max().item()
    rel_ld = ((r["log_dets"] - o["log_dets"]).abs() / o["log_dets"].abs().clamp_min(1.0)).max().item()
    print(f"bf16 full size: NLL {r['mle']:.6f} vs oracle {case['mle']:.6f} (|d| = {abs(r['mle'] - case['mle']):.2e}); max |dz| {dz:.3e}; "
          f"log-det rel {rel_ld:.2e}; frames aligned differently {100 * differ:.3f} %")
    assert abs(r["mle"] - case["mle"]) <= 1e-3, (r["mle"], case["mle"])
    assert rel_ld <= 1e-3
    assert dz <= 0.1
    assert differ <= 0.01
    assert ((r["log_dur"] - o["log_dur"]) * tmask).abs().max() <= 3e-2
    report = []
    for k, want in case["grads"].items():
        if "Decoder" not in k:
            continue
        a, b = r["grads"][k].flatten().double(), want.flatten().double()
        report.append(((a @ b / (a.norm() * b.norm() + 1e-30)).item(), (a.norm() / (b.norm() + 1e-30)).item(), k))
    report.sort()
    print("bf16 full size, worst decoder gradient tensors (cosine, norm ratio):", report[:3])
    for cos, ratio, k in report:
        assert cos >= 0.995 and 0.97 <= ratio <= 1.03, (k, cos, ratio)      # (observed 0.9999 / 1.00: a regression to 0.99 is a finding)
    # the text encoder on bf16-STORED rows (round 3): LayerNorm writes a bf16 copy, the FFN / QKV / duration-predictor convs and their data
    # gradients take the LDS-DMA kernel.  6 layers x (Conv_0, Conv_1) forward + 6 x 2 data gradients + prenet / duration predictor.
    n = r["launches"]
    assert n.get("conv_dma<LINEAR,3>", 0) >= 24 and n.get("conv_dma<LINEAR,1>", 0) >= 6, {k: v for k, v in n.items() if "conv" in k}
    enc = []
    for k, want in case["grads"].items():
        # (Key.bias: a constant added to every key shifts all scores of a query alike - softmax cancels it, its true gradient is zero
        #  and both sides hold rounding noise only)
        if "Encoder" not in k or want.abs().max() == 0 or k.endswith("Key.bias"):
            continue
        a, b = r["grads"][k].flatten().double(), want.flatten().double()
        enc.append(((a @ b / (a.norm() * b.norm() + 1e-30)).item(), (a.norm() / (b.norm() + 1e-30)).item(), k))
    enc.sort()
    print("bf16 full size, worst encoder gradient tensors (cosine, norm ratio):", enc[:3])
    for cos, ratio, k in enc:
        assert cos >= 0.99 and 0.95 <= ratio <= 1.05, (k, cos, ratio)


def test_reference_maximum_sizes_f32(case):
    """The reference's data filter admits texts of up to 200 letters (+ <S>, <E> = 202 tokens) and 1000 mel frames
    (Hyper_Parameters.yaml:94-99): above 128 tokens the encoder takes the two-workgroup `attn_*_long_*` kernels, which the 120-token cases
    never select.  Same model as above (ActNorm already initialised), B = 2 ragged, f32 against the oracle: alignment exact, NLL 1e-3,
    encoder outputs 1e-4, every gradient 5e-3."""
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS, MLE_Loss
    g = torch.Generator().manual_seed(31)
    Bm, Tt, Tm = 2, 202, 1000
    tl, ml = torch.tensor([202, 137]), torch.tensor([1000, 612])
    tokens = torch.randint(0, 35, (Bm, Tt), generator=g)
    mels = (torch.randn(Bm, 80, Tm, generator=g) * 1.5).clamp(-4, 4)
    for b in range(Bm):
        tokens[b, tl[b]:] = 1
        mels[b, :, ml[b]:] = -4.0
    cfg = case["cfg"]
    sdg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in case["sd"].items()}
    o = O.forward_train(sdg, cfg, tokens, tl, mels, ml)
    omle, olen = O.train_losses(o, ml, cfg)
    (omle + olen).backward()
    model = GlowTTS(Recursive_Parse(_hp("f32")))
    model.load_state_dict(case["sd"], strict=True)
    for f in model.layer_Dict["Decoder"].layer_Dict["Flows"]:
        f.layers[0].initialized = True
    model = model.cuda().eval()
    z, mm, ms, ld, dur, durt, attn, _ = model(tokens.cuda(), tl.cuda(), mels.cuda(), ml.cuda(), None, None, None)
    mle = MLE_Loss(model.hp)(z=z, mean=mm, std=ms, log_dets=ld, lengths=ml.cuda())
    length = torch.nn.functional.mse_loss(dur, durt)
    (mle + length).backward()
    torch.cuda.synchronize()
    tmask, mmask = O.mask_from_lengths(tl, Tt), O.mask_from_lengths(ml, Tm)
    assert torch.equal(attn.cpu(), o["attn"])
    assert ((z.detach().cpu() - o["z"].detach()) * mmask).abs().max() <= 1e-3
    assert ((mm.detach().cpu() - o["mel_mean"].detach()) * mmask).abs().max() <= 1e-4
    assert ((dur.detach().cpu() - o["log_dur"].detach()) * tmask).abs().max() <= 1e-4
    assert abs(mle.item() - omle.item()) <= 1e-3 and abs(length.item() - olen.item()) <= 1e-3
    for k, p in model.named_parameters():
        want = sdg[k].grad
        if want is None:
            continue
        err = (p.grad.cpu() - want).abs().max().item() / (want.abs().max().item() + 1e-6)
        assert err <= 5e-3, (k, err)
