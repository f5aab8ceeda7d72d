"""Child of tests/test_gpu_benchmarked_sizes.py::test_long_form_inverse_at_full_width (it captures hipGraphs; a failed capture takes the
process down).  BASELINE config 5's second half at full width: 2 utterances x 200 tokens -> more than 2000 mel frames each through
`GlowTTS.inference` (eager) and `GraphedInference` (two replayed graphs) against `oracle.inference` with the same injected noise
(Modules.py:128-204) - in Vanilla mode and in PE mode (config 5's own: the GST prosody vector conditions the duration predictor and every
WaveNet layer of the inverse flow, Modules.py:159-160, 863-866).

bf16.  bf16 ENCODER arithmetic may move one ceil(exp(log_dur) * scale) by a frame (Modules.py:173), which shifts every
frame behind it - a comparison of the whole call then measures that shift, not the inverse flow.  So the bf16 INVERSE FLOW is measured on
the f32 path's front half (`inference_front`: mean, log_std, durations, lengths - equal to the oracle's, asserted): the bf16 model's
`inference_back` on those inputs against the oracle's mel.  The measured error is printed ("BF16 INVERSE MEL") and held under the bar
BF16_MEL_BAR; it is NOT 1e-3 (north_star's tolerance, which the f32 mode meets at <= 2e-4): bf16 MFMA operands through 12 flows x 4 layers
give ~1e-2 rms (0.3 worst element) on mels in [-4, 4].  `Inference.py` therefore runs in `HIP_Precision: 'f32'` unless the yaml asks for bf16
explicitly (glow_tts_amd/inferencer.py; README "Precision")."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import glowtts_ref as O                                  # noqa: E402
from test_gpu_benchmarked_sizes import _build, _hp                    # noqa: E402
from glow_tts_amd.graph_infer import GraphedInference                 # noqa: E402

# Measured (MI355X, round 6): Vanilla max 0.295 / rms 1.11e-2 over 4550 frames of mels in [-4, 4]; the bars are those with headroom for another seed.
BF16_MEL_BAR = 0.6           # max |mel - oracle| of the bf16 inverse flow on identical (f32) prior inputs, > 2000 frames
BF16_MEL_RMS_BAR = 2.5e-2


def run_mode(mode):
    torch.manual_seed(11)
    g = torch.Generator().manual_seed(12)
    model = _build(mode, "f32")
    with torch.no_grad():
        for f in model.layer_Dict["Decoder"].layer_Dict["Flows"]:
            end = f.layers[2].layer_Dict["End"]
            end.weight.copy_(torch.randn(end.weight.shape, generator=g) * 0.02)
            end.bias.copy_(torch.randn(end.bias.shape, generator=g) * 0.02)
            f.layers[1].weight.add_(0.05 * torch.randn(4, 4, generator=g))
            f.layers[0].logs.copy_(torch.randn(f.layers[0].logs.shape, generator=g) * 0.1)
            f.layers[0].bias.copy_(torch.randn(f.layers[0].bias.shape, generator=g) * 0.1)
            f.layers[0].initialized = True
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = O.Cfg.from_yaml_dict(_hp(mode, "f32"))
    B, Tt = 2, 200
    tl = torch.tensor([200, 163])
    tokens = torch.randint(2, 35, (B, Tt), generator=g)
    for b in range(B):
        tokens[b, tl[b]:] = 1
    noise = torch.randn(B, 80, 4096, generator=g)
    pe, pe_dev = {}, {}
    if mode == "PE":                                         # the prosody reference: ragged mels of ordinary length (Inference.py:56-58)
        ref_l = torch.tensor([412, 300])
        ref = (torch.randn(B, 80, 412, generator=g) * 1.5).clamp(-4, 4)
        ref[1, :, 300:] = -4.0
        pe = dict(mels_for_prosody=ref, mel_lengths_for_prosody=ref_l)
        pe_dev = {k: v.cuda() for k, v in pe.items()}
    # length scales that stretch whatever the (random-init) duration predictor says to > 2000 frames per utterance (ceil() per token: iterate)
    ls = torch.tensor([8.0, 8.0])
    for _ in range(4):
        with torch.no_grad():
            want, wl, wa = O.inference(sd, cfg, tokens, tl, noise, ls, noise_scale=0.667, **pe)
        if int(wl.min()) >= 2100 and int(wl.max()) <= 2900:
            break
        ls = ls * torch.tensor([2400.0, 2250.0]) / wl.float()
    assert int(wl.min()) >= 2000 and int(wl.max()) <= 3072, wl
    f32_front = None
    for precision in ("f32", "bf16"):
        m = _build(mode, precision, sd).cuda().eval()
        with torch.no_grad():
            mels, lengths, attn = m.inference(tokens.cuda(), tl.cuda(), pe_dev.get("mels_for_prosody"), pe_dev.get("mel_lengths_for_prosody"), None, None,
                                              None, None, noise_scale=0.667, length_scale=ls.cuda(), noises=noise.cuda())
        torch.cuda.synchronize()
        if precision == "f32":                                  # the reference's arithmetic: lengths and alignment exact, mels within 2e-4
            assert torch.equal(lengths.cpu(), wl), (lengths, wl)
            assert torch.equal(attn.cpu().to(wa.dtype), wa)
            err = (mels.cpu() - want).abs().max().item()
            assert mels.shape == want.shape and err <= 2e-4, err
            print(f"{mode} eager f32: {tuple(mels.shape)} frames {wl.tolist()}, max |mel - oracle| {err:.2e}")
            with torch.no_grad():
                f32_front = m.inference_front(tokens.cuda(), tl.cuda(), pe_dev.get("mels_for_prosody"), pe_dev.get("mel_lengths_for_prosody"), None, None, ls.cuda())
        else:
            # the whole bf16 call: length drift, finiteness, masking (the encoder's bf16 arithmetic may move a ceil() by one frame)
            assert (lengths.cpu() - wl).abs().max() <= 4, (lengths, wl)
            assert torch.isfinite(mels).all()
            for b in range(B):
                n = (int(lengths[b]) // 2) * 2
                assert (mels[b, :, n:] == -4.0).all()
            print(f"{mode} eager bf16: frames {lengths.tolist()} (oracle {wl.tolist()})")
            # the bf16 INVERSE FLOW on the f32 front half (= the oracle's durations, lengths and prior, asserted above): the arithmetic alone
            with torch.no_grad():
                bm, bl, ba = m.inference_back(f32_front, None, 0.667, noise.cuda())
            torch.cuda.synchronize()
            assert torch.equal(bl.cpu(), wl) and torch.equal(ba.cpu().to(wa.dtype), wa) and bm.shape == want.shape
            d = (bm.cpu() - want)
            valid = O.mask_from_lengths((wl // 2) * 2, want.shape[2])
            berr, brms = d.abs().max().item(), float(((d * valid) ** 2).sum() / (valid.sum() * 80)) ** 0.5
            p999 = float(torch.quantile(d.abs().flatten()[::7].float(), 0.999))
            print(f"BF16 INVERSE MEL {mode}: max |mel - oracle| {berr:.3e}, 99.9th percentile {p999:.3e}, rms {brms:.3e} over {int(wl.sum())} frames "
                  f"(bars {BF16_MEL_BAR}, {BF16_MEL_RMS_BAR}; f32 mode: 2e-4)")
            assert berr <= BF16_MEL_BAR and brms <= BF16_MEL_RMS_BAR, (berr, brms)
        gi = GraphedInference(m, mel_buckets=(2048, 2560, 3072))
        for rep in range(2):
            gm, gl, ga = gi(tokens.cuda(), tl.cuda(), noise_scale=0.667, length_scale=ls.cuda(), noises=noise.cuda(), **pe_dev)
            torch.cuda.synchronize()
            assert torch.equal(gl, lengths) and torch.equal(ga, attn)
            gerr = (gm - mels).abs().max().item()
            assert gm.shape == mels.shape and gerr <= (1e-4 if precision == "f32" else 0.1), (precision, rep, gerr)      # (another padded length: other row tiles, other summation order)
        print(f"{mode} graphed {precision}: max |graphed - eager| {gerr:.2e}")
        del gi, m
        torch.cuda.empty_cache()


for mode_ in (sys.argv[1:] or ["Vanilla", "PE"]):
    run_mode(mode_)
print("LONGFORM OK")
