"""Generates the golden fixtures in this directory by running the UNMODIFIED reference
(/root/reference, imported through _ref_harness.py) and, in the same pass, pins the oracle
(oracle/glowtts_ref.py, oracle/mas_ref.c) against it.

Only runs in the build container (the reference never travels).  The fixtures are data:
inputs + the reference's outputs.  Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")

from _ref_harness import load_reference  # noqa: E402
from oracle import glowtts_ref as O  # noqa: E402
from oracle import mas_ref  # noqa: E402

TINY = {
    "Sound.Mel_Dim": 12,
    "Encoder.Channels": 32,
    "Encoder.Prenet.Stacks": 2,
    "Encoder.Transformer.Conv.Calc_Channels": 48,
    "Encoder.Transformer.Stacks": 2,
    "Encoder.Duration_Predictor.Channels": 24,
    "Decoder.Stack": 3,
    "Decoder.Affine_Coupling.Calc_Channels": 32,
    "Decoder.Affine_Coupling.WaveNet.Num_Layers": 2,
    "Speaker_Embedding.Num_Speakers": 5,
    "Speaker_Embedding.Embedding_Size": 16,
    "Prosody_Encoder.Size": 16,
    "Prosody_Encoder.Reference_Encoder.Conv.Kernel_Size": [3, 3, 3],
    "Prosody_Encoder.Reference_Encoder.Conv.Channels": [4, 8, 8],
    "Prosody_Encoder.Reference_Encoder.Conv.Strides": [2, 2, 2],
    "Prosody_Encoder.Reference_Encoder.GRU.Size": 8,
    "Prosody_Encoder.Style_Token.Num_Tokens": 6,
    "Prosody_Encoder.Style_Token.Size": 16,
    "Prosody_Encoder.Style_Token.Attention_Head": 4,
    "Speaker_Classifier_GR.Channels": [12],
    "Train.Adversarial_Speaker_Weight": 0.05,
    "Use_Cython_Alignment": True,
}


def tiny_cfg(mode):
    return O.Cfg(mode=mode, mel_dim=12, enc_channels=32, prenet_stacks=2, ffn_channels=48, enc_stacks=2,
                 dp_channels=24, n_flows=3, wn_channels=32, wn_layers=2, n_speakers=5, spk_dim=16, pro_dim=16,
                 pe_strides=(2, 2, 2), pe_kernels=(3, 3, 3), pe_heads=4, grl_weight=0.05, gr_hidden=1)


def np_sd(sd):
    return {"sd/" + k: v.detach().numpy().copy() for k, v in sd.items()}


def randomize(model, gen):
    """Make every path non-trivial: the reference zero-inits End (Modules.py:773-778) and its
    inv-1x1 weights are orthogonal (:718-725); perturb so logdet / inverse / coupling are exercised."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("layer_Dict.End.weight") or name.endswith("layer_Dict.End.bias"):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.08)
            elif name.endswith("layers.1.weight"):
                p.add_(torch.randn(p.shape, generator=gen) * 0.15)
            elif name.endswith(".bias") and p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=gen) * 0.05)


def make_model_case(mode, seed, fname):
    M = load_reference(dict(TINY, Mode=mode, **{"Speaker_Embedding.Type": "LUT"}))
    cfg = tiny_cfg(mode)
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    model = M.GlowTTS()
    randomize(model, gen)
    B, Tt, Tm = 3, 13, 40
    tokens = torch.randint(0, 35, (B, Tt), generator=gen)
    token_lengths = torch.tensor([13, 9, 11])
    mel_lengths = torch.tensor([40, 28, 34])
    mels = (torch.randn(B, 12, Tm, generator=gen) * 1.5).clamp(-4, 4)
    for b in range(B):
        tokens[b, token_lengths[b]:] = 1
        mels[b, :, mel_lengths[b]:] = -4.0
    speakers = torch.randint(0, 5, (B,), generator=gen) if mode in ("SE", "GR") else None
    pitches = None
    if mode == "GR":                                  # Datasets.py:240-247: per-frame pitch, zero-padded, same length as the mel
        pitches = torch.rand(B, Tm, generator=gen) * 2.0
        for b in range(B):
            pitches[b, mel_lengths[b]:] = 0.0

    # (1) ActNorm data-dependent init happens on the first (training) forward  Modules.py:685-687
    model.train()
    sd_before = {k: v.clone() for k, v in model.state_dict().items()}
    # dropout would make the init batch irreproducible: zero it for the init pass only
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    model(tokens, token_lengths, mels, mel_lengths, speakers, None, pitches)
    model.eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}

    # (2) training-graph forward + losses + grads, eval() => dropout off
    out = model(tokens, token_lengths, mels, mel_lengths, speakers, None, pitches)
    z, mel_mean, mel_log_std, log_dets, log_dur, log_dur_t, attn, classified = out
    mle = M.MLE_Loss()(z=z, mean=mel_mean, std=mel_log_std, log_dets=log_dets, lengths=mel_lengths)
    length = torch.nn.MSELoss()(log_dur, log_dur_t)
    ce = torch.nn.CrossEntropyLoss()(classified, speakers) if classified is not None else None       # Train.py:213-216
    model.zero_grad()
    (mle + length + (ce if ce is not None else 0.0)).backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    # (3) inference with injected noise: seed right before the call so randn_like is reproducible
    length_scale = torch.tensor([1.0, 1.2, 0.9])
    with torch.no_grad():
        torch.manual_seed(seed + 1)
        pm, pl = (mels, mel_lengths) if mode in ("PE", "GR") else (None, None)
        inf_mels, inf_lengths, inf_attn = model.inference(tokens, token_lengths, pm, pl, speakers, None, pitches, mel_lengths if pitches is not None else None,
                                                          noise_scale=0.667, length_scale=length_scale)
    torch.manual_seed(seed + 1)
    noise = torch.randn(B, 12, inf_attn.shape[2])

    # ---- pin the oracle on the same state dict ----
    o = O.forward_train(sd, cfg, tokens, token_lengths, mels, mel_lengths, speakers, pitches=pitches)
    def chk(a, b, name, tol=2e-5):
        err = (a - b).abs().max().item()
        print(f"  oracle vs reference  {name:18s} max|diff| = {err:.3e}")
        assert err <= tol * max(1.0, b.abs().max().item()), name
    chk(o["z"], z, "z"); chk(o["log_dets"], log_dets, "log_dets", 1e-5)
    chk(o["mel_mean"], mel_mean, "mel_mean"); chk(o["mel_log_std"], mel_log_std, "mel_log_std")
    chk(o["log_dur"], log_dur, "log_dur"); chk(o["log_dur_target"], log_dur_t, "log_dur_target")
    assert torch.equal(o["attn"], attn), "MAS path differs"
    olosses = O.train_losses(o, mel_lengths, cfg, speakers if mode == "GR" else None)
    chk(olosses[0], mle, "mle"); chk(olosses[1], length, "length")
    if mode == "GR":
        chk(o["classified"], classified, "classified speakers"); chk(olosses[2], ce, "speaker CE")
    sdg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    o2 = O.forward_train(sdg, cfg, tokens, token_lengths, mels, mel_lengths, speakers, pitches=pitches)
    sum(O.train_losses(o2, mel_lengths, cfg, speakers if mode == "GR" else None)).backward()
    worst = 0.0
    for k, g in grads.items():
        og = sdg[k].grad
        if og is None:
            assert g.abs().max() == 0, k
            continue
        rel = ((og - g).abs().max() / (g.abs().max() + 1e-5)).item()  # +1e-5: Key.bias grads are analytically 0 (softmax shift invariance)
        if rel > 5e-4:
            print(f"    grad {k}: rel {rel:.2e}  |g|max {g.abs().max().item():.2e}")
        worst = max(worst, rel)
    print(f"  oracle vs reference  grads (all {len(grads)} params) worst rel = {worst:.3e}")
    assert worst < 2e-3
    om, ol, oa = O.inference(sd, cfg, tokens, token_lengths, noise, length_scale, 0.667, speakers, None, pm, pl, pitches,
                             mel_lengths if pitches is not None else None)
    chk(om, inf_mels, "inference mels", 5e-5); assert torch.equal(ol, inf_lengths); assert torch.equal(oa, inf_attn)
    # ActNorm init restated: flow 0 sees the squeezed mels
    x0, m0 = O.squeeze(mels, O.mask_from_lengths(mel_lengths, Tm), 2)
    logs0, bias0 = O.actnorm_init(x0, m0)
    chk(logs0, sd["layer_Dict.Decoder.layer_Dict.Flows.0.layers.0.logs"], "actnorm init logs")
    chk(bias0, sd["layer_Dict.Decoder.layer_Dict.Flows.0.layers.0.bias"], "actnorm init bias")

    data = dict(np_sd(sd))
    data.update({"sd_before/" + k: v.numpy() for k, v in sd_before.items() if k.endswith("layers.0.logs") or k.endswith("layers.0.bias")})
    data.update(tokens=tokens.numpy(), token_lengths=token_lengths.numpy(), mels=mels.numpy(), mel_lengths=mel_lengths.numpy(),
                z=z.detach().numpy(), mel_mean=mel_mean.detach().numpy(), mel_log_std=mel_log_std.detach().numpy(),
                log_dets=log_dets.detach().numpy(), log_dur=log_dur.detach().numpy(), log_dur_target=log_dur_t.detach().numpy(),
                attn=attn.numpy().astype(np.int8), mle=mle.detach().numpy(), length=length.detach().numpy(),
                length_scale=length_scale.numpy(), noise=noise.numpy(), noise_scale=np.float32(0.667),
                inf_mels=inf_mels.numpy(), inf_lengths=inf_lengths.numpy(), inf_attn=inf_attn.numpy().astype(np.int8))
    if speakers is not None:
        data["speakers"] = speakers.numpy()
    if pitches is not None:
        data["pitches"] = pitches.numpy()
    if classified is not None:
        data.update(classified=classified.detach().numpy(), ce=ce.detach().numpy())
    data.update({"grad/" + k: v.numpy() for k, v in grads.items()})
    np.savez_compressed(os.path.join(HERE, fname), **data)
    print(f"wrote {fname}: {os.path.getsize(os.path.join(HERE, fname)) / 1024:.0f} KiB")


def make_mas_cases():
    """MAS known-answer cases produced by the reference's compiled core.pyx (oracle/_ref)."""
    core = mas_ref.reference_core()
    rng = np.random.default_rng(20260929)
    cases = {}
    specs = [("ragged", 5, 24, 70, 1.0, False), ("ties", 4, 16, 50, 1.0, True), ("square", 3, 12, 12, 1.0, False),
             ("one_token", 3, 6, 30, 1.0, False), ("x1000", 4, 20, 60, 1000.0, False), ("wide", 2, 130, 300, 1.0, False)]
    for name, B, Tx, Ty, scale, ties in specs:
        v = (rng.normal(-100, 30, (B, Tx, Ty)) * scale).astype(np.float32)
        if ties:
            v = (np.round(v / 40) * 40).astype(np.float32)
        tx = rng.integers(1, Tx + 1, B).astype(np.int32)
        ty = np.array([rng.integers(t, Ty + 1) for t in tx], dtype=np.int32)
        if name == "square":
            ty = tx.copy()
        if name == "one_token":
            tx[:] = 1
        tx[0], ty[0] = (Tx, Ty) if name != "one_token" else (1, Ty)
        mask = (np.arange(Tx)[None, :, None] < tx[:, None, None]) & (np.arange(Ty)[None, None, :] < ty[:, None, None])
        v = (v * mask).astype(np.float32)
        q = v.copy()
        path = np.zeros((B, Tx, Ty), dtype=np.int32)
        core.maximum_path_c(path, q, tx, ty)
        p2, q2 = mas_ref.maximum_path_c(v, tx, ty, return_q=True)
        assert (p2 == path).all() and (q2.view(np.int32) == q.view(np.int32)).all(), name
        cases.update({f"{name}/value": v, f"{name}/t_x": tx, f"{name}/t_y": ty, f"{name}/path": path.astype(np.int8),
                      f"{name}/q_sum": np.float64(q.astype(np.float64).sum()),
                      f"{name}/q_xor": np.bitwise_xor.reduce(q.view(np.uint32).ravel())})
        print(f"  MAS case {name}: oracle C == reference core.pyx (path + cumulative values bit-exact)")
    # more tokens than frames: no monotonic alignment exists; core.pyx then accumulates nothing and backtracks over the raw inputs
    for name, B, Tx, Ty in (("more_tokens", 3, 40, 17),):
        v = rng.normal(-100, 30, (B, Tx, Ty)).astype(np.float32)
        tx, ty = np.array([Tx, Tx - 9, 5], dtype=np.int32), np.array([Ty, Ty - 8, 1], dtype=np.int32)
        mask = (np.arange(Tx)[None, :, None] < tx[:, None, None]) & (np.arange(Ty)[None, None, :] < ty[:, None, None])
        v = (v * mask).astype(np.float32)
        q = v.copy()
        path = np.zeros((B, Tx, Ty), dtype=np.int32)
        core.maximum_path_c(path, q, tx, ty)
        p2, q2 = mas_ref.maximum_path_c(v, tx, ty, return_q=True)
        assert (p2 == path).all() and (q2.view(np.int32) == q.view(np.int32)).all() and (q == v).all(), name
        cases.update({f"{name}/value": v, f"{name}/t_x": tx, f"{name}/t_y": ty, f"{name}/path": path.astype(np.int8),
                      f"{name}/q_sum": np.float64(q.astype(np.float64).sum()),
                      f"{name}/q_xor": np.bitwise_xor.reduce(q.view(np.uint32).ravel())})
        print(f"  MAS case {name}: oracle C == reference core.pyx (t_x > t_y: raw-input backtrack)")
    np.savez_compressed(os.path.join(HERE, "mas_cases.npz"), **cases)
    print(f"wrote mas_cases.npz: {os.path.getsize(os.path.join(HERE, 'mas_cases.npz')) / 1024:.0f} KiB")


if __name__ == "__main__":
    make_mas_cases()
    if "--mas-only" in sys.argv:
        sys.exit(0)
    make_model_case("Vanilla", 1234, "tiny_vanilla.npz")
    make_model_case("SE", 4321, "tiny_se.npz")
    make_model_case("PE", 777, "tiny_pe.npz")
    make_model_case("GR", 999, "tiny_gr.npz")
