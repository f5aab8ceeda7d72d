"""GPU parity of the whole training graph / inverse path (glow_tts_amd.modules.GlowTTS) vs the golden vectors of
the reference and the oracle.  The golden state dict is loaded through load_state_dict: it also checks that the
key layout is the reference's."""
import numpy as np
import pytest
import torch

from oracle import glowtts_ref as O
from helpers import load_case, tiny_cfg, tiny_hp_dict  # noqa: F401

pytestmark = pytest.mark.gpu


def build(mode, precision, sd):
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS
    hp = tiny_hp_dict(mode)
    hp["HIP_Precision"] = precision
    model = GlowTTS(Recursive_Parse(hp))
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    for f in model.layer_Dict["Decoder"].layer_Dict["Flows"]:      # what the reference does after loading (Train.py:527-528)
        f.layers[0].initialized = True
    return model.cuda().eval()


ALL_MODES = [("Vanilla", "tiny_vanilla.npz"), ("SE", "tiny_se.npz"), ("PE", "tiny_pe.npz"), ("GR", "tiny_gr.npz")]


@pytest.mark.parametrize("mode,fname", ALL_MODES)
def test_state_dict_keys_match_reference(mode, fname):
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS
    sd, _, _ = load_case(fname)
    model = GlowTTS(Recursive_Parse(tiny_hp_dict(mode)))
    mine = model.state_dict()
    assert set(mine.keys()) == set(sd.keys())
    for k in sd:
        assert tuple(mine[k].shape) == tuple(sd[k].shape), k


@pytest.mark.parametrize("mode,fname", ALL_MODES)
def test_train_forward_losses_grads_f32(mode, fname):
    """Every Mode of Hyper_Parameters.yaml:17-18 against the golden vectors the reference produced in that mode: the 8 outputs of
    GlowTTS.forward (Modules.py:50-126; GR: also classified_Speakers), MLE + duration (+ speaker CE, Train.py:213-216) losses and
    every parameter gradient - PE through the GST prosody encoder, GR through the gradient-reversal classifier and the per-frame pitch
    conditioning of the WaveNet (Modules.py:867-869)."""
    from glow_tts_amd.modules import MLE_Loss
    sd, grads, r = load_case(fname)
    model = build(mode, "f32", sd)
    t = lambda k: torch.from_numpy(r[k]).cuda()
    spk = t("speakers") if "speakers" in r else None
    pit = t("pitches") if "pitches" in r else None
    z, mel_mean, mel_log_std, log_dets, log_dur, log_dur_t, attn, classified = model(t("tokens"), t("token_lengths"), t("mels"), t("mel_lengths"), spk, None, pit)
    torch.cuda.synchronize()
    assert np.array_equal(attn.cpu().numpy().astype(np.int8), r["attn"]), "alignment differs from the reference"
    for got, key, tol in [(z, "z", 1e-4), (mel_mean, "mel_mean", 1e-4), (mel_log_std, "mel_log_std", 1e-4), (log_dur, "log_dur", 1e-4),
                          (log_dur_t, "log_dur_target", 1e-5)]:
        assert (got.detach().cpu() - torch.from_numpy(r[key])).abs().max() <= tol, key
    assert (log_dets.detach().cpu() - torch.from_numpy(r["log_dets"])).abs().max() <= 1e-3
    mle = MLE_Loss(model.hp)(z=z, mean=mel_mean, std=mel_log_std, log_dets=log_dets, lengths=t("mel_lengths"))
    length = torch.nn.functional.mse_loss(log_dur, log_dur_t)
    assert abs(mle.item() - float(r["mle"])) <= 1e-4 and abs(length.item() - float(r["length"])) <= 1e-4      # NLL within 1e-3 (north_star)
    total = mle + length
    if mode == "GR":
        assert classified is not None and (classified.detach().cpu() - torch.from_numpy(r["classified"])).abs().max() <= 1e-4
        ce = torch.nn.functional.cross_entropy(classified, spk)                                                    # Train.py:213-216
        assert abs(ce.item() - float(r["ce"])) <= 1e-4
        total = total + ce
    else:
        assert classified is None
    model.zero_grad()
    total.backward()
    torch.cuda.synchronize()
    worst = 0.0
    for k, p in model.named_parameters():
        want = grads.get(k)
        if want is None:
            continue
        got = p.grad.cpu() if p.grad is not None else torch.zeros_like(want)
        err = (got - want).abs().max().item() / (want.abs().max().item() + 1e-5)
        worst = max(worst, err)
        assert err < 5e-3, (k, err)
    print("worst relative grad error", worst)


def test_ge2e_mode_takes_precomputed_dvectors():
    """Speaker_Embedding.Type 'GE2E' (BASELINE config 4; Modules.py:30-35, 75-77): the GE2E LSTM is an un-vendored submodule of the
    reference, so the L2-normalised d-vectors [B, Embedding_Size] arrive in `mels_for_ge2e`, detached.  Same weights as the SE golden model
    without its LUT; checked against the oracle's float-`speakers` path; reference-style checkpoints that still carry `layer_Dict.GE2E.*`
    keys load strictly (those keys are dropped, modules._drop_ge2e_keys)."""
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS, MLE_Loss
    sd, _, r = load_case("tiny_se.npz")
    sd = {k: v for k, v in sd.items() if k != "layer_Dict.LUT.weight"}
    sd["layer_Dict.GE2E.layer_Dict.LSTM.weight_ih_l0"] = torch.zeros(4, 4)             # what a reference GE2E checkpoint also holds
    hp = tiny_hp_dict("SE")
    hp["Speaker_Embedding"]["Type"] = "GE2E"
    hp["HIP_Precision"] = "f32"
    model = GlowTTS(Recursive_Parse(hp))
    missing, unexpected = model.load_state_dict(dict(sd), strict=True)
    assert not missing and not unexpected
    for f in model.layer_Dict["Decoder"].layer_Dict["Flows"]:
        f.layers[0].initialized = True
    model = model.cuda().eval()
    t = lambda k: torch.from_numpy(r[k])
    g = torch.Generator().manual_seed(5)
    dvec = torch.randn(3, 16, generator=g)
    dvec = (dvec / dvec.norm(dim=1, keepdim=True)).requires_grad_(True)
    dv_gpu = dvec.detach().cuda().requires_grad_(True)
    out = model(t("tokens").cuda(), t("token_lengths").cuda(), t("mels").cuda(), t("mel_lengths").cuda(), None, dv_gpu, None)
    z, mel_mean, mel_log_std, log_dets, log_dur, log_dur_t, attn, _ = out
    loss = MLE_Loss(model.hp)(z=z, mean=mel_mean, std=mel_log_std, log_dets=log_dets, lengths=t("mel_lengths").cuda()) + \
        torch.nn.functional.mse_loss(log_dur, log_dur_t)
    loss.backward()
    torch.cuda.synchronize()
    cfg = tiny_cfg("SE")
    sdg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items() if "GE2E" not in k}
    o = O.forward_train(sdg, cfg, t("tokens"), t("token_lengths"), t("mels"), t("mel_lengths"), dvec)
    omle, olen = O.train_losses(o, t("mel_lengths"), cfg)
    (omle + olen).backward()
    assert torch.equal(attn.cpu(), o["attn"])
    assert (z.detach().cpu() - o["z"].detach()).abs().max() <= 1e-4 and (log_dur.detach().cpu() - o["log_dur"].detach()).abs().max() <= 1e-4
    assert abs(loss.item() - (omle + olen).item()) <= 1e-4
    assert dv_gpu.grad is None and dvec.grad is None                                     # detached (Modules.py:77)
    for k, p in model.named_parameters():
        want = sdg[k].grad
        if want is None:
            continue
        err = (p.grad.cpu() - want).abs().max().item() / (want.abs().max().item() + 1e-5)
        assert err < 5e-3, (k, err)


def test_train_forward_bf16_nll_within_1e3():
    from glow_tts_amd.modules import MLE_Loss
    sd, _, r = load_case("tiny_vanilla.npz")
    model = build("Vanilla", "bf16", sd)
    t = lambda k: torch.from_numpy(r[k]).cuda()
    z, mel_mean, mel_log_std, log_dets, log_dur, log_dur_t, attn, _ = model(t("tokens"), t("token_lengths"), t("mels"), t("mel_lengths"), None, None, None)
    mle = MLE_Loss(model.hp)(z=z, mean=mel_mean, std=mel_log_std, log_dets=log_dets, lengths=t("mel_lengths"))
    assert abs(mle.item() - float(r["mle"])) <= 1e-3 * max(1.0, abs(float(r["mle"])))
    assert (z.detach().cpu() - torch.from_numpy(r["z"])).abs().max() <= 5e-2


@pytest.mark.parametrize("mode,fname", ALL_MODES)
def test_inference_matches_reference(mode, fname):
    """GlowTTS.inference (Modules.py:128-204) in every mode: PE / GR take the prosody reference mels, GR interpolates the pitch track to
    the predicted length (Pitch_Interpolater, :193-196) and feeds it to the inverse flows."""
    sd, _, r = load_case(fname)
    model = build(mode, "f32", sd)
    t = lambda k: torch.from_numpy(r[k]).cuda()
    spk = t("speakers") if "speakers" in r else None
    pm, pl = (t("mels"), t("mel_lengths")) if mode in ("PE", "GR") else (None, None)
    pit, pitl = (t("pitches"), t("mel_lengths")) if mode == "GR" else (None, None)
    mels, lengths, attn = model.inference(t("tokens"), t("token_lengths"), pm, pl, spk, None, pit, pitl,
                                          noise_scale=float(r["noise_scale"]), length_scale=t("length_scale"), noises=t("noise"))
    torch.cuda.synchronize()
    assert torch.equal(lengths.cpu(), torch.from_numpy(r["inf_lengths"]))
    assert np.array_equal(attn.cpu().numpy().astype(np.int8), r["inf_attn"])
    want = torch.from_numpy(r["inf_mels"])
    assert mels.shape == want.shape and (mels.cpu() - want).abs().max() <= 2e-4


def test_actnorm_init_on_first_forward_matches_reference():
    """Fresh ActNorm parameters (zeros, `initialized` False) -> after one forward they equal the reference's."""
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS
    sd, _, r = load_case("tiny_vanilla.npz")
    d = np.load(__import__("os").path.join(__import__("helpers").GOLDEN, "tiny_vanilla.npz"))
    sd0 = dict(sd)
    for k in d.files:
        if k.startswith("sd_before/"):
            sd0[k[len("sd_before/"):]] = torch.from_numpy(d[k])
    hp = tiny_hp_dict("Vanilla"); hp["HIP_Precision"] = "f32"
    model = GlowTTS(Recursive_Parse(hp))
    model.load_state_dict(sd0)
    model = model.cuda().eval()
    t = lambda k: torch.from_numpy(r[k]).cuda()
    model(t("tokens"), t("token_lengths"), t("mels"), t("mel_lengths"), None, None, None)
    for i, f in enumerate(model.layer_Dict["Decoder"].layer_Dict["Flows"]):
        assert f.layers[0].initialized
        assert (f.layers[0].logs.cpu() - sd[f"layer_Dict.Decoder.layer_Dict.Flows.{i}.layers.0.logs"]).abs().max() < 2e-4
        assert (f.layers[0].bias.cpu() - sd[f"layer_Dict.Decoder.layer_Dict.Flows.{i}.layers.0.bias"]).abs().max() < 2e-4


def test_train_bf16_gradients_close_to_reference():
    """bf16 mode (bf16 MFMA operands AND bf16-stored WaveNet state / gates / their gradients): every parameter gradient stays close
    to the reference's fp32 gradient - direction (cosine) and size - so the reduced storage precision does not bias training."""
    from glow_tts_amd.modules import MLE_Loss
    sd, grads, r = load_case("tiny_vanilla.npz")
    model = build("Vanilla", "bf16", sd)
    t = lambda k: torch.from_numpy(r[k]).cuda()
    z, mel_mean, mel_log_std, log_dets, log_dur, log_dur_t, attn, _ = model(t("tokens"), t("token_lengths"), t("mels"), t("mel_lengths"), None, None, None)
    mle = MLE_Loss(model.hp)(z=z, mean=mel_mean, std=mel_log_std, log_dets=log_dets, lengths=t("mel_lengths"))
    length = torch.nn.functional.mse_loss(log_dur, log_dur_t)
    model.zero_grad()
    (mle + length).backward()
    torch.cuda.synchronize()
    worst_cos, n = 1.0, 0
    for k, p in model.named_parameters():
        want = grads.get(k)
        if want is None or p.grad is None or want.abs().max() < 1e-6:
            continue
        got = p.grad.cpu().double().flatten(); want = want.double().flatten()
        cos = (got @ want / (got.norm() * want.norm() + 1e-30)).item()
        ratio = (got.norm() / (want.norm() + 1e-30)).item()
        worst_cos = min(worst_cos, cos); n += 1
        assert cos > 0.98 and 0.9 < ratio < 1.1, (k, cos, ratio)
    assert n > 100
    print("worst cosine", worst_cos)


def test_deferred_tail_weight_gradients_equal_plain_backward():
    """decoder.defer_tail_wgrads(): the 1x1 weight-gradient groups and the weight-norm backward behind them are queued during
    backward() and issued by flush_tail_wgrads() (bench.py overlaps the gradient all-reduce with them).  Same gradients, bit for bit;
    the tail set is exactly the parameters that change between "before flush" and "after flush"."""
    from glow_tts_amd import decoder as D
    from glow_tts_amd.modules import MLE_Loss
    sd, _, r = load_case("tiny_vanilla.npz")
    model = build("Vanilla", "f32", sd)
    t = lambda k: torch.from_numpy(r[k]).cuda()

    def step():
        z, mel_mean, mel_log_std, log_dets, log_dur, log_dur_t, _, _ = model(t("tokens"), t("token_lengths"), t("mels"), t("mel_lengths"), None, None, None)
        loss = MLE_Loss(model.hp)(z=z, mean=mel_mean, std=mel_log_std, log_dets=log_dets, lengths=t("mel_lengths")) + \
            torch.nn.functional.mse_loss(log_dur, log_dur_t)
        model.zero_grad(set_to_none=True)
        loss.backward()

    step()
    want = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    with D.defer_tail_wgrads():
        step()
        assert len(D.TAIL["pending"]) >= 2
    tail_ids = {id(p) for p in model._dec_stacks.tail_leaves()}
    assert tail_ids
    for k, p in model.named_parameters():                      # everything outside the tail is final before the flush
        if id(p) not in tail_ids and k in want:
            assert torch.equal(p.grad, want[k]), k
    D.flush_tail_wgrads()
    torch.cuda.synchronize()
    for k, p in model.named_parameters():
        if k in want:
            assert torch.equal(p.grad, want[k]), k


@pytest.mark.late(1)
def test_graphed_train_step_matches_eager():
    """glow_tts_amd.graph_step.GraphedTrainStep: the replayed hipGraph gives the eager step's loss and gradients (f32), also for a
    second batch copied into the static buffers.  Runs in a child process: a failed stream capture takes the process down on
    ROCm 7.2 instead of raising, and must not take the rest of the suite with it."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "graph_step_check.py")], capture_output=True, text=True, timeout=600, cwd=os.path.dirname(here))
    assert out.returncode == 0 and "GRAPH STEP OK" in out.stdout, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])


@pytest.mark.late(1)
def test_graphed_inference_matches_eager():
    """glow_tts_amd.graph_infer.GraphedInference: encoder graph + one length read + bucketed inverse-flow graph reproduce
    GlowTTS.inference (same injected noise) for several length scales, Vanilla and speaker-conditioned.  Child process, see above."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "graph_infer_check.py")], capture_output=True, text=True, timeout=600, cwd=os.path.dirname(here))
    assert out.returncode == 0 and "GRAPH INFER OK" in out.stdout, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])


@pytest.mark.late(2)
def test_two_ranks_on_one_gpu(tmp_path):
    """Data parallel with the real model on hardware: two ranks share the one GPU of the test box - RCCL if it
    accepts that (probed in a child with a time-out: it usually refuses a duplicate device), else gloo over the same FlatGradReducer /
    ActNorm / loss-weighting code.  tests/dp_gpu_check.py does the checks; its log is kept in gpurun_out/ (copied to profiles/)."""
    import os, socket, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(here)

    def port():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p
    os.makedirs(os.path.join(repo, "gpurun_out"), exist_ok=True)
    log = os.path.join(repo, "gpurun_out", "dp_2rank_one_gpu.log")
    open(log, "w").close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    backend, why = "gloo", ""
    try:
        pr = subprocess.run([sys.executable, os.path.join(here, "dp_gpu_check.py"), "probe", str(port()), log], capture_output=True, text=True, timeout=180, cwd=repo, env=env)
        if pr.returncode == 0 and "NCCL PROBE OK" in pr.stdout:
            backend = "nccl"
        else:
            why = (pr.stderr.strip().splitlines() or ["?"])[-1][:300]
    except subprocess.TimeoutExpired:
        why = "probe timed out"
    with open(log, "a") as f:
        f.write(f"[dp] backend {backend}" + (f" (RCCL refused two ranks on one device: {why})" if backend == "gloo" else " (RCCL, two ranks on one device)") + "\n")
    out = subprocess.run([sys.executable, os.path.join(here, "dp_gpu_check.py"), backend, str(port()), log], capture_output=True, text=True, timeout=900, cwd=repo, env=env)
    assert out.returncode == 0 and "DP GPU CHECK OK" in out.stdout, (out.returncode, out.stdout[-3000:], out.stderr[-3000:])


def _random_batch(seed, vocab):
    """Ragged batches at the sizes a fixed fixture never hits: one utterance, one token, as many frames as tokens, odd lengths,
    a padded frame axis that is not a multiple of the squeeze factor (the tail frame is cut)."""
    g = torch.Generator().manual_seed(seed)
    B = [1, 2, 3, 5, 4, 7][seed % 6]
    tl = torch.randint(1, 12, (B,), generator=g)
    extra = 2 * torch.randint(0, 20, (B,), generator=g)
    ml = tl * 2 + extra * (torch.arange(B) % 3 != 0)          # every third utterance: exactly 2 frames per token (T_y/2 == T_x)
    Tx, Ty = int(tl.max()), int(ml.max()) + (seed % 4 == 1)  # every fourth case: an odd padded frame axis (cut by the squeeze)
    tokens = torch.randint(1, vocab, (B, Tx), generator=g)
    mels = torch.randn(B, 12, Ty, generator=g) * 0.7
    for b in range(B):
        tokens[b, tl[b]:] = 0
        mels[b, :, ml[b]:] = 0
    return tokens, tl, mels, ml


@pytest.mark.parametrize("seed", range(12))
def test_random_ragged_shapes_match_oracle_f32(seed):
    """Shape fuzz of the training graph against the oracle (Vanilla / SE alternating): outputs, bit-exact alignment, losses and every
    parameter gradient at batch / length combinations the golden fixtures do not cover."""
    from glow_tts_amd.modules import MLE_Loss
    mode, fname = [("Vanilla", "tiny_vanilla.npz"), ("SE", "tiny_se.npz")][seed % 2]
    sd, _, r = load_case(fname)
    model = build(mode, "f32", sd)
    vocab = sd["layer_Dict.Encoder.layer_Dict.Embedding.weight"].shape[0]
    tokens, tl, mels, ml = _random_batch(seed, vocab)
    spk = torch.randint(0, 5, (tokens.shape[0],), generator=torch.Generator().manual_seed(seed)) if mode == "SE" else None
    if seed == 0:                                             # lengths that the squeeze cannot split are an error, as in Modules.py:71
        with pytest.raises(AssertionError):
            model(tokens.cuda(), tl.cuda(), mels.cuda(), (ml - 1).cuda(), None, None, None)
    out = model(tokens.cuda(), tl.cuda(), mels.cuda(), ml.cuda(), spk.cuda() if spk is not None else None, None, None)
    z, mel_mean, mel_log_std, log_dets, log_dur, log_dur_t, attn, _ = out
    loss = MLE_Loss(model.hp)(z=z, mean=mel_mean, std=mel_log_std, log_dets=log_dets, lengths=ml.cuda()) + \
        torch.nn.functional.mse_loss(log_dur, log_dur_t)
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    cfg = tiny_cfg(mode)
    sdg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    o = O.forward_train(sdg, cfg, tokens, tl, mels, ml, spk)
    omle, olen = O.train_losses(o, ml, cfg)
    (omle + olen).backward()
    assert attn.shape == o["attn"].shape and torch.equal(attn.cpu(), o["attn"]), "alignment differs from the oracle"
    for got, key, tol in [(z, "z", 1e-4), (mel_mean, "mel_mean", 1e-4), (mel_log_std, "mel_log_std", 1e-4), (log_dur, "log_dur", 1e-4),
                          (log_dur_t, "log_dur_target", 1e-5), (log_dets, "log_dets", 1e-3)]:
        assert got.shape == o[key].shape, key
        assert (got.detach().cpu() - o[key].detach()).abs().max() <= tol, key
    assert abs(loss.item() - (omle + olen).item()) <= 1e-4 * max(1.0, abs((omle + olen).item()))
    for k, p in model.named_parameters():
        want = sdg[k].grad
        if want is None:
            continue
        got = p.grad.cpu() if p.grad is not None else torch.zeros_like(want)
        err = (got - want).abs().max().item() / (want.abs().max().item() + 1e-5)
        assert err < 5e-3, (k, err)


@pytest.mark.parametrize("seed", range(8))
def test_random_inference_shapes_match_oracle(seed):
    """Shape fuzz of GlowTTS.inference (Modules.py:128-204) against the oracle: per-utterance length scales, single utterances,
    one-token inputs, predicted lengths that are odd (the inverse decoder's Squeeze cuts the last frame, :897-898)."""
    mode, fname = [("Vanilla", "tiny_vanilla.npz"), ("SE", "tiny_se.npz")][seed % 2]
    sd, _, r = load_case(fname)
    model = build(mode, "f32", sd)
    vocab = sd["layer_Dict.Encoder.layer_Dict.Embedding.weight"].shape[0]
    g = torch.Generator().manual_seed(100 + seed)
    B = [1, 2, 4, 3, 6, 1, 5, 2][seed]
    tl = torch.randint(1, 14, (B,), generator=g)
    if seed == 5:
        tl[:] = 1
    tokens = torch.randint(1, vocab, (B, int(tl.max())), generator=g)
    for b in range(B):
        tokens[b, tl[b]:] = 0
    ls = 0.6 + 1.2 * torch.rand(B, generator=g)
    if seed == 5:
        ls = ls + 2.0                                         # (a one-frame prediction squeezes to nothing: the reference's convs reject it)
    ns = [0.0, 0.333, 0.667, 1.0][seed % 4]
    spk = torch.randint(0, 5, (B,), generator=g) if mode == "SE" else None
    noise = torch.randn(B, 12, 400, generator=g)
    want_mels, want_len, want_attn = O.inference(sd, tiny_cfg(mode), tokens, tl, noise, ls, noise_scale=ns, speakers=spk)
    mels, lengths, attn = model.inference(tokens.cuda(), tl.cuda(), None, None, spk.cuda() if spk is not None else None, None, None, None,
                                          noise_scale=ns, length_scale=ls.cuda(), noises=noise.cuda())
    torch.cuda.synchronize()
    assert torch.equal(lengths.cpu(), want_len)
    assert attn.shape == want_attn.shape and torch.equal(attn.cpu().to(want_attn.dtype), want_attn)
    assert mels.shape == want_mels.shape and (mels.cpu() - want_mels).abs().max() <= 2e-4


@pytest.mark.parametrize("seed", [2, 3, 4, 5, 8, 9])
def test_random_ragged_shapes_bf16_nll(seed):
    """The same ragged batches in bf16 precision (the benchmarked arithmetic): the MLE loss stays within the north-star bound of the fp32
    oracle, z within bf16 resolution, the duration loss within 1e-2 (its targets move when a near-tie of the alignment flips)."""
    from glow_tts_amd.modules import MLE_Loss
    mode, fname = [("Vanilla", "tiny_vanilla.npz"), ("SE", "tiny_se.npz")][seed % 2]
    sd, _, r = load_case(fname)
    model = build(mode, "bf16", sd)
    vocab = sd["layer_Dict.Encoder.layer_Dict.Embedding.weight"].shape[0]
    tokens, tl, mels, ml = _random_batch(seed, vocab)
    spk = torch.randint(0, 5, (tokens.shape[0],), generator=torch.Generator().manual_seed(seed)) if mode == "SE" else None
    z, mel_mean, mel_log_std, log_dets, log_dur, log_dur_t, attn, _ = model(tokens.cuda(), tl.cuda(), mels.cuda(), ml.cuda(),
                                                                              spk.cuda() if spk is not None else None, None, None)
    mle = MLE_Loss(model.hp)(z=z, mean=mel_mean, std=mel_log_std, log_dets=log_dets, lengths=ml.cuda())
    cfg = tiny_cfg(mode)
    o = O.forward_train(sd, cfg, tokens, tl, mels, ml, spk)
    omle, olen = O.train_losses(o, ml, cfg)
    assert abs(mle.item() - omle.item()) <= 1e-3 * max(1.0, abs(omle.item())), (mle.item(), omle.item())
    assert (z.cpu() - o["z"]).abs().max() <= 5e-2
    assert attn.shape == o["attn"].shape and float((attn.cpu() != o["attn"]).float().mean()) < 0.02
