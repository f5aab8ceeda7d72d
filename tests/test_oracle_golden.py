"""CPU suite: the oracle (oracle/) replayed against the committed golden vectors, which were produced
by the unmodified reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import glowtts_ref as O
from oracle import mas_ref
from helpers import load_case, tiny_cfg

MAS_CASES = ["ragged", "ties", "square", "one_token", "x1000", "wide", "more_tokens"]


@pytest.mark.parametrize("name", MAS_CASES)
def test_mas_oracle_matches_reference_vectors(name, golden_dir):
    d = np.load(f"{golden_dir}/mas_cases.npz")
    path, q = mas_ref.maximum_path_c(d[f"{name}/value"], d[f"{name}/t_x"], d[f"{name}/t_y"], return_q=True)
    assert np.array_equal(path, d[f"{name}/path"].astype(np.int32))           # bit-exact alignment
    assert np.bitwise_xor.reduce(q.view(np.uint32).ravel()) == d[f"{name}/q_xor"]   # bit-exact cumulative scores
    # every column of a valid utterance holds exactly one 1, and the path is monotonic
    for b in range(path.shape[0]):
        tx, ty = d[f"{name}/t_x"][b], d[f"{name}/t_y"][b]
        assert (path[b, :, :ty].sum(0) == 1).all() and path[b, :, ty:].sum() == 0 and path[b, tx:].sum() == 0
        idx = path[b, :, :ty].argmax(0)
        assert idx[-1] == tx - 1 and (np.diff(idx) >= 0).all() and (np.diff(idx) <= 1).all()
        assert idx[0] == 0 or tx > ty                         # (more tokens than frames: the walk cannot reach the first token)


@pytest.mark.parametrize("name", ["ragged", "square", "one_token"])
def test_python_mas_twin_matches_core_pyx_vectors(name, golden_dir):
    """oracle.mas_ref.maximum_path_python (restating Modules.py:951-980, the Use_Cython_Alignment = false path that BASELINE config 1
    names) gives core.pyx's paths at ordinary score magnitudes (the two only part when a cumulative score passes -1e7)."""
    d = np.load(f"{golden_dir}/mas_cases.npz")
    path = mas_ref.maximum_path_python(d[f"{name}/value"], d[f"{name}/t_x"], d[f"{name}/t_y"])
    assert np.array_equal(path, d[f"{name}/path"].astype(np.int32))


def test_mas_oracle_matches_reference_build_if_present():
    core = mas_ref.reference_core()
    if core is None:
        pytest.skip("oracle/_ref not built (reference absent)")
    rng = np.random.default_rng(7)
    v = rng.normal(-100, 30, (3, 17, 41)).astype(np.float32)
    tx = np.array([17, 5, 9], np.int32); ty = np.array([41, 30, 9], np.int32)
    mask = (np.arange(17)[None, :, None] < tx[:, None, None]) & (np.arange(41)[None, None, :] < ty[:, None, None])
    v = (v * mask).astype(np.float32)
    p = mas_ref.maximum_path_c(v, tx, ty)
    p2 = np.zeros_like(p); v2 = v.copy()
    core.maximum_path_c(p2, v2, tx, ty)
    assert np.array_equal(p, p2)


@pytest.mark.parametrize("mode,fname", [("Vanilla", "tiny_vanilla.npz"), ("SE", "tiny_se.npz"), ("PE", "tiny_pe.npz"), ("GR", "tiny_gr.npz")])
def test_model_oracle_matches_reference_vectors(mode, fname):
    """Every Mode of Hyper_Parameters.yaml:17-18: PE adds the GST prosody encoder (Modules.py:312-385), GR the adversarial speaker
    classifier behind the gradient-reversal layer (:407-435, Gradient_Reversal_Layer.py) and per-frame pitch conditioning (:867-869)."""
    sd, grads, r = load_case(fname)
    cfg = tiny_cfg(mode)
    t = lambda k: torch.from_numpy(r[k])
    spk = t("speakers") if "speakers" in r else None
    pit = t("pitches") if "pitches" in r else None
    sdg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    o = O.forward_train(sdg, cfg, t("tokens"), t("token_lengths"), t("mels"), t("mel_lengths"), spk, pitches=pit)
    for k in ["z", "mel_mean", "mel_log_std", "log_dets", "log_dur", "log_dur_target"]:
        assert torch.allclose(o[k], t(k), atol=3e-5, rtol=1e-5), k
    assert np.array_equal(o["attn"].numpy().astype(np.int8), r["attn"])
    losses = O.train_losses(o, t("mel_lengths"), cfg, spk if mode == "GR" else None)
    assert abs(losses[0].item() - float(r["mle"])) < 1e-5 and abs(losses[1].item() - float(r["length"])) < 1e-5
    if mode == "GR":
        assert torch.allclose(o["classified"], t("classified"), atol=1e-5) and abs(losses[2].item() - float(r["ce"])) < 1e-5
    sum(losses).backward()
    for k, g in grads.items():
        og = sdg[k].grad
        og = torch.zeros_like(g) if og is None else og
        assert (og - g).abs().max() <= 1e-3 * (g.abs().max() + 1e-5), k
    with torch.no_grad():
        pm, pl = (t("mels"), t("mel_lengths")) if mode in ("PE", "GR") else (None, None)
        m, l, a = O.inference(sd, cfg, t("tokens"), t("token_lengths"), t("noise"), t("length_scale"),
                              float(r["noise_scale"]), spk, None, pm, pl, pit, t("mel_lengths") if pit is not None else None)
    assert torch.allclose(m, t("inf_mels"), atol=5e-5) and torch.equal(l, t("inf_lengths"))
    assert np.array_equal(a.numpy().astype(np.int8), r["inf_attn"])


def test_actnorm_init_and_invertibility():
    sd, _, r = load_case("tiny_vanilla.npz")
    cfg = tiny_cfg("Vanilla")
    mels, ml = torch.from_numpy(r["mels"]), torch.from_numpy(r["mel_lengths"])
    mask = O.mask_from_lengths(ml, mels.shape[2])
    x0, m0 = O.squeeze(mels, mask, 2)
    logs, bias = O.actnorm_init(x0, m0)
    assert torch.allclose(logs, sd["layer_Dict.Decoder.layer_Dict.Flows.0.layers.0.logs"], atol=1e-6)
    assert torch.allclose(bias, sd["layer_Dict.Decoder.layer_Dict.Flows.0.layers.0.bias"], atol=1e-6)
    z, _, _ = O.decoder(sd, mels, mask, cfg, reverse=False)
    back, _, _ = O.decoder(sd, z, mask, cfg, reverse=True)
    assert torch.allclose(back * mask, mels * mask, atol=2e-4)


def test_optimizer_oracle_matches_reference_golden():
    """oracle/radam_ref.py replays the golden run of the reference's Radam.py + clip_grad_norm_ + Modified_Noam_Scheduler
    (tests/golden/make_optim_golden.py): 12 steps over 4 tensors, both branches of the rectification."""
    import os
    from oracle import radam_ref as R
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "radam_case.npz"))
    LR, B1, B2, EPS, WD, BASE, CLIP, STEPS = d["hyper"]
    n = 4
    st = [(d["p0/%d" % i], np.zeros_like(d["p0/%d" % i]), np.zeros_like(d["p0/%d" % i])) for i in range(n)]
    for step in range(1, int(STEPS) + 1):
        grads = [d["g%d/%d" % (step, i)] for i in range(n)]
        total, coef = R.clip_coef(grads, CLIP)
        assert abs(total - float(d["norm%d" % step])) <= 1e-5 * max(1.0, total)
        lr = R.modified_noam_lr(LR, BASE, step - 1)
        assert abs(lr - d["lrs"][step - 1]) <= 1e-12
        st = [R.radam_step(p, g * np.float32(coef), m, v, step, lr, B1, B2, EPS, WD) for (p, m, v), g in zip(st, grads)]
        for i in range(n):
            assert np.abs(st[i][0] - d["p%d/%d" % (step, i)]).max() <= 2e-6
    for i in range(n):
        assert np.abs(st[i][1] - d["m/%d" % i]).max() <= 1e-6 and np.abs(st[i][2] - d["v/%d" % i]).max() <= 1e-6
    assert all(abs(R.noam_lr(LR, 50, i) - v) <= 1e-12 for i, v in enumerate(d["noam50"]))
