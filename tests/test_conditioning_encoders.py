"""CPU suite: the conditioning encoders of the PE / GR modes (glow_tts_amd/prosody.py - plain PyTorch modules) against
the oracle on the golden state dicts the REFERENCE produced in those modes (tests/golden/tiny_pe.npz, tiny_gr.npz; the oracle's
restatement of Modules.py:312-435 is pinned against the same reference run by tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import glowtts_ref as O
from helpers import load_case, tiny_cfg, tiny_hp_dict


def _module(cls, mode, prefix, sd):
    from glow_tts_amd.hparams import Recursive_Parse
    m = cls(Recursive_Parse(tiny_hp_dict(mode)))
    own = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    missing, unexpected = m.load_state_dict(own, strict=True)
    assert not missing and not unexpected
    return m


@pytest.mark.parametrize("mode,fname", [("PE", "tiny_pe.npz"), ("GR", "tiny_gr.npz")])
def test_prosody_encoder_matches_oracle_forward_and_backward(mode, fname):
    from glow_tts_amd.prosody import Prosody_Encoder
    sd, _, r = load_case(fname)
    m = _module(Prosody_Encoder, mode, "layer_Dict.Prosody_Encoder.", sd)
    mels, ml = torch.from_numpy(r["mels"]), torch.from_numpy(r["mel_lengths"])
    sdg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items() if "Prosody_Encoder" in k}
    want = O.prosody_encoder(sdg, mels, ml, tiny_cfg(mode))
    got = m(mels, ml)
    assert got.shape == want.shape == (3, 16) and (got - want).abs().max() <= 1e-5
    g = torch.Generator().manual_seed(1)
    w = torch.randn(want.shape, generator=g)
    (want * w).sum().backward()
    (got * w).sum().backward()
    for k, p in m.named_parameters():
        ref = sdg["layer_Dict.Prosody_Encoder." + k].grad
        assert (p.grad - ref).abs().max() <= 1e-4 * ref.abs().max() + 1e-6, k      # (Key.bias: analytically 0, softmax shift invariance)
    # ragged lengths pick different compressed steps (Modules.py:373-374): 40 -> step 4, 28 -> 3, 34 -> 4
    assert torch.ceil(ml / 8.0).long().tolist() == [5, 4, 5]


def test_speaker_classifier_gr_reverses_the_gradient():
    """Modules.py:407-435 + Gradient_Reversal_Layer.py:6-35: logits equal the oracle's; d loss / d prosody is -weight x the plain gradient."""
    from glow_tts_amd.prosody import Speaker_Classifier_GR
    sd, _, r = load_case("tiny_gr.npz")
    m = _module(Speaker_Classifier_GR, "GR", "layer_Dict.Speaker_Classifier_GR.", sd)
    cfg = tiny_cfg("GR")
    g = torch.Generator().manual_seed(2)
    pro = torch.randn(3, 16, generator=g)
    a, b = pro.clone().requires_grad_(True), pro.clone().requires_grad_(True)
    got, want = m(a), O.speaker_classifier_gr(sd, b, cfg)
    assert (got - want).abs().max() <= 1e-6
    spk = torch.from_numpy(r["speakers"])
    torch.nn.functional.cross_entropy(got, spk).backward()
    torch.nn.functional.cross_entropy(want, spk).backward()
    assert (a.grad - b.grad).abs().max() <= 1e-7
    # against the un-reversed gradient
    c = pro.clone().requires_grad_(True)
    x = c.unsqueeze(2)
    x = torch.relu(O.conv(sd, "layer_Dict.Speaker_Classifier_GR.layer.Hidden_0", x))
    x = O.conv(sd, "layer_Dict.Speaker_Classifier_GR.layer.Output_0", x).squeeze(2)
    torch.nn.functional.cross_entropy(x, spk).backward()
    assert torch.allclose(a.grad, -cfg.grl_weight * c.grad, atol=1e-8)


def test_pitch_interpolater_matches_oracle():
    from glow_tts_amd.prosody import Pitch_Interpolater
    g = torch.Generator().manual_seed(3)
    pit = torch.rand(4, 50, generator=g)
    base, new = torch.tensor([50, 37, 20, 2]), torch.tensor([64, 21, 20, 9])
    got = Pitch_Interpolater()(pit, base, new)
    want = O.pitch_interpolate(pit, base, new)
    assert got.shape == want.shape == (4, 64) and torch.allclose(got, want, atol=1e-7)
    assert Pitch_Interpolater()(pit, base, new, 70).shape == (4, 70)
