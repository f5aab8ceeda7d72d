"""Shared helpers for the parity tests: fixture loading and the oracle config of the tiny golden model."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    d = np.load(os.path.join(GOLDEN, name))
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd/")}
    grads = {k[5:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("grad/")}
    rest = {k: d[k] for k in d.files if "/" not in k}
    return sd, grads, rest


def tiny_cfg(mode):
    from oracle import glowtts_ref as O
    return O.Cfg(mode=mode, mel_dim=12, enc_channels=32, prenet_stacks=2, ffn_channels=48, enc_stacks=2,
                 dp_channels=24, n_flows=3, wn_channels=32, wn_layers=2, n_speakers=5, spk_dim=16, pro_dim=16)


def tiny_hp_dict(mode):
    """The same shrunk Hyper_Parameters dict tests/golden/make_golden.py gave the reference."""
    import yaml
    with open(os.path.join(os.path.dirname(GOLDEN), "..", "glow_tts_amd", "Hyper_Parameters.default.yaml")) as f:
        hp = yaml.safe_load(f)
    hp["Mode"] = mode
    hp["Sound"]["Mel_Dim"] = 12
    e = hp["Encoder"]
    e["Channels"] = 32
    e["Prenet"]["Stacks"] = 2
    e["Transformer"]["Conv"]["Calc_Channels"] = 48
    e["Transformer"]["Stacks"] = 2
    e["Duration_Predictor"]["Channels"] = 24
    hp["Decoder"]["Stack"] = 3
    hp["Decoder"]["Affine_Coupling"]["Calc_Channels"] = 32
    hp["Decoder"]["Affine_Coupling"]["WaveNet"]["Num_Layers"] = 2
    hp["Speaker_Embedding"]["Num_Speakers"] = 5
    hp["Speaker_Embedding"]["Embedding_Size"] = 16
    hp["Speaker_Embedding"]["Type"] = "LUT"
    hp["Prosody_Encoder"]["Size"] = 16
    return hp
