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
                 dp_channels=24, n_flows=3, wn_channels=32, wn_layers=2, n_speakers=5, spk_dim=16, pro_dim=16,
                 pe_strides=(2, 2, 2), pe_kernels=(3, 3, 3), pe_heads=4, grl_weight=0.05, gr_hidden=1)


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
    pe = hp["Prosody_Encoder"]
    pe["Size"] = 16
    pe["Reference_Encoder"]["Conv"] = {"Kernel_Size": [3, 3, 3], "Channels": [4, 8, 8], "Strides": [2, 2, 2]}
    pe["Reference_Encoder"]["GRU"]["Size"] = 8
    pe["Style_Token"] = {"Num_Tokens": 6, "Size": 16, "Attention_Head": 4}
    hp["Speaker_Classifier_GR"]["Channels"] = [12]
    hp["Train"]["Adversarial_Speaker_Weight"] = 0.05
    return hp


def full_width_state(n_flows, g, spk_dim=0):
    """Seeded decoder weights at the default Hyper_Parameters sizes (C=160, H=192, 4 layers, k=5); spk_dim > 0 adds the SE-mode
    Speaker_l conditioning convs (Modules.py:832-838)."""
    from oracle import glowtts_ref as O
    cfg = O.Cfg(n_flows=n_flows, mode="SE" if spk_dim else "Vanilla", spk_dim=spk_dim or 256)
    sd = {}
    for f in range(cfg.n_flows):
        q = f"layer_Dict.Decoder.layer_Dict.Flows.{f}.layers"
        sd[q + ".0.logs"] = torch.randn(1, 160, 1, generator=g) * 0.1
        sd[q + ".0.bias"] = torch.randn(1, 160, 1, generator=g) * 0.1
        w4 = torch.linalg.qr(torch.randn(4, 4, generator=g))[0] + 0.05 * torch.randn(4, 4, generator=g)
        if torch.det(w4) < 0:                       # the reference keeps det > 0 (Modules.py:722-723)
            w4[:, 0] = -w4[:, 0]
        sd[q + ".1.weight"] = w4
        def wn(name, o, i, k):
            sd[f"{q}.2.layer_Dict.{name}.weight_v"] = torch.randn(o, i, k, generator=g) / (i * k) ** 0.5
            sd[f"{q}.2.layer_Dict.{name}.weight_g"] = torch.rand(o, 1, 1, generator=g) + 0.5
            sd[f"{q}.2.layer_Dict.{name}.bias"] = torch.randn(o, generator=g) * 0.05
        wn("Start", 192, 80, 1)
        for l in range(4):
            wn(f"WaveNet.layer_Dict.In_{l}", 384, 192, 5)
            wn(f"WaveNet.layer_Dict.Res_Skip_{l}", 384 if l < 3 else 192, 192, 1)
            if spk_dim:
                wn(f"WaveNet.layer_Dict.Speaker_{l}", 384, spk_dim, 1)
        sd[f"{q}.2.layer_Dict.End.weight"] = torch.randn(160, 192, 1, generator=g) * 0.02
        sd[f"{q}.2.layer_Dict.End.bias"] = torch.randn(160, generator=g) * 0.02
    return cfg, sd


def launch_counts():
    """name -> launches since the last reset (glowtts_launch_log_dump)."""
    import ctypes
    from glow_tts_amd import _lib
    buf = ctypes.create_string_buffer(1 << 17)
    _lib.lib().glowtts_launch_log_dump(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, n = line.rsplit(" ", 1)
        out[name] = int(n)
    return out


def launch_reset():
    from glow_tts_amd import _lib
    _lib.lib().glowtts_launch_log_reset()
