"""CPU suite: the C-ABI library loads and exports every symbol include/glowtts_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "glowtts_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(glowtts_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    so = os.path.join(REPO, "glow_tts_amd", "libglowtts_hip.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    L = ctypes.CDLL(so)
    names = declared_symbols()
    assert len(names) >= 5
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/glowtts_hip.h but not exported"
    assert L.glowtts_abi_version() == 1


def test_product_path_has_no_oracle_import():
    """The product package must never import the oracle (or any CPU fallback)."""
    pkg = os.path.join(REPO, "glow_tts_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in re.sub(r"#.*", "", src).replace('"""', ""), f"{f} mentions oracle"


def test_model_deepcopy_and_state_dict_roundtrip_cpu():
    """The module stays an ordinary torch.nn.Module on the host: deepcopy (run-time caches dropped), state_dict round trip."""
    import copy, os, sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import tiny_hp_dict
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS
    m = GlowTTS(Recursive_Parse(tiny_hp_dict("Vanilla")))
    m._enc_cache["x"] = object()
    c = copy.deepcopy(m)
    assert c._enc_cache == {} and c._dec_stacks is None
    sd = m.state_dict()
    assert set(sd) == set(c.state_dict()) and all(torch.equal(sd[k], c.state_dict()[k]) for k in sd)
    assert all(a.data_ptr() != b.data_ptr() for a, b in zip(m.parameters(), c.parameters()))
    c.load_state_dict(sd, strict=True)
