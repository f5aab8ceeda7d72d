"""Imports the *unmodified* reference (read-only mount at /root/reference) so that golden
vectors can be generated from it.  Only usable in the build container: the reference never
travels to the GPU box, and nothing here is imported by tests at run time (tests read the
committed .npz fixtures).

What it takes:
  * `hp` is parsed from ./Hyper_Parameters.yaml in the CWD at import time (Modules.py:9-13)
    -> we chdir into a temp dir holding a generated yaml.
  * Modules.py:7 imports Speaker_Embedding.Modules, an empty git submodule -> a stub module
    is registered in sys.modules.
  * monotonic_align/__init__.py:3 expects the nested build path
    monotonic_align.monotonic_align.core -> the .so built by oracle/build_ref.sh (from the
    reference's own core.pyx) is registered under that name.
"""
import importlib.machinery
import importlib.util
import os
import sys
import tempfile
import types

import yaml

REF = os.environ.get("GLOWTTS_REFERENCE", "/root/reference")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _set(d, dotted, value):
    keys = dotted.split(".")
    for k in keys[:-1]:
        d = d[k]
    d[keys[-1]] = value


def load_reference(overrides=None):
    """Returns the reference `Modules` module imported under a fresh hp built from the
    reference yaml + `overrides` ({"Decoder.Stack": 2, ...}).  Re-imports every call."""
    with open(os.path.join(REF, "Hyper_Parameters.yaml"), encoding="utf-8") as f:
        cfg = yaml.load(f, Loader=yaml.Loader)
    cfg["Device"] = "-1"
    for k, v in (overrides or {}).items():
        _set(cfg, k, v)
    tmp = tempfile.mkdtemp(prefix="glowtts_ref_")
    with open(os.path.join(tmp, "Hyper_Parameters.yaml"), "w", encoding="utf-8") as f:
        yaml.dump(cfg, f)

    # stub for the un-vendored GE2E submodule (source absent -> parity unpinned)
    import torch

    se_pkg = types.ModuleType("Speaker_Embedding")
    se_mod = types.ModuleType("Speaker_Embedding.Modules")

    class Encoder(torch.nn.Module):  # never called by the golden generator
        def __init__(self, *a, **k):
            super().__init__()

    se_mod.Encoder = Encoder
    se_mod.Normalize = lambda x: x
    se_pkg.Modules = se_mod
    sys.modules["Speaker_Embedding"] = se_pkg
    sys.modules["Speaker_Embedding.Modules"] = se_mod

    # the reference's compiled MAS, built by oracle/build_ref.sh from its own core.pyx
    so_dir = os.path.join(REPO, "oracle", "_ref", "monotonic_align")
    so = [f for f in os.listdir(so_dir) if f.startswith("core.") and f.endswith(".so")]
    if not so:
        raise RuntimeError("run oracle/build_ref.sh first")
    for name in list(sys.modules):
        if name == "monotonic_align" or name.startswith("monotonic_align."):
            del sys.modules[name]
    nested = types.ModuleType("monotonic_align.monotonic_align")
    nested.__path__ = []
    sys.modules["monotonic_align.monotonic_align"] = nested
    loader = importlib.machinery.ExtensionFileLoader(
        "monotonic_align.monotonic_align.core", os.path.join(so_dir, so[0]))
    spec = importlib.util.spec_from_loader("monotonic_align.monotonic_align.core", loader)
    core = importlib.util.module_from_spec(spec)
    loader.exec_module(core)
    sys.modules["monotonic_align.monotonic_align.core"] = core
    nested.core = core

    for name in ("Modules", "RPR_MHA", "Gradient_Reversal_Layer", "Arg_Parser"):
        sys.modules.pop(name, None)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        import Modules  # noqa: the reference module, imported with CWD = tmp
    finally:
        os.chdir(cwd)
    return Modules
