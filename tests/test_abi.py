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
    assert L.glowtts_abi_version() == 7


def test_product_path_has_no_oracle_import():
    """The product package must never import, dlopen or execute the oracle (or any CPU fallback of the GPU path): no mention of it in
    any Python or HIP / C++ source of the package, exactly one place that opens a shared library (`_lib.py`, the in-tree
    libglowtts_hip.so), no dlopen / system / popen in the native sources, no subprocess use in the package."""
    pkg = os.path.join(REPO, "glow_tts_amd")
    cdll_sites = []
    for root, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(root, f)
            if f.endswith(".py"):
                src = open(path).read()
                assert "oracle" not in re.sub(r"#.*", "", src).replace('"""', ""), f"{f} mentions oracle"
                assert not re.search(r"\bsubprocess\b|os\.system|os\.popen", src), f"{f} spawns processes"
                if re.search(r"ctypes\.(CDLL|cdll|PyDLL)|LoadLibrary", src):
                    cdll_sites.append(os.path.relpath(path, pkg))
            elif f.endswith((".hip", ".h", ".cpp", ".c")):
                src = open(path).read()
                assert "oracle" not in src.lower(), f"{f} mentions oracle"
                assert not re.search(r"\bdlopen\b|\bsystem\s*\(|\bpopen\b|\bexecv", src), f"{f} loads or executes external code"
    assert cdll_sites == ["_lib.py"], cdll_sites
    lib_src = open(os.path.join(pkg, "_lib.py")).read()
    assert 'LIB_PATH = os.path.join(_HERE, "libglowtts_hip.so")' in lib_src


def test_host_mas_twin_matches_core_pyx_vectors(golden_dir):
    """glowtts_mas_f32_host (the C-ABI twin of core.pyx:40 `maximum_path_c` for host-resident score matrices) on the golden
    vectors the reference's compiled core.pyx produced: paths and cumulative scores bit-exact; and the Python wrapper's signature."""
    import numpy as np
    import torch
    L = ctypes.CDLL(os.path.join(REPO, "glow_tts_amd", "libglowtts_hip.so"))
    L.glowtts_mas_f32_host.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_int]
    d = np.load(os.path.join(golden_dir, "mas_cases.npz"))
    for name in ["ragged", "ties", "square", "one_token", "x1000", "wide", "more_tokens"]:
        for threads in (1, 3):
            v = d[f"{name}/value"].copy()
            tx, ty = np.ascontiguousarray(d[f"{name}/t_x"], np.int32), np.ascontiguousarray(d[f"{name}/t_y"], np.int32)
            path = np.zeros(v.shape, np.int32)
            assert L.glowtts_mas_f32_host(v.ctypes.data, path.ctypes.data, tx.ctypes.data, ty.ctypes.data, *v.shape, -1e9, threads) == 0
            assert np.array_equal(path, d[f"{name}/path"].astype(np.int32)), name
            assert np.bitwise_xor.reduce(v.view(np.uint32).ravel()) == d[f"{name}/q_xor"], name
    from glow_tts_amd.monotonic_align import maximum_path_host
    v = torch.from_numpy(d["ragged/value"]).double()
    tx, ty = d["ragged/t_x"], d["ragged/t_y"]
    mask = torch.from_numpy(((np.arange(v.shape[1])[None, :, None] < tx[:, None, None]) & (np.arange(v.shape[2])[None, None, :] < ty[:, None, None])).astype(np.float64))
    p = maximum_path_host(v, mask)
    assert p.dtype == v.dtype and np.array_equal(p.numpy().astype(np.int32), d["ragged/path"].astype(np.int32))
    # more tokens than frames (no monotonic alignment): as core.pyx behaves - nothing accumulated, backtrack over the raw inputs (oracle C,
    # itself checked against the compiled core.pyx for such shapes, tests/golden/make_golden.py)
    from oracle import mas_ref
    rng = np.random.default_rng(3)
    w = rng.normal(-100, 30, (2, 40, 17)).astype(np.float32)
    txw, tyw = np.array([40, 31], np.int32), np.array([17, 9], np.int32)
    w *= ((np.arange(40)[None, :, None] < txw[:, None, None]) & (np.arange(17)[None, None, :] < tyw[:, None, None]))
    want, _ = mas_ref.maximum_path_c(w, txw, tyw, return_q=True)
    w2, pw = w.copy(), np.zeros(w.shape, np.int32)
    assert L.glowtts_mas_f32_host(w2.ctypes.data, pw.ctypes.data, txw.ctypes.data, tyw.ctypes.data, *w.shape, -1e9, 1) == 0
    assert np.array_equal(pw, want) and np.array_equal(w2, w)
    bad = np.array([v.shape[1] + 1] * v.shape[0], np.int32)
    vv = d["ragged/value"].copy(); pp = np.zeros(vv.shape, np.int32)
    assert L.glowtts_mas_f32_host(vv.ctypes.data, pp.ctypes.data, bad.ctypes.data, ty.ctypes.data, *vv.shape, -1e9, 1) == -1      # GLOWTTS_E_ARG


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


def test_collater_matches_reference_layout_and_buckets():
    """glow_tts_amd.data.Collater against a direct restatement of Datasets.py:23-39,67-74,225-250 (pad with '<E>', -Max_Abs_Mel, 0;
    truncate mels to a multiple of Num_Squeeze; mels transposed to [B, Mel, T]) and the bucketed / pinned variants."""
    import numpy as np
    import torch
    from glow_tts_amd.data import Collater
    rng = np.random.default_rng(0)
    batch = []
    for tt, tm in [(7, 41), (12, 33), (3, 58), (9, 2)]:
        batch.append((rng.integers(1, 40, tt), rng.standard_normal((tm, 5)).astype(np.float32), int(rng.integers(0, 9)),
                      rng.random(tm).astype(np.float32)))
    tok, tl, mel, ml, spk, ge2e, pit = Collater(num_squeeze=2, end_token_id=99, max_abs_mel=4.0)(batch)
    want_ml = [(len(m) // 2) * 2 for _, m, _, _ in batch]
    assert tl.tolist() == [7, 12, 3, 9] and ml.tolist() == want_ml and ge2e is None and spk.tolist() == [b[2] for b in batch]
    assert tok.shape == (4, 12) and mel.shape == (4, 5, max(want_ml)) and pit.shape == (4, max(want_ml))
    for b, (t, m, _, p) in enumerate(batch):
        assert tok[b, :len(t)].tolist() == t.tolist() and (tok[b, len(t):] == 99).all()
        assert torch.equal(mel[b, :, :want_ml[b]], torch.from_numpy(m[:want_ml[b]].T.copy())) and (mel[b, :, want_ml[b]:] == -4.0).all()
        n = min(len(p), pit.shape[1])
        assert torch.equal(pit[b, :n], torch.from_numpy(p[:n])) and (pit[b, n:] == 0).all()
    c2 = Collater(num_squeeze=2, end_token_id=99, token_buckets=[8, 16, 32], mel_buckets=[32, 64, 128], pin_memory=False)
    tok2, _, mel2, ml2, _, _, _ = c2(batch)
    assert tok2.shape == (4, 16) and mel2.shape == (4, 5, 64) and ml2.tolist() == want_ml
    assert torch.equal(tok2[:, :12], tok) and (tok2[:, 12:] == 99).all() and torch.equal(mel2[:, :, :max(want_ml)], mel)
    import pytest
    with pytest.raises(ValueError):
        Collater(token_buckets=[4])(batch)


def test_hot_kernels_use_no_scratch_memory(tmp_path):
    """Register-allocation guard: none of the kernels on the benchmarked path may spill (a DGATE epilogue variant once pushed the
    1024-thread LDS-DMA kernel from 90 to 128 VGPRs + 348 bytes of scratch and cost 2 % of the step).  Reads the code objects' metadata
    notes out of the built library (llvm-objdump --offloading + llvm-readelf --notes; nothing is executed)."""
    import re
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(f"{llvm}/llvm-objdump") and os.path.exists(f"{llvm}/llvm-readelf")):
        pytest.skip("ROCm LLVM tools not found")
    so = os.path.join(REPO, "glow_tts_amd", "libglowtts_hip.so")
    shutil.copy(so, tmp_path / "lib.so")
    subprocess.run([f"{llvm}/llvm-objdump", "--offloading", "lib.so"], cwd=tmp_path, check=True, capture_output=True)
    objs = [f for f in os.listdir(tmp_path) if f.endswith("gfx950")]
    assert objs, "no gfx950 code objects in the library"
    kernels = {}
    for f in objs:
        notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        for name, scratch, vgpr in re.findall(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)", notes, re.S):
            kernels[name] = (int(scratch), int(vgpr))
    assert len(kernels) > 100
    hot = [r"wn_fwd_kernelILb1ELb0ELb1ELb0ELi0E", r"wn_bwd_kernel", r"conv_dma_kernelILi\dELi\dELi2ELb0E", r"conv_chain_kernel", r"conv_skinny_kernel", r"wgrad_kernel", r"attn_(fwd|bwd)_mfma_kernel",
           r"mas_dp_kernelILi[12]E", r"mas_dp2_kernel", r"ln_(fwd|bwd)_kernel", r"actnorm_inv", r"gate_bwd_kernel", r"mas_path_linear_kernel", r"expand_fwd4_kernel",
           r"squeeze_kernel", r"optim|radam|adam"]
    seen = {h: 0 for h in hot}
    for name, (scratch, vgpr) in kernels.items():
        for h in hot:
            if re.search(h, name):
                seen[h] += 1
                assert scratch == 0, f"{name} spills: {scratch} bytes of scratch per lane at {vgpr} VGPRs"
    for h, n in seen.items():
        assert n > 0 or h == r"optim|radam|adam", f"pattern {h} matched no kernel"
