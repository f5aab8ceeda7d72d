"""GPU parity: HIP Monotonic Alignment Search (through the C ABI) vs the oracle and the golden vectors
produced by the reference's core.pyx.  Bar: bit-exact paths AND bit-exact cumulative scores."""
import numpy as np
import pytest
import torch

from oracle import mas_ref

pytestmark = pytest.mark.gpu

MAS_CASES = ["ragged", "ties", "square", "one_token", "x1000", "wide", "more_tokens"]


def hip_mas(v, tx, ty, want_q=False):
    from glow_tts_amd import monotonic_align as ma
    vd = torch.from_numpy(v).cuda()
    txd, tyd = torch.from_numpy(tx).cuda(), torch.from_numpy(ty).cuda()
    if want_q:
        idx, q = ma.maximum_path_idx(vd, txd, tyd, want_q=True)
    else:
        idx, q = ma.maximum_path_idx(vd, txd, tyd), None
    path = ma.path_from_idx(idx, v.shape[1], torch.int32)
    torch.cuda.synchronize()
    return path.cpu().numpy(), idx.cpu().numpy(), (q.cpu().numpy() if q is not None else None)


@pytest.mark.parametrize("name", MAS_CASES)
def test_golden_vectors(name, golden_dir):
    d = np.load(f"{golden_dir}/mas_cases.npz")
    v, tx, ty = d[f"{name}/value"], d[f"{name}/t_x"], d[f"{name}/t_y"]
    path, idx, q = hip_mas(v, tx, ty, want_q=True)
    assert np.array_equal(path, d[f"{name}/path"].astype(np.int32))
    assert np.bitwise_xor.reduce(q.view(np.uint32).ravel()) == d[f"{name}/q_xor"]     # cumulative scores bit-exact
    _, q_ref = mas_ref.maximum_path_c(v, tx, ty, return_q=True)
    assert np.array_equal(q.view(np.uint32), q_ref.view(np.uint32))


@pytest.mark.parametrize("Tx,Ty,B", [(1, 1, 2), (3, 7, 5), (64, 64, 3), (65, 200, 4), (120, 800, 8), (129, 403, 3),
                                     (200, 1000, 4), (257, 999, 2), (400, 1201, 2), (512, 1024, 2)])
def test_random_vs_oracle(Tx, Ty, B):
    rng = np.random.default_rng(Tx * 1000 + Ty)
    v = rng.normal(-100, 30, (B, Tx, Ty)).astype(np.float32)
    tx = rng.integers(1, Tx + 1, B).astype(np.int32)
    ty = np.array([rng.integers(t, Ty + 1) for t in tx], dtype=np.int32)
    tx[0], ty[0] = Tx, Ty
    mask = (np.arange(Tx)[None, :, None] < tx[:, None, None]) & (np.arange(Ty)[None, None, :] < ty[:, None, None])
    v = (v * mask).astype(np.float32)
    want, q_ref = mas_ref.maximum_path_c(v, tx, ty, return_q=True)
    path, idx, q = hip_mas(v, tx, ty, want_q=True)
    assert np.array_equal(path, want)
    assert np.array_equal(q.view(np.uint32), q_ref.view(np.uint32))
    for b in range(B):
        assert (idx[b, ty[b]:] == -1).all() and (idx[b, :ty[b]] == want[b, :, :ty[b]].argmax(0)).all()


@pytest.mark.parametrize("Tx,Ty,B", [(5, 3, 2), (40, 17, 3), (130, 64, 2), (3, 1, 1), (200, 120, 2)])
def test_more_tokens_than_frames_follows_the_reference(Tx, Ty, B):
    """t_x > t_y: no monotonic alignment exists.  core.pyx then accumulates nothing (its column loops are empty) and backtracks over the raw
    inputs; the oracle C restatement does the same and was checked against the compiled core.pyx for these shapes.  All three device entry
    points and the host twin reproduce that path bit for bit; the scores stay the inputs."""
    rng = np.random.default_rng(Tx * 77 + Ty)
    v = rng.normal(-100, 30, (B, Tx, Ty)).astype(np.float32)
    tx = np.full(B, Tx, np.int32); ty = np.full(B, Ty, np.int32)
    if B > 1:
        tx[1], ty[1] = Tx - 1, max(1, Ty - 1)                      # ragged, still more tokens than frames
    mask = (np.arange(Tx)[None, :, None] < tx[:, None, None]) & (np.arange(Ty)[None, None, :] < ty[:, None, None])
    v = (v * mask).astype(np.float32)
    want, q_ref = mas_ref.maximum_path_c(v, tx, ty, return_q=True)
    assert np.array_equal(q_ref, v)
    path, idx, q = hip_mas(v, tx, ty, want_q=True)
    assert np.array_equal(path, want) and np.array_equal(q, v)
    path_t, _, _, _ = hip_mas_t(v, tx, ty)
    assert np.array_equal(path_t, want)
    from glow_tts_amd.monotonic_align import maximum_path_host
    ph = maximum_path_host(v.copy(), mask.astype(np.float32))
    assert np.array_equal(np.asarray(ph).astype(np.int32), want)


def test_ties_and_sentinel_scale():
    rng = np.random.default_rng(5)
    B, Tx, Ty = 6, 50, 170
    v = (np.round(rng.normal(-3, 2, (B, Tx, Ty))) * 1e6).astype(np.float32)     # many exact ties, |scores| beyond 1e7
    tx = np.full(B, Tx, np.int32); ty = np.full(B, Ty, np.int32)
    want = mas_ref.maximum_path_c(v, tx, ty)
    path, _, _ = hip_mas(v, tx, ty)
    assert np.array_equal(path, want)


def test_full_size_properties():
    """BASELINE size (B=32, 120x800): compare with the oracle and check the structural invariants."""
    rng = np.random.default_rng(11)
    B, Tx, Ty = 32, 120, 800
    v = rng.normal(-100, 30, (B, Tx, Ty)).astype(np.float32)
    ty = (2 * rng.integers(300, 401, B)).astype(np.int32); ty[0] = Ty
    tx = np.round(0.15 * ty).astype(np.int32)
    mask = (np.arange(Tx)[None, :, None] < tx[:, None, None]) & (np.arange(Ty)[None, None, :] < ty[:, None, None])
    v = (v * mask).astype(np.float32)
    path, idx, _ = hip_mas(v, tx, ty)
    assert np.array_equal(path, mas_ref.maximum_path_c(v, tx, ty))
    for b in range(B):
        i = idx[b, :ty[b]]
        assert i[0] == 0 and i[-1] == tx[b] - 1 and ((np.diff(i) == 0) | (np.diff(i) == 1)).all()
        assert path[b].sum() == ty[b]


def test_wrapper_matches_reference_signature():
    """monotonic_align.maximum_path(value, mask) -> same dtype/device as value (__init__.py:6-21)."""
    from glow_tts_amd import monotonic_align as ma
    rng = np.random.default_rng(3)
    v = torch.from_numpy(rng.normal(-50, 10, (2, 9, 30)).astype(np.float32)).cuda()
    mask = torch.zeros(2, 9, 30, device="cuda")
    mask[0, :9, :30] = 1; mask[1, :4, :22] = 1
    p = ma.maximum_path(v, mask)
    assert p.dtype == v.dtype and p.device == v.device and p.shape == v.shape
    want = mas_ref.maximum_path_c((v * mask).cpu().numpy(), np.array([9, 4], np.int32), np.array([30, 22], np.int32))
    assert np.array_equal(p.cpu().numpy().astype(np.int32), want)


# ---------------------------------------------------------------------------------------------------------------------------------
# The PRODUCTION layout: GlowTTS.forward runs the search on the transposed score matrix value_t [B, T_mel, T_tok] written by the
# log-prior GEMM (glowtts_mas_dp_f32_t -> mas_dp_kernel<R, VEC, WRITE_Q, transposed = true>).  Same bar as above - paths AND
# cumulative scores bit-exact against core.pyx's golden vectors (core.pyx:9-45) and the oracle - on R = 1, 2, 3, 4, 6, 8 token rows
# per lane and on both loader variants (float2 / float4 gathers when T_tok % R == 0, scalar otherwise).
# ---------------------------------------------------------------------------------------------------------------------------------
def hip_mas_t(v, tx, ty, want_q=True):
    """v [B,Tx,Ty] (numpy) -> runs glowtts_mas_dp_f32_t on its transpose; returns path [B,Tx,Ty] i32, idx, q [B,Tx,Ty]."""
    import ctypes
    from glow_tts_amd import _lib, alignment, monotonic_align as ma
    from helpers import launch_counts, launch_reset
    L = alignment._lib2()
    B, Tx, Ty = v.shape
    vt = torch.from_numpy(np.ascontiguousarray(v.transpose(0, 2, 1))).cuda()
    txd, tyd = torch.from_numpy(tx).cuda(), torch.from_numpy(ty).cuda()
    idx = torch.empty(B, Ty, dtype=torch.int32, device="cuda")
    qt = vt.clone() if want_q else None          # cells outside the band keep their input value, as core.pyx leaves `values`
    launch_reset()
    _lib.check(L.glowtts_mas_dp_f32_t(_lib.ptr(vt), _lib.ptr(txd), _lib.ptr(tyd), _lib.ptr(idx), _lib.ptr(qt), B, Tx, Ty, -1e9, _lib.stream()), "mas_t")
    path = ma.path_from_idx(idx, Tx, torch.int32)
    torch.cuda.synchronize()
    kinds = [k for k in launch_counts() if k.startswith("mas_dp<") or k.startswith("mas_dp2<")]
    q = np.ascontiguousarray(qt.cpu().numpy().transpose(0, 2, 1)) if want_q else None
    return path.cpu().numpy(), idx.cpu().numpy(), q, kinds


@pytest.mark.parametrize("name", MAS_CASES)
def test_transposed_golden_vectors(name, golden_dir):
    d = np.load(f"{golden_dir}/mas_cases.npz")
    v, tx, ty = d[f"{name}/value"], d[f"{name}/t_x"], d[f"{name}/t_y"]
    path, idx, q, kinds = hip_mas_t(v, tx, ty)
    assert np.array_equal(path, d[f"{name}/path"].astype(np.int32))
    assert np.bitwise_xor.reduce(q.view(np.uint32).ravel()) == d[f"{name}/q_xor"]
    _, q_ref = mas_ref.maximum_path_c(v, tx, ty, return_q=True)
    assert np.array_equal(q.view(np.uint32), q_ref.view(np.uint32))
    assert kinds and all(k.endswith(",t>") or k.startswith("mas_dp2<") for k in kinds), kinds


# (Tx, Ty, B, kernel variant expected): R = ceil(Tx / 64) rounded to an instantiated value; vector loads iff R in {2, 4} and Tx % R == 0;
# "dp2": two rows per lane with an even T_tok take the hand-scheduled production kernel mas_dp2_kernel (mas_dp2.hip)
T_SHAPES = [(1, 9, 3, "R1,scalar"), (63, 200, 3, "R1,scalar"), (64, 300, 3, "R1,scalar"), (65, 301, 3, "R2,scalar"),
            (66, 67, 3, "dp2"), (100, 100, 3, "dp2"), (126, 127, 3, "dp2"), (128, 129, 3, "dp2"), (70, 1000, 3, "dp2"),
            (120, 800, 8, "dp2"), (128, 640, 4, "dp2"), (129, 403, 3, "R3,scalar"), (192, 500, 3, "R3,scalar"),
            (200, 1000, 4, "R4,vec"), (254, 700, 2, "R4,scalar"), (256, 1024, 2, "R4,vec"), (257, 999, 2, "R6,scalar"),
            (384, 901, 2, "R6,scalar"), (400, 1201, 2, "R8,scalar"), (512, 1024, 2, "R8,scalar")]


@pytest.mark.parametrize("Tx,Ty,B,variant", T_SHAPES)
def test_transposed_random_vs_oracle(Tx, Ty, B, variant):
    rng = np.random.default_rng(Tx * 1000 + Ty + 7)
    v = rng.normal(-100, 30, (B, Tx, Ty)).astype(np.float32)
    tx = rng.integers(1, Tx + 1, B).astype(np.int32)
    ty = np.array([rng.integers(t, Ty + 1) for t in tx], dtype=np.int32)
    tx[0], ty[0] = Tx, Ty
    if B > 2:
        tx[1], ty[1] = min(Tx, Ty), min(Tx, Ty)                   # t_x == t_y: the diagonal-only path
    mask = (np.arange(Tx)[None, :, None] < tx[:, None, None]) & (np.arange(Ty)[None, None, :] < ty[:, None, None])
    v = (v * mask).astype(np.float32)
    want, q_ref = mas_ref.maximum_path_c(v, tx, ty, return_q=True)
    path, idx, q, kinds = hip_mas_t(v, tx, ty)
    assert kinds == (["mas_dp2<q>"] if variant == "dp2" else [f"mas_dp<{variant},q,t>"]), kinds
    assert np.array_equal(path, want)
    assert np.array_equal(q.view(np.uint32), q_ref.view(np.uint32))
    for b in range(B):
        assert (idx[b, ty[b]:] == -1).all() and (idx[b, :ty[b]] == want[b, :, :ty[b]].argmax(0)).all()
    # and without q_out_t (the instantiation the training step uses)
    path2, _, _, kinds2 = hip_mas_t(v, tx, ty, want_q=False)
    assert kinds2 == (["mas_dp2<noq>"] if variant == "dp2" else [f"mas_dp<{variant},noq,t>"]), kinds2
    assert np.array_equal(path2, want)


def test_transposed_ties_sentinel_scale_and_full_size():
    rng = np.random.default_rng(6)
    B, Tx, Ty = 6, 50, 170
    v = (np.round(rng.normal(-3, 2, (B, Tx, Ty))) * 1e6).astype(np.float32)     # exact ties, |scores| beyond the Python twin's -1e7 sentinel
    tx = np.full(B, Tx, np.int32); ty = np.full(B, Ty, np.int32)
    want, q_ref = mas_ref.maximum_path_c(v, tx, ty, return_q=True)
    path, _, q, _ = hip_mas_t(v, tx, ty)
    assert np.array_equal(path, want) and np.array_equal(q.view(np.uint32), q_ref.view(np.uint32))
    # the same on the hand-scheduled kernel's shapes (two rows per lane), with scores large enough that cumulative sums pass the -1e9
    # sentinel of row -1 and of the diagonal: core.pyx then prefers the sentinel (max(v_cur, max_neg_val)), and so must the kernel
    for (B, Tx, Ty, scale) in [(5, 100, 340, 1e6), (4, 128, 200, 2e7), (3, 66, 90, 3e7)]:
        v = (np.round(rng.normal(-3, 2, (B, Tx, Ty))) * scale).astype(np.float32)
        tx = np.full(B, Tx, np.int32); ty = np.full(B, Ty, np.int32)
        tx[-1], ty[-1] = Tx - 7, Ty - 31
        want, q_ref = mas_ref.maximum_path_c(v, tx, ty, return_q=True)
        path, _, q, kinds = hip_mas_t(v, tx, ty)
        assert kinds == ["mas_dp2<q>"], kinds
        assert np.array_equal(path, want) and np.array_equal(q.view(np.uint32), q_ref.view(np.uint32)), (Tx, Ty, scale)
    # BASELINE size, ragged (Set V of SURVEY 8d): B = 32, 120 x 800 and the reference's maximum 200 x 1000
    for (B, Tx, Ty) in [(32, 120, 800), (32, 200, 1000)]:
        v = rng.normal(-100, 30, (B, Tx, Ty)).astype(np.float32)
        ty = (2 * rng.integers(Ty * 3 // 8, Ty // 2 + 1, B)).astype(np.int32); ty[0] = Ty
        tx = np.maximum(1, np.round(Tx / Ty * ty)).astype(np.int32)
        mask = (np.arange(Tx)[None, :, None] < tx[:, None, None]) & (np.arange(Ty)[None, None, :] < ty[:, None, None])
        v = (v * mask).astype(np.float32)
        want, q_ref = mas_ref.maximum_path_c(v, tx, ty, return_q=True)
        path, idx, q, _ = hip_mas_t(v, tx, ty)
        assert np.array_equal(path, want) and np.array_equal(q.view(np.uint32), q_ref.view(np.uint32))


def test_strict_twin_of_maximum_path_c():
    """glowtts_mas_f32: the direct replacement of core.pyx:40 maximum_path_c (value, pre-zeroed int32 path, t_xs, t_ys)."""
    import ctypes
    from glow_tts_amd import _lib
    d_rng = np.random.default_rng(21)
    for (B, Tx, Ty) in [(3, 37, 96), (4, 120, 801), (2, 200, 1000)]:
        v = d_rng.normal(-100, 30, (B, Tx, Ty)).astype(np.float32)
        tx = d_rng.integers(1, Tx + 1, B).astype(np.int32)
        ty = np.array([d_rng.integers(t, Ty + 1) for t in tx], dtype=np.int32)
        tx[0], ty[0] = Tx, Ty
        mask = (np.arange(Tx)[None, :, None] < tx[:, None, None]) & (np.arange(Ty)[None, None, :] < ty[:, None, None])
        v = (v * mask).astype(np.float32)
        vd = torch.from_numpy(v).cuda()
        path = torch.zeros(B, Tx, Ty, dtype=torch.int32, device="cuda")
        scratch = torch.empty(B, Ty, dtype=torch.int32, device="cuda")
        txd, tyd = torch.from_numpy(tx).cuda(), torch.from_numpy(ty).cuda()
        _lib.check(_lib.lib().glowtts_mas_f32(_lib.ptr(vd), _lib.ptr(path), _lib.ptr(txd), _lib.ptr(tyd),
                                              _lib.ptr(scratch), B, Tx, Ty, -1e9, _lib.stream()), "glowtts_mas_f32")
        torch.cuda.synchronize()
        assert np.array_equal(path.cpu().numpy(), mas_ref.maximum_path_c(v, tx, ty))
