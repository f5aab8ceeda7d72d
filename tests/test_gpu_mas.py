"""GPU parity: HIP Monotonic Alignment Search (through the C ABI) vs the oracle and the golden vectors
produced by the reference's core.pyx.  Bar: bit-exact paths AND bit-exact cumulative scores."""
import numpy as np
import pytest
import torch

from oracle import mas_ref

pytestmark = pytest.mark.gpu

MAS_CASES = ["ragged", "ties", "square", "one_token", "x1000", "wide"]


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
