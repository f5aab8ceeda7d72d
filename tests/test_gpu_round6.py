"""GPU tests of round 6's additions (each against a plain fp64 / fp32 torch restatement of the reference op it replaces)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cond_ref(kinds):
    """sum over kinds of vec @ (g v / ||v||)^T + bias  (Modules.py:832-845 weight_norm'ed Conv1d(D -> 2H, k = 1), :863-866), fp64."""
    out = None
    for g, v, b, vec in kinds:
        g, v, b, vec = g.double(), v.double(), b.double(), vec.double()
        w = g.view(-1, 1) * v.view(g.numel(), -1) / v.view(g.numel(), -1).norm(dim=1, keepdim=True)
        c = vec @ w.t() + b.view(1, -1)
        out = c if out is None else out + c
    return out


@pytest.mark.parametrize("D,B,supported", [(256, 32, True), (512, 40, True), (512, 41, False), (512, 64, False), (384, 64, True)])
def test_cond_linear_supported_matches_the_launch(D, B, supported):
    """glowtts_cond_linear_supported (ABI 6) is the kernels' own acceptance test: where it says yes, CondLinear runs and matches the fp64 product
    (forward and all four gradients); where it says no, glowtts_cond_linear_fwd / _bwd really reject the shape (decoder.py's guard was
    wider than the kernels' LDS budget, D = 512 with B > 40 raised in the forward)."""
    from glow_tts_amd import _lib, decoder
    L = decoder._L()
    N = 2 * 2 * 384
    assert bool(L.glowtts_cond_linear_supported(N, D, B)) == supported
    g_ = torch.Generator().manual_seed(D + B)
    g = (torch.rand(N, 1, 1, generator=g_) + 0.5).cuda().requires_grad_()
    v = torch.randn(N, D, 1, generator=g_).cuda().requires_grad_()
    b = torch.randn(N, generator=g_).cuda().requires_grad_()
    vec = torch.randn(B, D, generator=g_).cuda().requires_grad_()
    if not supported:
        with pytest.raises(_lib.GlowTTSHipError):
            decoder.CondLinear.apply(g, v, b, vec)
        return
    out = decoder.CondLinear.apply(g, v, b, vec)
    dout = torch.randn(B, N, generator=g_).cuda()
    out.backward(dout)
    ref_in = [t.detach().double().requires_grad_() for t in (g, v, b, vec)]
    ref = _cond_ref([ref_in])
    ref.backward(dout.double())
    assert (out.double() - ref).abs().max() <= 1e-4 * ref.abs().max()
    for got, want in zip((g, v, b, vec), ref_in):
        assert (got.grad.double() - want.grad).abs().max() <= 2e-4 * want.grad.abs().max(), got.shape


def test_wide_conditioning_vectors_take_the_matmul_path():
    """DecoderStacks.conditioning with 512-wide speaker vectors at B = 64 - a shape the conditioning kernels do not hold in LDS - must fall back to weight
    norm + matmul (and give the same numbers as the kernel gives at B = 32), not raise."""
    import copy
    from glow_tts_amd import hparams
    from glow_tts_amd.hparams import Recursive_Parse
    from glow_tts_amd.modules import GlowTTS
    d = copy.deepcopy(hparams.load_yaml(hparams.DEFAULT_YAML))
    d["Mode"] = "SE"
    d["Speaker_Embedding"]["Type"] = "GE2E"
    d["Speaker_Embedding"]["Embedding_Size"] = 512
    d["Decoder"]["Stack"] = 2
    d["Encoder"]["Transformer"]["Stacks"] = 1
    torch.manual_seed(3)
    model = GlowTTS(Recursive_Parse(d)).cuda()
    stacks = model._stacks(model._params())
    vec = torch.randn(64, 512, device="cuda")
    wide = stacks.conditioning(vec, None)                    # B = 64: matmul path
    half = stacks.conditioning(vec[:32].contiguous(), None)  # B = 32: the HIP kernel
    assert wide.shape == (64, 2, 4, 384)
    assert (wide[:32] - half).abs().max() <= 1e-4 * half.abs().max()
    wide.sum().backward()
    assert all(p.grad is not None for k, p in model.named_parameters() if "Speaker_" in k)
