"""GPU parity at the BENCHMARKED width: the flow decoder at the default Hyper_Parameters sizes (C = 160, 192 WaveNet channels,
4 layers, k = 5) forward + BACKWARD through `DecoderFunction` against torch.autograd on the oracle (oracle/glowtts_ref.decoder,
pinned against the reference; Modules.py:780-887 for the coupling / WaveNet arithmetic).

At this width the bf16 path takes the kernels bench.py times - `conv_chain_kernel<RESSKIP,COUPLE>` / `<LINEAR,DGATE>`, the LDS-DMA
`conv_dma_kernel` for the k = 5 conv, its data gradient and the 1x1 convs, and the grouped wide-staging `wgrad_kernel` - which the
32-channel golden model never selects.  The library's launch log (glowtts_launch_count) is asserted so that a silent fallback to the
register-staged kernels cannot pass for coverage.

Bars: f32 mode (the reference's arithmetic): every parameter gradient and d(mel) within 2e-3 of the oracle's, relative to the
tensor's largest entry.  bf16 mode: cosine >= 0.98 and norm ratio in [0.9, 1.1] per gradient tensor (bf16 operands, fp32
accumulation; the errors are stated, not hidden: the test prints the worst tensor)."""
import pytest
import torch

from oracle import glowtts_ref as O
from helpers import full_width_state, launch_counts, launch_reset

pytestmark = pytest.mark.gpu

N_FLOWS, B, TM = 3, 4, 640


@pytest.fixture(autouse=True)
def _mixed_backward():
    """One flow on the fused data-gradient kernel, two on the per-conv launches - what a chip-filling batch runs (the automatic choice of
    decoder.TUNE["fused_wn_bwd"] would fuse all three flows of this small batch and leave the per-conv backward untested here)."""
    from glow_tts_amd import decoder as D
    old = D.TUNE["fused_wn_bwd"]
    D.TUNE["fused_wn_bwd"] = N_FLOWS // 2
    yield
    D.TUNE["fused_wn_bwd"] = old
LENGTHS = [640, 522, 240, 2]            # ragged: the longest, two inner ones, one squeezed frame


def _case(spk_dim, seed):
    g = torch.Generator().manual_seed(seed)
    cfg, sd = full_width_state(N_FLOWS, g, spk_dim=spk_dim)
    ml = torch.tensor(LENGTHS)
    mels = (torch.randn(B, 80, TM, generator=g) * 1.5).clamp(-4, 4)
    spk = None
    if spk_dim:
        spk = torch.randn(B, spk_dim, generator=g)
        spk = spk / spk.norm(dim=1, keepdim=True)
    wz = torch.randn(B, 80, TM, generator=g)
    wl = torch.randn(B, generator=g) * 0.05
    return cfg, sd, mels, ml, spk, wz, wl


def _oracle_grads(cfg, sd, mels, ml, spk, wz, wl, drop=None):
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = mels.clone().requires_grad_(True)
    s = spk.clone().requires_grad_(True) if spk is not None else None
    mask = O.mask_from_lengths(ml, mels.shape[2])
    zo, ldo, _ = O.decoder(sdg, x, mask, cfg, speakers=s, drop=drop)
    ((zo * wz).sum() + (ldo * wl).sum()).backward()
    return zo.detach(), ldo.detach(), {k: v.grad for k, v in sdg.items()}, x.grad, (s.grad if s is not None else None)


def _hip_grads(cfg, sd, mels, ml, spk, wz, wl, precision, drop_p=0.0):
    from glow_tts_amd import decoder as D
    dc = D.DecoderConfig(cfg.mel_dim, cfg.n_flows, cfg.n_squeeze, cfg.n_split, cfg.wn_channels, cfg.wn_layers, cfg.wn_kernel, precision)
    P = {k: v.cuda().requires_grad_(True) for k, v in sd.items()}
    x = mels.cuda().requires_grad_(True)
    s = spk.cuda().requires_grad_(True) if spk is not None else None
    W = D.stack_decoder_weights(P, dc)
    cond = D.conditioning(P, dc, speakers=s) if s is not None else None
    launch_reset()
    z, logdet, _ = D.DecoderFunction.apply(dc, x, ml.cuda(), cond, drop_p, None, None, None, *W)
    ((z * wz.cuda()).sum() + (logdet * wl.cuda()).sum()).backward()
    torch.cuda.synchronize()
    return z.detach().cpu(), logdet.detach().cpu(), {k: p.grad.cpu() for k, p in P.items()}, x.grad.cpu(), (s.grad.cpu() if s is not None else None), launch_counts()


def _count(counts, prefix):
    return sum(n for k, n in counts.items() if k.startswith(prefix))


@pytest.mark.parametrize("spk_dim", [0, 256])
def test_full_width_backward_f32(spk_dim):
    case = _case(spk_dim, 1234 + spk_dim)
    zo, ldo, go, dxo, dso = _oracle_grads(*case)
    z, ld, g, dx, ds, counts = _hip_grads(*case, precision=0)
    assert (z - zo).abs().max() <= 2e-4
    assert ((ld - ldo).abs() <= 1e-3 * ldo.abs().clamp_min(1.0)).all()
    worst = ("", 0.0)
    for k, want in go.items():
        err = (g[k] - want).abs().max().item() / (want.abs().max().item() + 1e-6)
        worst = max(worst, (k, err), key=lambda t: t[1])
        assert err <= 2e-3, (k, err)
    mask = O.mask_from_lengths(case[3], TM)
    assert ((dx - dxo) * mask).abs().max() <= 2e-3 * dxo.abs().max()
    if dso is not None:
        assert (ds - dso).abs().max() <= 2e-3 * dso.abs().max()
    print("f32 worst gradient:", worst)
    # exact-fp32 mode runs the register-staged kernel on v_mfma_f32_32x32x2_f32 for every conv
    assert _count(counts, "conv_cl<GATE,5,f32") == N_FLOWS * 4 and _count(counts, "conv_dma") == 0, counts


@pytest.mark.parametrize("spk_dim", [0, 256])
def test_full_width_backward_bf16_takes_the_benchmarked_kernels(spk_dim):
    case = _case(spk_dim, 4321 + spk_dim)
    zo, ldo, go, dxo, dso = _oracle_grads(*case)
    z, ld, g, dx, ds, counts = _hip_grads(*case, precision=1)
    # which kernels ran (per flow: 4 In convs, 3 Res_Skip, the chained last Res_Skip -> End, and the mirror image in the backward)
    L = 4
    assert _count(counts, "wn_fwd<") == N_FLOWS, counts                                   # the fused coupling network: one launch per flow
    assert _count(counts, "conv_dma<GATE,5>") == 0 and _count(counts, "conv_dma<RESSKIP,1>") == 0 and _count(counts, "conv_chain<RESSKIP,COUPLE>") == 0, counts
    # the backward the training step runs by default: per-conv launches for the flows it reaches first, the fused data-gradient kernel for
    # the last half (decoder.TUNE["fused_wn_bwd"], parity in tests/test_gpu_wavenet_fused.py)
    nfb = N_FLOWS // 2
    assert _count(counts, "wn_bwd<") == nfb, counts
    assert _count(counts, "conv_dma<LINEAR,5>") == (N_FLOWS - nfb) * L, counts           # In_l data gradient
    assert _count(counts, "conv_dma<DGATE,1>") == (N_FLOWS - nfb) * (L - 1), counts
    assert _count(counts, "conv_chain<LINEAR,DGATE>") == N_FLOWS - nfb, counts
    # all In_l weight gradients: one grouped launch of the LDS-DMA kernel (round 4: 192 x 64 x 5-tap tiles, csrc/wgrad_cl.hip wgrad_dma_kernel)
    assert _count(counts, "wgrad_dma<5>/grouped") == 1 and _count(counts, "wgrad<5,") == 0, counts
    # Res_Skip_l AND (round 4) Start / End, whose operands - d h0, x_a, d(m, logs), the skip sum - now exist as bf16 rows: one launch, no fp32 staging
    # (round 5: in two row splits on the LDS-DMA kernel's 192 x 192 one-tap tiles when the batch halves - decoder.TUNE["wgrad_tail_splits"] -, else the staged kernel)
    split = B % 2 == 0
    assert _count(counts, "wgrad_dma<1>/grouped") == (1 if split else 0) and _count(counts, "wgrad<1,bf16,dybf16,xbf16,wide>/grouped") == (0 if split else 1), counts
    assert _count(counts, "wgrad<1,bf16,dyf32,xf32") == 0, counts
    assert _count(counts, "conv_cl<GATE") == 0 and _count(counts, "conv_cl<RESSKIP") == 0 and _count(counts, "conv_cl<DGATE") == 0, counts
    mask = O.mask_from_lengths(case[3], TM)
    rms = lambda t: t.double().pow(2).mean().sqrt().item()
    print("bf16 z: max abs error", ((z - zo) * mask).abs().max().item(), "relative rms", rms((z - zo) * mask) / rms(zo * mask))
    assert ((z - zo) * mask).abs().max() <= 0.1 and rms((z - zo) * mask) <= 1e-2 * rms(zo * mask)
    # log-determinant: a sum of len/2 * 80 coupling log-scales computed from bf16 operands -> relative part + random-walk part
    n_el = (case[3] // 2 * 80).float()
    print("bf16 log-determinants", ld.tolist(), "oracle", ldo.tolist())
    assert ((ld - ldo).abs() <= 2e-3 * ldo.abs() + 2e-3 * n_el.sqrt()).all()
    report = []
    for k, want in go.items():
        a, b = g[k].flatten().double(), want.flatten().double()
        cos = (a @ b / (a.norm() * b.norm() + 1e-30)).item()
        ratio = (a.norm() / (b.norm() + 1e-30)).item()
        report.append((cos, ratio, k))
    report.sort()
    print("bf16 worst gradient tensors (cosine, norm ratio):", report[:3])
    for cos, ratio, k in report:
        assert cos >= 0.98 and 0.9 <= ratio <= 1.1, (k, cos, ratio)
    a, b = (dx * mask).flatten().double(), (dxo * mask).flatten().double()
    assert (a @ b / (a.norm() * b.norm())).item() >= 0.98 and 0.9 <= (a.norm() / b.norm()).item() <= 1.1
    if dso is not None:
        a, b = ds.flatten().double(), dso.flatten().double()
        assert (a @ b / (a.norm() * b.norm())).item() >= 0.98


@pytest.mark.parametrize("spk_dim", [0, 256])
def test_full_width_dropout_masks_agree_between_forward_backward_and_precisions(spk_dim):
    """Training-mode WaveNet dropout (Modules.py:861-862, p = 0.3 here) at the benchmarked width.  The keep mask is a counter hash of (seed, row,
    channel), regenerated by the backward; the bf16 path draws it in the fused `wn_fwd_kernel` / `wn_bwd_kernel` / `conv_dma_kernel<DGATE>` / `conv_chain<LINEAR,DGATE>`,
    the f32 path in the register-staged kernels.  Same seed => the two precisions see the SAME masks, so (1) z agrees to bf16 accuracy,
    (2) every gradient of the bf16 path has cosine >= 0.98 with the f32 path's - a forward / backward mask mismatch in any of the DMA kernels
    would decorrelate them (p = 0.3 rescales 30 % of the gate gradients to zero), (3) the conditioning gradient, taken BEFORE the mask, agrees
    too, (4) the same seed reproduces z bit for bit, another seed does not."""
    case = _case(spk_dim, 777 + spk_dim)
    torch.manual_seed(5)
    z32, ld32, g32, dx32, ds32, _ = _hip_grads(*case, precision=0, drop_p=0.3)
    torch.manual_seed(5)
    z16, ld16, g16, dx16, ds16, counts = _hip_grads(*case, precision=1, drop_p=0.3)
    torch.manual_seed(5)
    z16b = _hip_grads(*case, precision=1, drop_p=0.3)[0]
    torch.manual_seed(6)
    z16c = _hip_grads(*case, precision=1, drop_p=0.3)[0]
    assert torch.equal(z16, z16b) and (z16 - z16c).abs().max() > 1e-2
    nfb = N_FLOWS // 2                                   # the backward's last flows take the fused data-gradient kernel
    assert _count(counts, "wn_fwd<drop") == N_FLOWS and _count(counts, "conv_chain<LINEAR,DGATE>") == N_FLOWS - nfb
    assert _count(counts, "wn_bwd<drop") == nfb
    mask = O.mask_from_lengths(case[3], TM)
    assert ((z16 - z32) * mask).abs().max() <= 0.15
    worst = (2.0, "")
    for k, want in g32.items():
        a, b = g16[k].flatten().double(), want.flatten().double()
        cos = (a @ b / (a.norm() * b.norm() + 1e-30)).item()
        worst = min(worst, (cos, k))
        assert cos >= 0.98 and 0.9 <= (a.norm() / (b.norm() + 1e-30)).item() <= 1.1, (k, cos)
    print("dropout, bf16 vs f32 path, worst gradient cosine:", worst)
    if ds32 is not None:
        a, b = ds16.flatten().double(), ds32.flatten().double()
        assert (a @ b / (a.norm() * b.norm())).item() >= 0.98


def _hip_keep_multipliers(seed_word, flow, layer, B, T, H, p):
    """The WaveNet dropout multipliers the HIP epilogues draw (csrc/device_common.h: drop_rowkey / drop_colkey / drop_draw), restated with
    numpy uint32 arithmetic: [B, 2H, T] for gate pre-activation channel (tanh j | sigmoid H + j) of squeezed frame t of utterance b.
    Row index = position in the rows layout [B][T + 4] (two pad rows in front of every utterance)."""
    import numpy as np
    u = lambda v: np.asarray(v, dtype=np.uint64) & 0xFFFFFFFF
    thr = int(p * 65536.0 + 0.5)
    ik = np.float32(65536.0) / np.float32(65536.0 - thr)
    seed = (1000003 * flow + int(seed_word) + layer) & 0xFFFFFFFF
    rows = (np.arange(B)[:, None] * (T + 4) + 2 + np.arange(T)[None, :]).astype(np.uint64)            # [B, T]
    x = u(rows * 0x9E3779B1 + seed)
    x ^= x >> 16
    x = u(x * 0x85EBCA6B)
    x ^= x >> 13
    ck = u((np.arange(H, dtype=np.uint64) + 1) * 0x27D4EB2F)                                          # [H]
    d = u((x[:, None, :] ^ ck[None, :, None]) * 0xC2B2AE35)                                           # [B, H, T]
    d ^= d >> 16
    keep_t, keep_s = (d & 0xFFFF) >= thr, (d >> 16) >= thr
    m = np.concatenate([keep_t, keep_s], 1).astype(np.float32) * ik
    return torch.from_numpy(m)


def test_full_width_backward_f32_in_train_mode_with_the_same_dropout_masks():
    """Training mode (Modules.py:861-862 WaveNet dropout, p = 0.1): the keep masks of the HIP path - a counter hash of (seed word, flow, layer,
    row, channel pair) regenerated by the backward kernels - are restated on the host and INJECTED into the oracle, whose forward and
    torch.autograd backward then see the same network: outputs and every gradient at the eval-mode bars (until now train
    mode was covered by mask statistics and forward / backward consistency only)."""
    p_drop = 0.1
    case = _case(0, 777)
    cfg, sd, mels, ml = case[0], case[1], case[2], case[3]
    T, H = TM // cfg.n_squeeze, cfg.wn_channels
    torch.manual_seed(99)
    seed_word = int(torch.randint(0, 2 ** 31 - 1, (1,), device="cuda", dtype=torch.int32).item())     # the word DecoderFunction will draw
    masks = {}

    def drop(pfx, layer, ins):
        flow = int(pfx.split(".Flows.")[1].split(".")[0])
        key = (flow, layer)
        if key not in masks:
            masks[key] = _hip_keep_multipliers(seed_word, flow, layer, B, T, H, p_drop)
        return ins * masks[key].to(ins.dtype)
    zo, ldo, go, dxo, _ = _oracle_grads(*case, drop=drop)
    torch.manual_seed(99)
    z, ld, g, dx, _, counts = _hip_grads(*case, precision=0, drop_p=p_drop)
    kept = torch.stack([m.ne(0).float().mean() for m in masks.values()]).mean().item()
    assert len(masks) == N_FLOWS * 4 and abs(kept - (1 - p_drop)) < 5e-3, (len(masks), kept)
    assert (z - zo).abs().max() <= 2e-4, (z - zo).abs().max()
    assert ((ld - ldo).abs() <= 1e-3 * ldo.abs().clamp_min(1.0)).all()
    worst = ("", 0.0)
    for k, want in go.items():
        err = (g[k] - want).abs().max().item() / (want.abs().max().item() + 1e-6)
        worst = max(worst, (k, err), key=lambda t: t[1])
        assert err <= 2e-3, (k, err)
    mask = O.mask_from_lengths(ml, TM)
    assert ((dx - dxo) * mask).abs().max() <= 2e-3 * dxo.abs().max()
    print("train mode, f32, same dropout masks: worst gradient", worst, "kept fraction", kept)
