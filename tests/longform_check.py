"""Child of tests/test_gpu_benchmarked_sizes.py::test_long_form_inverse_at_full_width (it captures hipGraphs; a failed capture takes the
process down).  BASELINE config 5's second half at full width: 2 utterances x 200 tokens -> more than 2000 mel frames each through
`GlowTTS.inference` (eager) and `GraphedInference` (two replayed graphs) against `oracle.inference` with the same injected noise."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import glowtts_ref as O                                  # noqa: E402
from test_gpu_benchmarked_sizes import _build, _hp                    # noqa: E402
from glow_tts_amd.graph_infer import GraphedInference                 # noqa: E402

torch.manual_seed(11)
g = torch.Generator().manual_seed(12)
model = _build("Vanilla", "f32")
with torch.no_grad():
    for f in model.layer_Dict["Decoder"].layer_Dict["Flows"]:
        end = f.layers[2].layer_Dict["End"]
        end.weight.copy_(torch.randn(end.weight.shape, generator=g) * 0.02)
        end.bias.copy_(torch.randn(end.bias.shape, generator=g) * 0.02)
        f.layers[1].weight.add_(0.05 * torch.randn(4, 4, generator=g))
        f.layers[0].logs.copy_(torch.randn(f.layers[0].logs.shape, generator=g) * 0.1)
        f.layers[0].bias.copy_(torch.randn(f.layers[0].bias.shape, generator=g) * 0.1)
        f.layers[0].initialized = True
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
cfg = O.Cfg.from_yaml_dict(_hp("Vanilla", "f32"))
B, Tt = 2, 200
tl = torch.tensor([200, 163])
tokens = torch.randint(2, 35, (B, Tt), generator=g)
for b in range(B):
    tokens[b, tl[b]:] = 1
ls = torch.tensor([11.0, 12.5])
noise = torch.randn(B, 80, 4096, generator=g)
with torch.no_grad():
    want, wl, wa = O.inference(sd, cfg, tokens, tl, noise, ls, noise_scale=0.667)
assert int(wl.min()) >= 2000, wl
for precision in ("f32", "bf16"):
    m = _build("Vanilla", precision, sd).cuda().eval()
    with torch.no_grad():
        mels, lengths, attn = m.inference(tokens.cuda(), tl.cuda(), None, None, None, None, None, None, noise_scale=0.667, length_scale=ls.cuda(), noises=noise.cuda())
    torch.cuda.synchronize()
    if precision == "f32":                                  # the reference's arithmetic: lengths and alignment exact, mels within 2e-4
        assert torch.equal(lengths.cpu(), wl), (lengths, wl)
        assert torch.equal(attn.cpu().to(wa.dtype), wa)
        err = (mels.cpu() - want).abs().max().item()
        assert mels.shape == want.shape and err <= 2e-4, err
        print(f"eager f32: {tuple(mels.shape)} frames {wl.tolist()}, max |mel - oracle| {err:.2e}")
    else:
        # bf16 encoder arithmetic may move a ceil(exp(log_dur) * scale) by one frame (observed: 2295 vs 2296), which shifts everything behind
        # it: the bf16 leg checks the length drift, finiteness / masking, and (below) that the graphed path reproduces the eager one; the bf16
        # inverse flow itself is compared at this length in tests/test_gpu_wavenet_fused.py::test_fused_inverse_matches_per_conv_launches
        assert (lengths.cpu() - wl).abs().max() <= 4, (lengths, wl)
        assert torch.isfinite(mels).all()
        for b in range(B):
            n = (int(lengths[b]) // 2) * 2
            assert (mels[b, :, n:] == -4.0).all()
        print(f"eager bf16: frames {lengths.tolist()} (oracle {wl.tolist()})")
    gi = GraphedInference(m, mel_buckets=(2048, 2560, 3072))
    for rep in range(2):
        gm, gl, ga = gi(tokens.cuda(), tl.cuda(), noise_scale=0.667, length_scale=ls.cuda(), noises=noise.cuda())
        torch.cuda.synchronize()
        assert torch.equal(gl, lengths) and torch.equal(ga, attn)
        gerr = (gm - mels).abs().max().item()
        assert gm.shape == mels.shape and gerr <= (1e-4 if precision == "f32" else 0.1), (precision, rep, gerr)      # (another padded length: other row tiles, other summation order)
    print(f"graphed {precision}: max |graphed - eager| {gerr:.2e}")
print("LONGFORM OK")
