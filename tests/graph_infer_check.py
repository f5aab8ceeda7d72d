"""Child process of tests/test_gpu_model.py::test_graphed_inference_matches_eager (a failed stream capture takes the process down)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_gpu_model import build, load_case            # noqa: E402
from glow_tts_amd.graph_infer import GraphedInference    # noqa: E402

for mode, fname in (("Vanilla", "tiny_vanilla.npz"), ("SE", "tiny_se.npz")):
    sd, _, r = load_case(fname)
    model = build(mode, "f32", sd)
    t = lambda k: torch.from_numpy(r[k]).cuda()
    tokens, tl = t("tokens"), t("token_lengths")
    spk = t("speakers") if mode == "SE" and "speakers" in r else None
    gi = GraphedInference(model, mel_buckets=(32, 64, 128, 256, 512))
    for scale in (1.0, 1.7, 0.6):                         # different mel lengths -> different buckets, one front graph
        ls = torch.tensor([scale], device="cuda")
        torch.manual_seed(3)
        noises = torch.randn(tokens.shape[0], int(model.hp.Sound.Mel_Dim), 512, device="cuda")
        want, wl, wa = model.inference(tokens, tl, None, None, spk, None, None, None, noise_scale=0.667, length_scale=ls, noises=noises)
        got, gl, ga = gi(tokens, tl, speakers=spk, noise_scale=0.667, length_scale=ls, noises=noises)
        torch.cuda.synchronize()
        assert torch.equal(wl, gl), (wl, gl)
        assert got.shape == want.shape and ga.shape == wa.shape, (got.shape, want.shape, ga.shape, wa.shape)
        assert torch.equal(ga, wa)
        assert (got - want).abs().max().item() <= 1e-5, (mode, scale, (got - want).abs().max().item())
    # internal noise: a different draw at every replay, the padded tail stays at the fill value
    a, la, _ = gi(tokens, tl, speakers=spk, noise_scale=0.667, length_scale=1.0)
    a = a.clone()
    b, lb, _ = gi(tokens, tl, speakers=spk, noise_scale=0.667, length_scale=1.0)
    torch.cuda.synchronize()
    assert torch.equal(la, lb) and not torch.equal(a, b)
    for i in range(a.shape[0]):
        n = (int(la[i]) // 2) * 2
        assert (b[i, :, n:] == -float(model.hp.Sound.Max_Abs_Mel)).all()
    assert len(gi.front) == 1 and len(gi.back) >= 2
print("GRAPH INFER OK")
