"""Child process of tests/test_gpu_model.py::test_graphed_train_step_matches_eager."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_gpu_model import build, load_case          # noqa: E402
from glow_tts_amd.graph_step import GraphedTrainStep  # noqa: E402
from glow_tts_amd.modules import MLE_Loss             # noqa: E402

sd, _, r = load_case("tiny_vanilla.npz")
model = build("Vanilla", "f32", sd)                     # eager reference
gmodel = build("Vanilla", "f32", sd)                    # graphed (captured before any eager backward of ITS parameters)
mle = MLE_Loss(model.hp)
t = lambda k: torch.from_numpy(r[k]).cuda()


def loss_fn(m, tokens, tl, mels, ml):
    z, mm, ms, ld, dur, durt, _, _ = m(tokens, tl, mels, ml, None, None, None)
    return mle(z=z, mean=mm, std=ms, log_dets=ld, lengths=ml) + torch.nn.functional.mse_loss(dur, durt)


b1 = (t("tokens"), t("token_lengths"), t("mels"), t("mel_lengths"))
b2 = (b1[0].flip(0).contiguous(), b1[1].flip(0).contiguous(), (b1[2].flip(0) * 0.9).contiguous(), b1[3].flip(0).contiguous())
want = []
for b in (b1, b2):
    model.zero_grad(set_to_none=True)
    l = loss_fn(model, *b)
    l.backward()
    torch.cuda.synchronize()
    want.append((l.item(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
step = GraphedTrainStep(gmodel, loss_fn)
for b, (wl, wg) in zip((b1, b2), want):
    l = step(*b)
    torch.cuda.synchronize()
    assert abs(l.item() - wl) <= 1e-5 * max(1.0, abs(wl)), (l.item(), wl)
    for k, p in gmodel.named_parameters():
        if k in wg:
            assert (p.grad - wg[k]).abs().max() <= 1e-5 * max(1.0, wg[k].abs().max().item()), k
print("GRAPH STEP OK")
