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
# ---- the whole Train_Step (Train.py:193-233) in the graph: clip + RAdam + scheduler, against the same sequence run eagerly ----
from glow_tts_amd.optim import Modified_Noam_Scheduler, RAdam, clip_grad_norm_   # noqa: E402
ma, mb = build("Vanilla", "f32", sd), build("Vanilla", "f32", sd)
oa, ob = RAdam(ma.parameters(), lr=1e-3, eps=1e-6, weight_decay=1e-6), RAdam(mb.parameters(), lr=1e-3, eps=1e-6, weight_decay=1e-6)
sa, sb = Modified_Noam_Scheduler(oa, base=4000), Modified_Noam_Scheduler(ob, base=4000)
gstep = GraphedTrainStep(mb, loss_fn, warmup=2, optimizer=ob, scheduler=sb, max_grad_norm=5.0)
seq = [b1, b1, b1, b2, b1, b2]                          # graphed side: 2 warm-up steps + 1 replay on b1, then b2, b1, b2
la = []
for b in seq:
    ma.zero_grad(set_to_none=True)
    l = loss_fn(ma, *b)
    l.backward()
    clip_grad_norm_(list(ma.parameters()), 5.0)
    oa.step(); sa.step()
    la.append(l.item())
lb = [gstep(*b).item() for b in seq[2:]]
torch.cuda.synchronize()
for x, y in zip(la[2:], lb):
    assert abs(x - y) <= 2e-5 * max(1.0, abs(x)), (la, lb)
assert oa.param_groups[0]["lr"] == ob.param_groups[0]["lr"]
for (k, pa), pb_ in zip(ma.named_parameters(), mb.parameters()):
    assert (pa - pb_).abs().max() <= 2e-5 * max(1.0, pa.abs().max().item()), k
    assert oa.state[pa]["step"] == ob.state[pb_]["step"] == len(seq), (k, oa.state[pa]["step"], ob.state[pb_]["step"])
print("GRAPH STEP OK")
