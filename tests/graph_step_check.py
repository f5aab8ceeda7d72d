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
seq = [b1, b2, b1, b2, b1, b1]                          # one optimizer step per call: a new shape costs dry warm-up passes, no extra update
la = []
for b in seq:
    ma.zero_grad(set_to_none=True)
    l = loss_fn(ma, *b)
    l.backward()
    clip_grad_norm_(list(ma.parameters()), 5.0)
    oa.step(); sa.step()
    la.append(l.item())
lb = [gstep(*b).item() for b in seq]
torch.cuda.synchronize()
assert gstep.steps_taken == len(seq)
for x, y in zip(la, lb):
    assert abs(x - y) <= 2e-5 * max(1.0, abs(x)), (la, lb)
assert oa.param_groups[0]["lr"] == ob.param_groups[0]["lr"]
for (k, pa), pb_ in zip(ma.named_parameters(), mb.parameters()):
    assert (pa - pb_).abs().max() <= 2e-5 * max(1.0, pa.abs().max().item()), k
    assert oa.state[pa]["step"] == ob.state[pb_]["step"] == len(seq), (k, oa.state[pa]["step"], ob.state[pb_]["step"])
# ---- two input SHAPES alternating (each captured graph owns its pinned job tables; a shared table would make the first graph replay
# with the second graph's pointers and row count): gradients of every replay against eager, then the full Train_Step trajectory ----
Tt, Tm = b1[0].shape[1], b1[2].shape[2]
b3 = (b1[0][:, :Tt - 2].contiguous(), b1[1].clamp(max=Tt - 2), b1[2][:, :, :Tm - 4].contiguous(), b1[3].clamp(max=Tm - 4))
md = build("Vanilla", "f32", sd)
step2 = GraphedTrainStep(md, loss_fn)
for i, b in enumerate([b1, b3, b1, b3, b3, b1]):
    model.zero_grad(set_to_none=True)
    l = loss_fn(model, *b)
    l.backward()
    lg = step2(*b)
    torch.cuda.synchronize()
    assert abs(lg.item() - l.item()) <= 1e-5 * max(1.0, abs(l.item())), (i, lg.item(), l.item())
    for (k, p), pg in zip(model.named_parameters(), md.parameters()):
        if p.grad is not None:
            assert (pg.grad - p.grad).abs().max() <= 1e-5 * max(1.0, p.grad.abs().max().item()), (i, k)
assert len(step2.graphs) == 2
me, mf = build("Vanilla", "f32", sd), build("Vanilla", "f32", sd)
oe, of = RAdam(me.parameters(), lr=1e-3, eps=1e-6, weight_decay=1e-6), RAdam(mf.parameters(), lr=1e-3, eps=1e-6, weight_decay=1e-6)
se, sf = Modified_Noam_Scheduler(oe, base=4000), Modified_Noam_Scheduler(of, base=4000)
gstep2 = GraphedTrainStep(mf, loss_fn, warmup=2, optimizer=of, scheduler=sf, max_grad_norm=5.0)
calls = [b1, b3, b1, b3, b1]                            # graphed: exactly one training step per call, also on the first call of a shape
eager_seq = list(calls)
for b in eager_seq:
    me.zero_grad(set_to_none=True)
    l = loss_fn(me, *b)
    l.backward()
    clip_grad_norm_(list(me.parameters()), 5.0)
    oe.step(); se.step()
for b in calls:                                         # no host synchronisation between the replays: the host runs ahead of the stream
    gstep2(*b)
torch.cuda.synchronize()
assert gstep2.steps_taken == len(eager_seq)
assert oe.param_groups[0]["lr"] == of.param_groups[0]["lr"]
for (k, pa), pb_ in zip(me.named_parameters(), mf.parameters()):
    assert (pa - pb_).abs().max() <= 5e-5 * max(1.0, pa.abs().max().item()), k
    assert oe.state[pa]["step"] == of.state[pb_]["step"] == len(eager_seq), (k, oe.state[pa]["step"], of.state[pb_]["step"])
# ---- the data-parallel form of bench.py: the step as two graphs (body with the k-tap weight gradients | deferred 1x1 tail) ----
from glow_tts_amd import decoder as D   # noqa: E402
mc = build("Vanilla", "f32", sd)
side = torch.cuda.Stream()


def fwd_bwd(b):
    l = loss_fn(mc, *b)
    mc.zero_grad(set_to_none=True)
    l.backward()
    return l.detach()


side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        fwd_bwd(b1)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g_body, g_tail = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with D.defer_tail_wgrads():
    with torch.cuda.graph(g_body):
        lc = fwd_bwd(b1)
with torch.cuda.graph(g_tail, pool=g_body.pool()):
    D.flush_tail_wgrads()
junk = [torch.randn(1 << 20, device="cuda") for _ in range(8)]      # allocator traffic between capture and replay must not matter
for _ in range(2):
    for p in mc.parameters():
        if p.grad is not None:
            p.grad.fill_(7.0)
    g_body.replay()
    g_tail.replay()
    torch.cuda.synchronize()
    wl, wg = want[0]
    assert abs(lc.item() - wl) <= 1e-5 * max(1.0, abs(wl)), (lc.item(), wl)
    for k, p in mc.named_parameters():
        if k in wg:
            assert (p.grad - wg[k]).abs().max() <= 1e-5 * max(1.0, wg[k].abs().max().item()), k
print("GRAPH STEP OK")
