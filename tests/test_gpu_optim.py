"""Optimizer side of the step (glow_tts_amd/optim.py) against the reference's golden run and against torch."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "radam_case.npz")


def test_radam_clip_scheduler_replay_reference_golden():
    """The golden run of the reference's Radam.py (+ clip_grad_norm_ + Modified_Noam_Scheduler): same parameters after every one of
    the 12 steps (both rectification branches, weight decay, clipping active on every 4th step), same moments, same learning rates."""
    from glow_tts_amd.optim import Modified_Noam_Scheduler, Noam_Scheduler, RAdam, clip_grad_norm_
    d = np.load(GOLD)
    LR, B1, B2, EPS, WD, BASE, CLIP, STEPS = d["hyper"]
    params = [torch.nn.Parameter(torch.from_numpy(d["p0/%d" % i]).cuda()) for i in range(4)]
    opt = RAdam(params, lr=LR, betas=(B1, B2), eps=EPS, weight_decay=WD)
    sch = Modified_Noam_Scheduler(opt, base=BASE)
    for step in range(1, int(STEPS) + 1):
        for i, p in enumerate(params):
            p.grad = torch.from_numpy(d["g%d/%d" % (step, i)]).cuda()
        total = clip_grad_norm_(params, CLIP)
        assert abs(total.item() - float(d["norm%d" % step])) <= 1e-5 * max(1.0, float(d["norm%d" % step]))
        assert abs(opt.param_groups[0]["lr"] - d["lrs"][step - 1]) <= 1e-12
        opt.step()
        sch.step()
        for i, p in enumerate(params):
            assert (p.detach().cpu() - torch.from_numpy(d["p%d/%d" % (step, i)])).abs().max() <= 2e-6, (step, i)
    for i, p in enumerate(params):
        st = opt.state[p]
        assert st["step"] == int(STEPS)
        assert (st["exp_avg"].cpu() - torch.from_numpy(d["m/%d" % i])).abs().max() <= 1e-6
        assert (st["exp_avg_sq"].cpu() - torch.from_numpy(d["v/%d" % i])).abs().max() <= 1e-6
    sd = opt.state_dict()                       # the reference's layout: per-parameter step / exp_avg / exp_avg_sq, one group
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and set(sd["param_groups"][0]) >= {"lr", "betas", "eps", "weight_decay", "params"}
    o2 = RAdam([torch.nn.Parameter(torch.zeros(1, device="cuda"))], lr=LR)
    s2 = Noam_Scheduler(o2, warmup_steps=50)
    for v in d["noam50"]:
        assert abs(o2.param_groups[0]["lr"] - v) <= 1e-12
        o2.step(); s2.step()


def test_clip_and_fused_scale_match_torch_on_stacked_and_loose_tensors():
    """Global-norm clipping vs torch.nn.utils.clip_grad_norm_ on a mix of leaf stacks (coalesced into one job) and loose tensors of
    awkward sizes; the fused form (coefficient applied inside the update) gives the same parameters as clip-then-step."""
    from glow_tts_amd.decoder import LeafStack
    from glow_tts_amd.optim import RAdam, _JobTable, clip_grad_norm_, grad_norm_and_coef
    g = torch.Generator().manual_seed(3)
    mk = lambda *s: torch.nn.Parameter(torch.randn(*s, generator=g).cuda())
    leaves = [mk(5, 7, 3) for _ in range(6)]
    LeafStack(leaves, (6,)).tensor()                                   # the six leaves become views of one flat tensor
    loose = [mk(4097), mk(1), mk(33, 129), mk(8192)]
    params = leaves + loose
    flatg = torch.randn(6, 5, 7, 3, generator=g).cuda() * 2
    for i, p in enumerate(leaves):
        p.grad = flatg[i]                                              # adjacent gradients, as _StackedLeaves.backward produces
    for p in loose:
        p.grad = torch.randn(p.shape, generator=g).cuda() * 2
    ref = [p.detach().clone().requires_grad_() for p in params]
    for r, p in zip(ref, params):
        r.grad = p.grad.clone()
    want = torch.nn.utils.clip_grad_norm_(ref, 3.0)
    opt_a, opt_b = RAdam(params, lr=1e-2, weight_decay=1e-3), None
    pb = [torch.nn.Parameter(p.detach().clone()) for p in params]
    for q, p in zip(pb, params):
        q.grad = p.grad.clone()
    opt_b = RAdam(pb, lr=1e-2, weight_decay=1e-3)
    norm, coef = grad_norm_and_coef(pb, 3.0)
    opt_b.step(grad_scale=coef)                                        # fused
    got = clip_grad_norm_(params, 3.0)                                 # in place, like torch
    assert abs(got.item() - want.item()) <= 1e-5 * want.item() and abs(norm.item() - want.item()) <= 1e-5 * want.item()
    for r, p in zip(ref, params):
        assert (r.grad - p.grad).abs().max() <= 1e-6
    opt_a.step()
    for a, b in zip(params, pb):
        assert (a - b).abs().max() <= 1e-6
    tab = next(iter(opt_a._tables.values()))
    assert isinstance(tab, _JobTable) and tab.njobs == 1 + len(loose)  # 6 stacked leaves -> one job
