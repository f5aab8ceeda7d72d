"""Golden vectors for the optimizer side of the step, produced by the UNMODIFIED reference classes (Radam.py, Noam_Scheduler.py) imported
from /root/reference in the build container, and - in the same pass - a check of the oracle restatement (oracle/radam_ref.py) against them.
Usage: python tests/golden/make_optim_golden.py   ->  tests/golden/radam_case.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
from Radam import RAdam                       # noqa: E402  (the reference's)
from Noam_Scheduler import Modified_Noam_Scheduler, Noam_Scheduler   # noqa: E402
from oracle import radam_ref as R             # noqa: E402

g = torch.Generator().manual_seed(42)
shapes = [(7, 5, 3), (11,), (4, 4), (1, 13, 1)]
params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
LR, B1, B2, EPS, WD, BASE, CLIP, STEPS = 1e-3, 0.9, 0.999, 1e-6, 1e-6, 4000, 5.0, 12
opt = RAdam(params, lr=LR, betas=(B1, B2), eps=EPS, weight_decay=WD)
sch = Modified_Noam_Scheduler(opt, base=BASE)
out = {"p0/%d" % i: p.detach().numpy().copy() for i, p in enumerate(params)}
ora = [(p.detach().numpy().copy(), np.zeros(p.shape, np.float32), np.zeros(p.shape, np.float32)) for p in params]
lrs = []
for step in range(1, STEPS + 1):
    grads = [torch.randn(p.shape, generator=g) * (3.0 if step % 4 == 0 else 0.3) for p in params]     # some steps exceed the clip norm
    for p, gr in zip(params, grads):
        p.grad = gr.clone()
    total = torch.nn.utils.clip_grad_norm_(params, CLIP)
    lr_now = opt.param_groups[0]["lr"]
    lrs.append(lr_now)
    opt.step()
    sch.step()
    # oracle, same inputs
    tn, coef = R.clip_coef([x.numpy() for x in grads], CLIP)
    assert abs(tn - float(total)) <= 1e-5 * max(1.0, tn)
    ora = [R.radam_step(p_, (gr.numpy() * np.float32(coef)), m_, v_, step, lr_now, B1, B2, EPS, WD) for (p_, m_, v_), gr in zip(ora, grads)]
    for i, (p, gr) in enumerate(zip(params, grads)):
        out["g%d/%d" % (step, i)] = gr.numpy().copy()
        out["p%d/%d" % (step, i)] = p.detach().numpy().copy()
        assert np.abs(ora[i][0] - p.detach().numpy()).max() <= 2e-6, (step, i)
    out["norm%d" % step] = np.float32(float(total))
    assert abs(R.modified_noam_lr(LR, BASE, sch.last_epoch) - opt.param_groups[0]["lr"]) <= 1e-12
for i, p in enumerate(params):
    st = opt.state[p]
    out["m/%d" % i], out["v/%d" % i] = st["exp_avg"].numpy().copy(), st["exp_avg_sq"].numpy().copy()
    assert np.abs(ora[i][1] - out["m/%d" % i]).max() <= 1e-6 and np.abs(ora[i][2] - out["v/%d" % i]).max() <= 1e-6
out["lrs"] = np.array(lrs, np.float64)
out["hyper"] = np.array([LR, B1, B2, EPS, WD, BASE, CLIP, STEPS], np.float64)
# plain Noam schedule values
o2 = RAdam([torch.nn.Parameter(torch.zeros(1))], lr=LR)
s2 = Noam_Scheduler(o2, warmup_steps=50)
nl = []
for _ in range(120):
    nl.append(o2.param_groups[0]["lr"]); o2.step(); s2.step()
out["noam50"] = np.array(nl, np.float64)
assert all(abs(R.noam_lr(LR, 50, i) - v) <= 1e-12 for i, v in enumerate(nl))
np.savez_compressed(os.path.join(HERE, "radam_case.npz"), **out)
print("oracle == reference (RAdam 12 steps incl. both branches, clipping, Modified Noam, Noam); wrote radam_case.npz")
