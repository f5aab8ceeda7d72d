"""Optimizer side of the step on the BASELINE model (515 tensors, 28.6 M parameters): clip_grad_norm_ + RAdam.step() as multi-tensor HIP
launches, alone and inside the replayed hipGraph of the whole Train_Step; torch's own per-tensor / foreach optimizers for scale."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from glow_tts_amd.graph_step import GraphedTrainStep
from glow_tts_amd.optim import Modified_Noam_Scheduler, RAdam, clip_grad_norm_
dev = torch.device("cuda:0")
model, mle, hp = bench.build_model("bf16", dev)
batch = bench.synthetic_batch(32, 120, 800, 80, 1234, dev)
def loss_fn(m, tokens, tl, mels, ml):
    z, mm, ms, ld, dur, durt, _, _ = m(tokens, tl, mels, ml, None, None, None)
    return mle(z=z, mean=mm, std=ms, log_dets=ld, lengths=ml) + torch.nn.functional.mse_loss(dur, durt)
def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3
loss_fn(model, *batch).backward()
params = list(model.parameters())
opt = RAdam(params, lr=1e-3, eps=1e-6, weight_decay=1e-6)
print(f"clip_grad_norm_ + RAdam.step (HIP multi-tensor, eager): {timeit(lambda: (clip_grad_norm_(params, 5.0), opt.step())):.3f} ms, "
      f"{len(params)} tensors -> {next(iter(opt._tables.values())).njobs} jobs")
t_opt = torch.optim.RAdam(params, lr=1e-3, eps=1e-6, weight_decay=1e-6, foreach=True)
print(f"torch clip_grad_norm_ + torch.optim.RAdam(foreach=True):  {timeit(lambda: (torch.nn.utils.clip_grad_norm_(params, 5.0), t_opt.step())):.3f} ms")
t_opt2 = torch.optim.RAdam(params, lr=1e-3, eps=1e-6, weight_decay=1e-6, foreach=False)
print(f"torch clip_grad_norm_ + torch.optim.RAdam(per tensor):    {timeit(lambda: (torch.nn.utils.clip_grad_norm_(params, 5.0, foreach=False), t_opt2.step()), n=5):.3f} ms")
model2, _, _ = bench.build_model("bf16", dev)
opt2 = RAdam(model2.parameters(), lr=1e-3, eps=1e-6, weight_decay=1e-6)
step = GraphedTrainStep(model2, loss_fn, optimizer=opt2, scheduler=Modified_Noam_Scheduler(opt2, base=4000), max_grad_norm=5.0)
print(f"whole Train_Step (fwd + losses + bwd + clip + RAdam + scheduler) as one hipGraph: {timeit(lambda: step(*batch)):.3f} ms/step")
